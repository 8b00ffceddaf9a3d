// a1: fused pairwise-score + per-row top-k (no N x N matrix in HBM).
//
// Replaces knn() of hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:4-10.  Bit-defined arithmetic
// (must equal oracle/knn_oracle.c, which equals torch-CPU on the build image):
//   gram   = fmaf chain over d (k-ordered, from +0)
//   sq(j)  = cascade sum of squares (groups of 16, see oracle)
//   score  = ((-sq_j) - (-2*gram)) - sq_i ;  order = (score desc, index asc)
// This file is compiled with -ffp-contract=off so only the explicit fmaf's fuse.
//
// Kernels (selection rule and measurements in launch_knn_mfma): knn_mfma_bq_kernel (k <= 32 once there is a wave per SIMD: the
// default), knn_mfma_sort_kernel (small batches; k = 33..40 with the flagged repair), knn_mfma_queue_kernel (the exact repair /
// k > 32 form), knn_kernel (any D / no workspace).  All use the same bit-defined arithmetic below.
//  * The MFMA kernels (D = 3 and D = 63, k <= 40: the HPCS shapes): one workgroup = one cloud x 32 queries.
//    The Gram tile (32 candidates x 32 queries) is produced by v_mfma_f32_32x32x2_f32 with the candidates as the A
//    operand (fetched straight from the [D,N] slab: lanes 0-31 read 128 contiguous bytes of row d, lanes 32-63 of
//    row d+1) and the queries as the B operand (held in registers for the whole kernel).  The f32 MFMA is a
//    k-ordered fma chain, i.e. exactly the pinned rounding sequence.  Lane l then owns 16 scores of query l&31
//    (rows (r&3)+8(r>>2)+4(l>>5), increasing in r) and selects among them (sorting network / batch queue / queue); a final
//    merge through LDS (score desc, index asc) yields the row.  |x|^2 comes from a pre-pass into the caller's workspace.
//    (Round 1's per-score insertion kernel, knn_mfma_kernel, was the cross-check of these until round 4; removed.)
//  * knn_kernel (any D, k <= 64): the scalar fallback described next.
//
// Fallback layout / mapping: one workgroup = one cloud x QT queries (one query per thread).  The cloud's
// [D,N] slab is walked in candidate tiles of TC points (256 for D=3, 128 otherwise) staged in LDS as [D+1][TC] (row D = sq):
// global reads are coalesced along N, LDS reads in the scoring loop are wave-uniform broadcasts.
// Each thread keeps its sorted top-KK list in registers (static indexing only).
#include <type_traits>

#include "common.h"

namespace {

constexpr int QT = 64;   // queries per workgroup (one wave): B*N/64 workgroups fill 256 CUs at B>=16

__device__ __forceinline__ float sumsq_cascade(const float* col, int D, int64_t stride) {
    float acc1 = 0.0f;
    int i = 0;
    for (; i + 16 <= D; i += 16) {
        float c = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v = col[(int64_t)(i + j) * stride];
            const float s = v * v;
            c = c + s;
        }
        acc1 = acc1 + c;
    }
    float r = 0.0f;
    for (; i < D; ++i) {
        const float v = col[(int64_t)i * stride];
        const float s = v * v;
        r = r + s;
    }
    return r + acc1;
}

template <int KK>
__device__ __forceinline__ void topk_insert(float (&bv)[KK], int (&bi)[KK], float s, int j) {
    // strict '>' : an equal score never displaces an earlier (smaller-index) candidate
#pragma unroll
    for (int m = KK - 1; m >= 1; --m) {
        const bool shift = s > bv[m - 1];
        const bool here = (s > bv[m]) && !shift;
        const float nv = shift ? bv[m - 1] : (here ? s : bv[m]);
        const int ni = shift ? bi[m - 1] : (here ? j : bi[m]);
        bv[m] = nv;
        bi[m] = ni;
    }
    if (s > bv[0]) {
        bv[0] = s;
        bi[0] = j;
    }
}

// DT > 0: D known at compile time (query lives in registers).  DT == 0: runtime D, query in LDS.
template <int DT, int KK, int TC>
__global__ __launch_bounds__(QT) void knn_kernel(const float* __restrict__ x, int D, int N, int k,
                                                  int64_t* __restrict__ idx) {
    extern __shared__ float lds[];  // [(D+1)][TC] candidates (+ [D][QT] queries when DT==0)
    const int Dr = DT > 0 ? DT : D;
    float* cand = lds;
    float* qlds = lds + (size_t)(Dr + 1) * TC;

    const int tiles_q = (N + QT - 1) / QT;
    const int b = blockIdx.x / tiles_q;
    const int q0 = (blockIdx.x % tiles_q) * QT;
    const int tid = threadIdx.x;
    const int qi = q0 + tid;
    const bool qvalid = qi < N;
    const float* xb = x + (int64_t)b * Dr * N;

    float q[DT > 0 ? DT : 1];
    float sq_i = 0.0f;
    if (qvalid) sq_i = sumsq_cascade(xb + qi, Dr, N);
    if (DT > 0) {
#pragma unroll
        for (int d = 0; d < DT; ++d) q[d] = qvalid ? xb[(int64_t)d * N + qi] : 0.0f;
    } else {
        for (int d = 0; d < Dr; ++d) qlds[d * QT + tid] = qvalid ? xb[(int64_t)d * N + qi] : 0.0f;
    }

    float bv[KK];
    int bi[KK];
#pragma unroll
    for (int m = 0; m < KK; ++m) {
        bv[m] = -INFINITY;
        bi[m] = 0x7fffffff;
    }

    for (int c0 = 0; c0 < N; c0 += TC) {
        const int nc = min(TC, N - c0);
        __syncthreads();
        for (int e = tid; e < Dr * TC; e += QT) {
            const int d = e / TC, j = e % TC;
            cand[d * TC + j] = j < nc ? xb[(int64_t)d * N + c0 + j] : 0.0f;
        }
        for (int j = tid; j < nc; j += QT) cand[Dr * TC + j] = sumsq_cascade(xb + c0 + j, Dr, N);
        __syncthreads();

        for (int j = 0; j < nc; ++j) {
            float g = 0.0f;
            if (DT > 0) {
#pragma unroll
                for (int d = 0; d < DT; ++d) g = __builtin_fmaf(q[d], cand[d * TC + j], g);
            } else {
                for (int d = 0; d < Dr; ++d) g = __builtin_fmaf(qlds[d * QT + tid], cand[d * TC + j], g);
            }
            const float inner = -2.0f * g;
            const float t = (-cand[Dr * TC + j]) - inner;
            const float s = t - sq_i;
            if (__any(s > bv[KK - 1])) topk_insert<KK>(bv, bi, s, c0 + j);
        }
    }

    if (qvalid) {
        int64_t* o = idx + ((int64_t)b * N + qi) * k;
#pragma unroll
        for (int m = 0; m < KK; ++m)
            if (m < k) o[m] = bi[m];
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void knn_sumsq_kernel(const float* __restrict__ x, int D, int N, int64_t total, float* __restrict__ sq) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t b = e / N, j = e % N;
    sq[e] = sumsq_cascade(x + b * D * N + j, D, N);
}

constexpr int MQ = 32;         // queries per workgroup (MFMA column block)
constexpr int MW = 4;          // waves per workgroup = candidate tile interleave

// Sorting-network variant of the 4-wave kernel (k <= 32).  With eight short lists per query nearly every score enters some
// list, so the per-score sorted insertion (~120 instructions) runs 16 times per tile.  Here a (score, index) pair is ONE
// 64-bit key (order-preserving image of the float in the high word, ~index in the low word: larger key = better, ties
// to the smaller index), the tile's 16 keys are sorted by a bitonic network (80 compare-exchanges of 5 instructions) and
// merged with the sorted list by the classic max(A[i], B[31-i]) + bitonic-merge step (16 + 80 compare-exchanges).
// The key is held as a DOUBLE: K = hi32 * 2^20 + (2^20 - 1 - j) < 2^52 placed in the mantissa of a number in [2^52, 2^53)
// (bit pattern 0x433.. | K: four integer instructions, no conversion), so keys order like the integers K and a
// compare-exchange is v_max_f64 + v_min_f64 -- measured 9.8 SIMD cycles against 51.5 for v_cmp_gt_u64 + four v_cndmask
// (64-bit integer compares are slow on this chip) and 36.6 for split 32-bit compares (scratch/ce_bench.hip).  Needs
// j < 2^20 (hpcs_knn sends larger clouds to the scalar kernel).
__device__ __forceinline__ double knn_key(float s, int j) {
    const unsigned u = __float_as_uint(s + 0.0f);                      // -0 -> +0: equal scores must give equal high words
    const unsigned hi = u ^ ((unsigned)((int)u >> 31) | 0x80000000u);  // negative: ~u, else u | sign bit (order-preserving)
    const unsigned lo20 = 0xFFFFFu - (unsigned)j;                      // j < 2^20
    return __hiloint2double((int)(0x43300000u | (hi >> 12)), (int)((hi << 20) | lo20));
}
__device__ __forceinline__ double knn_key_worst() {                    // score -inf, below every real index
    const unsigned hi = ~0xFF800000u;
    return __hiloint2double((int)(0x43300000u | (hi >> 12)), (int)(hi << 20));
}
__device__ __forceinline__ int knn_key_index(double key) { return (int)(0xFFFFFu - ((unsigned)__double2loint(key) & 0xFFFFFu)); }
__device__ __forceinline__ float knn_key_score(double key) {
    const unsigned hi = (((unsigned)__double2hiint(key) & 0xFFFFFu) << 12) | ((unsigned)__double2loint(key) >> 20);
    return __uint_as_float((hi & 0x80000000u) ? (hi ^ 0x80000000u) : ~hi);
}
__device__ __forceinline__ void knn_ce(double& a, double& b) {        // a = better (larger), b = worse
    double mx, mn;                                                     // (two statements: a half nobody reads is dropped)
    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
    a = mx;
    b = mn;
}
__device__ __forceinline__ double knn_max(double a, double b) {
    double mx;
    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
    return mx;
}
#define KNN_CE(a, b) knn_ce((a), (b));

// k > KK (k = 33..40 on the 32-wide lists): a query's k nearest are merged from 8 sub-lists of 32, which is exact unless one
// sub-list would have contributed MORE than its 32 entries -- then (and only then) the merge exhausts that sub-list, the
// workgroup raises its tile's flag and the queued kernel below recomputes the tile exactly (40 nearest with 33+ of them in
// one of 8 interleaved candidate subsets: never observed, but not excluded, so it is detected rather than assumed).
template <int KSTEPS, int KK>
__global__ __launch_bounds__(MW * 64) void knn_mfma_sort_kernel(const float* __restrict__ x, const float* __restrict__ sq, int D,
                                                                int N, int k, int64_t* __restrict__ idx, int* __restrict__ redo) {
    static_assert(KK <= 32, "the merge network is 32 wide");
    __shared__ double lkey[MQ][2 * MW][KK];
    const int tiles_q = (N + MQ - 1) / MQ;
    const int b = blockIdx.x / tiles_q;
    const int q0 = (blockIdx.x % tiles_q) * MQ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    const float* xb = x + (int64_t)b * D * N;
    const float* sqb = sq + (int64_t)b * N;
    const int qi = q0 + lr;
    const bool qok = qi < N;
    float bq[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        const int d = 2 * kk + lk;
        bq[kk] = (qok && d < D) ? xb[(int64_t)d * N + qi] : 0.f;
    }
    const float sq_i = qok ? sqb[qi] : 0.f;
    const double WORST = knn_key_worst();
    double lst[32];                                        // sorted, best first; entries >= KK stay WORST
#pragma unroll
    for (int m = 0; m < 32; ++m) lst[m] = WORST;

    const int ntile = (N + 31) / 32;
    for (int t = w; t < ntile; t += MW) {
        const int c0 = t * 32;
        const int ci = c0 + lr;
        const bool cok = ci < N;
        float af[KSTEPS];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int d = 2 * kk + lk;
            af[kk] = (cok && d < D) ? xb[(int64_t)d * N + ci] : 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bq[kk], acc, 0, 0, 0);
        double key[16];
        double best = WORST;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = c0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float sq_j = j < N ? sqb[j] : 0.f;
            const float inner = -2.0f * acc[r];
            const float tt = (-sq_j) - inner;
            const float sc = j < N ? tt - sq_i : -INFINITY;
            key[r] = j < N ? knn_key(sc, j) : WORST;
            best = knn_max(key[r], best);
        }
        if (!__any(best > lst[KK - 1])) continue;                      // nothing in this tile reaches any list of the wave
        // bitonic sort of the 16 keys, best first
#pragma unroll
        for (int kb = 2; kb <= 16; kb <<= 1)
#pragma unroll
            for (int jb = kb >> 1; jb > 0; jb >>= 1)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int l = i ^ jb;
                    if (l > i) {
                        if ((i & kb) == 0) { KNN_CE(key[i], key[l]) } else { KNN_CE(key[l], key[i]) }
                    }
                }
        // the 32 best of list (32, padded) and tile (16, padded to 32): C[i] = max(A[i], B[31 - i]) is bitonic
#pragma unroll
        for (int i = 16; i < 32; ++i) {
            lst[i] = knn_max(key[31 - i], lst[i]);
        }
        // bitonic merge, best first
#pragma unroll
        for (int jb = 16; jb > 0; jb >>= 1)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int l = i ^ jb;
                if (l > i) { KNN_CE(lst[i], lst[l]) }
            }
#pragma unroll
        for (int m = KK; m < 32; ++m) lst[m] = WORST;                  // only the KK best survive a tile
    }
#pragma unroll
    for (int m = 0; m < KK; ++m) lkey[lr][2 * w + lk][m] = lst[m];
    __syncthreads();
    if (threadIdx.x < MQ && q0 + (int)threadIdx.x < N) {               // 8-way merge of the sorted sub-lists
        const int q = threadIdx.x;
        int head[2 * MW];
#pragma unroll
        for (int u = 0; u < 2 * MW; ++u) head[u] = 0;
        int64_t* o = idx + ((int64_t)b * N + q0 + q) * k;
        for (int m = 0; m < k; ++m) {
            double bestk = 0.0;
            int bu = 0;
#pragma unroll
            for (int u = 0; u < 2 * MW; ++u) {
                const int h = head[u];
                const double v = h < KK ? lkey[q][u][h] : 0.0;
                if (v > bestk) { bestk = v; bu = u; }
            }
#pragma unroll
            for (int u = 0; u < 2 * MW; ++u) head[u] += (u == bu) ? 1 : 0;
            o[m] = (int64_t)knn_key_index(bestk);
        }
        if (redo && k > KK) {
            bool exhausted = false;
#pragma unroll
            for (int u = 0; u < 2 * MW; ++u) exhausted = exhausted || head[u] >= KK;
            if (exhausted) redo[blockIdx.x] = 1;
        }
    }
}

// Queued variant: one wave = 32 queries x ALL candidate tiles, so a query's candidates are split over only
// two lists (the lane halves), each walked in ascending index order.  A score that beats the lane's current k-th best is
// only APPENDED to a per-lane LDS queue (two predicated stores); the ~120-instruction sorted insertion runs in batches,
// when some lane's queue is about to overflow, for as many rounds as the fullest queue holds.  In the kernel above the
// wave-wide `__any` gate is almost always true (8 short lists per query -> ~45 % of the scores enter some list), so
// every score pays the full insertion; here a lane appends ~100 of its 512 scores and the wave runs ~150 insertion
// rounds instead of 2 x 1024.  Thresholds in the queue test may be stale (too low): a superset is queued and the
// insertion itself (strict '>', candidates in ascending order per list) decides, so the result is unchanged.
constexpr int QCAP = 24;       // queue slots per lane: a tile appends at most 16, drained when more than 8 are pending

// NW waves per workgroup share the 32 queries and take interleaved candidate tiles.  Only NW = 1 (all tiles in one wave)
// is used: with NW = 4 the eight short lists per query let ~45 % of the scores through the queue test and the queue
// only adds work (measured 337 vs 157 us for the queue-less 4-wave kernel at B=32, N=1024, k=20).
template <int KSTEPS, int KK, int NW>
__global__ __launch_bounds__(NW * 64) void knn_mfma_queue_kernel(const float* __restrict__ x, const float* __restrict__ sq, int D,
                                                                 int N, int k, int64_t* __restrict__ idx,
                                                                 const int* __restrict__ only_flagged) {
    if (only_flagged && !only_flagged[blockIdx.x]) return;        // repair pass behind the sorting-network kernel (k > 32)
    __shared__ float qv_[NW][QCAP][64];
    __shared__ int qi_[NW][QCAP][64];
    __shared__ float lval[MQ][2 * NW][KK];
    __shared__ int lidx[MQ][2 * NW][KK];
    const int tiles_q = (N + MQ - 1) / MQ;
    const int b = blockIdx.x / tiles_q;
    const int q0 = (blockIdx.x % tiles_q) * MQ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float (*qv)[64] = qv_[w];
    int (*qi)[64] = qi_[w];
    const int lr = lane & 31, lk = lane >> 5;
    const float* xb = x + (int64_t)b * D * N;
    const float* sqb = sq + (int64_t)b * N;

    const int qidx = q0 + lr;
    const bool qok = qidx < N;
    float bq[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        const int d = 2 * kk + lk;
        bq[kk] = (qok && d < D) ? xb[(int64_t)d * N + qidx] : 0.f;
    }
    const float sq_i = qok ? sqb[qidx] : 0.f;

    float bv[KK];
    int bi[KK];
#pragma unroll
    for (int m = 0; m < KK; ++m) {
        bv[m] = -INFINITY;
        bi[m] = 0x7fffffff;
    }
    int cnt = 0;
    auto drain = [&]() {
        for (int s_ = 0; __any(s_ < cnt); ++s_) {
            const bool has = s_ < cnt;
            const float v = has ? qv[s_][lane] : -INFINITY;
            const int jj = has ? qi[s_][lane] : 0x7fffffff;
            topk_insert<KK>(bv, bi, v, jj);
        }
        cnt = 0;
    };

    const int ntile = (N + 31) / 32;
    float af[KSTEPS];
    {
        const int ci = w * 32 + lr;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int d = 2 * kk + lk;
            af[kk] = (ci < N && d < D) ? xb[(int64_t)d * N + ci] : 0.f;
        }
    }
    for (int t = w; t < ntile; t += NW) {
        const int c0 = t * 32;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bq[kk], acc, 0, 0, 0);
        if (t + NW < ntile) {                      // next tile's candidate operand is in flight during the selection
            const int ci = c0 + 32 * NW + lr;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 2 * kk + lk;
                af[kk] = (ci < N && d < D) ? xb[(int64_t)d * N + ci] : 0.f;
            }
        }
        const float thr = bv[KK - 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = c0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float sq_j = j < N ? sqb[j] : 0.f;
            const float inner = -2.0f * acc[r];
            const float tt = (-sq_j) - inner;
            const float sc = j < N ? tt - sq_i : -INFINITY;
            if (sc > thr) {
                qv[cnt][lane] = sc;
                qi[cnt][lane] = j;
                ++cnt;
            }
        }
        if (__any(cnt > QCAP - 16)) drain();
    }
    drain();
#pragma unroll
    for (int m = 0; m < KK; ++m) {
        lval[lr][2 * w + lk][m] = bv[m];
        lidx[lr][2 * w + lk][m] = bi[m];
    }
    __syncthreads();
    // merge of the 2*NW sorted sub-lists (score desc, index asc): one thread per query
    if (threadIdx.x < MQ && q0 + (int)threadIdx.x < N) {
        const int q = threadIdx.x;
        int head[2 * NW];
#pragma unroll
        for (int u = 0; u < 2 * NW; ++u) head[u] = 0;
        int64_t* o = idx + ((int64_t)b * N + q0 + q) * k;
        for (int m = 0; m < k; ++m) {
            float best = -INFINITY;
            int bidx = 0x7fffffff, bu = 0;
#pragma unroll
            for (int u = 0; u < 2 * NW; ++u) {
                const int h = head[u];
                const float v = h < KK ? lval[q][u][h] : -INFINITY;
                const int i2 = h < KK ? lidx[q][u][h] : 0x7fffffff;
                if (v > best || (v == best && i2 < bidx)) {
                    best = v;
                    bidx = i2;
                    bu = u;
                }
            }
#pragma unroll
            for (int u = 0; u < 2 * NW; ++u) head[u] += (u == bu) ? 1 : 0;
            o[m] = bidx;
        }
    }
}

// Batch-queue variant (k <= 32): the queued kernel's layout (one wave = 32 queries x ALL candidate tiles, two lists per query:
// the lane halves) with the sorting-network kernel's list handling.  A score that can still enter the query's k best is only
// APPENDED (score bits + index) to a per-lane LDS ring; when some lane holds more than 16 pending keys the wave sorts 16 per
// lane with the bitonic network and merges them into the 32-wide sorted lists in one pass (~950 instructions per 16 keys per
// lane, against 16 serial insertions of ~120 in the queued kernel, and against one network pass per TILE in the 4-wave
// kernel, whose 8 short lists per query let nearly every tile through).  The admission threshold is the better of the two
// lists' k-th keys of the query (exchanged between lane l and l+32 after every merge): a key below the OTHER list's k-th key
// has k better keys ahead of it there, so it cannot be among the query's k best -- exact in the (score, index) order, ties
// included.  Thresholds are stale between merges (a superset is queued); the merge itself decides.  Expected per list of 512
// candidates: ~60-100 appended keys, ~8-12 merges per wave.
constexpr int BQ_RING = 32;    // pending slots per lane: a tile appends at most 16, a merge leaves at most 16
// NW = 2: two waves share a query tile and take alternate candidate tiles (four lists per query) -- for launches with fewer
// than two waves per SIMD (B=32, N=1024: a D=63 wave is a chain of 32 x 32 dependent MFMAs per tile) and for k = 33..40 on
// 32-wide lists, where the final merge flags a tile whenever one list would have had to supply more than its 32 entries
// (`redo`, repaired by the queued kernel exactly as behind the sorting-network kernel).
template <int KSTEPS, int KK, int NW>
__global__ __launch_bounds__(64 * NW) void knn_mfma_bq_kernel(const float* __restrict__ x, const float* __restrict__ sq, int D, int N,
                                                             int k, int64_t* __restrict__ idx, int* __restrict__ redo) {
    static_assert(KK <= 32, "the merge network is 32 wide");
    __shared__ int2 ring[NW][BQ_RING][64];                 // (score bits, index); 16 KB per wave, reused for the final merge
    const int tiles_q = (N + MQ - 1) / MQ;
    const int b = blockIdx.x / tiles_q;
    const int q0 = (blockIdx.x % tiles_q) * MQ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 31, lk = lane >> 5;
    const float* xb = x + (int64_t)b * D * N;
    const float* sqb = sq + (int64_t)b * N;
    const int qi = q0 + lr;
    const bool qok = qi < N;
    float bq[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
        const int d = 2 * kk + lk;
        bq[kk] = (qok && d < D) ? xb[(int64_t)d * N + qi] : 0.f;
    }
    const float sq_i = qok ? sqb[qi] : 0.f;
    const double WORST = knn_key_worst();
    double lst[32];                                        // sorted, best first; entries >= KK stay WORST
#pragma unroll
    for (int m = 0; m < 32; ++m) lst[m] = WORST;
    float thr_f = -INFINITY;                                           // admission threshold (inclusive: a superset is queued)
    // pending entries of this lane: a STACK, addressed by ONE running byte offset wofs = cnt * 512 + lane * 8 into the ring
    // -- a push is two stores at wofs, wofs + 4 (one ds_write2_b32, any two registers) and one add; the count is wofs >> 9
    // where it is needed
    char* ringb = reinterpret_cast<char*>(&ring[w][0][0]);
    int wofs = lane * 8;
#pragma unroll
    for (int r = 0; r < BQ_RING; ++r) *reinterpret_cast<int*>(ringb + wofs + r * 512) = (int)0xFF800000u;   // every slot: score -inf

    auto merge16 = [&]() {
        // the newest (up to) 16 pending entries: popping from the top needs no wrap-around and no compaction.  Slots above a
        // lane's top always hold a score of -inf (set at start, restored after every pop), so a lane with fewer than 16 pending
        // entries pads its batch with keys below every real one without any per-key select.
        const int cnt = wofs >> 9;
        const int base = max(cnt - 16, 0);
        double key[16];
        int2 raw[16];
        const int rofs = base * 512 + lane * 8;
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[r] = *reinterpret_cast<const int2*>(ringb + rofs + r * 512);   // 16 reads in flight
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(raw[r].x), "+v"(raw[r].y));     // (not sunk below the stores)
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<int*>(ringb + rofs + r * 512) = (int)0xFF800000u;  // popped: -inf again
#pragma unroll
        for (int r = 0; r < 16; ++r) key[r] = knn_key(__int_as_float(raw[r].x), raw[r].y & 0xFFFFF);
        wofs = base * 512 + lane * 8;
#pragma unroll
        for (int kb = 2; kb <= 16; kb <<= 1)
#pragma unroll
            for (int jb = kb >> 1; jb > 0; jb >>= 1)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int l = i ^ jb;
                    if (l > i) {
                        if ((i & kb) == 0) { KNN_CE(key[i], key[l]) } else { KNN_CE(key[l], key[i]) }
                    }
                }
#pragma unroll
        for (int i = 16; i < 32; ++i) {
            lst[i] = knn_max(key[31 - i], lst[i]);
        }
#pragma unroll
        for (int jb = 16; jb > 0; jb >>= 1)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int l = i ^ jb;
                if (l > i) { KNN_CE(lst[i], lst[l]) }
            }
#pragma unroll
        for (int m = KK; m < 32; ++m) lst[m] = WORST;
        // Admission threshold of the query: any key with KK keys ahead of it in the two lists together is out.  Valid lower
        // bounds of the union's KK-th key: either list's own KK-th key, and min(A[i-1], B[j-1]) for i + j = KK (i keys of
        // one list and j of the other are ahead).  With i ~ j ~ KK/2 the bound is tight from the first merge on (each
        // list holds 16 keys then), where the lists' own KK-th keys do not exist yet: 104 -> 77 queued keys per lane.
        constexpr int HI = (KK + 1) / 2, HJ = KK - HI;
        auto partner = [&](double v) {
            return __hiloint2double(__shfl_xor(__double2hiint(v), 32), __shfl_xor(__double2loint(v), 32));
        };
        const double p_last = partner(lst[KK - 1]), p_i = partner(lst[HI - 1]), p_j = HI == HJ ? p_i : partner(lst[HJ - 1]);
        double b1, b2;
        asm("v_min_f64 %0, %1, %2" : "=v"(b1) : "v"(lst[HI - 1]), "v"(p_j));
        asm("v_min_f64 %0, %1, %2" : "=v"(b2) : "v"(lst[HJ - 1]), "v"(p_i));
        // k > KK (k = 33..40 on 32-wide lists): only bounds whose use the final merge can DETECT -- a key rejected below a
        // list's own 32nd entry that belonged to the k best forces that list to run dry there (`redo`, repaired exactly);
        // the cross-list bound has no such witness (16 + 16 better keys exhaust neither list while the key may rank 33..40)
        double thr_key = knn_max(lst[KK - 1], p_last);
        if (k <= KK) thr_key = knn_max(thr_key, knn_max(b1, b2));
        thr_f = knn_key_score(thr_key);
    };

    const int ntile = (N + 31) / 32;
    const bool sq_vec = (N % 4 == 0) && ((((uintptr_t)sq) & 15u) == 0);
    float af[KSTEPS];
    {
        const int ci = w * 32 + lr;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int d = 2 * kk + lk;
            af[kk] = (ci < N && d < D) ? xb[(int64_t)d * N + ci] : 0.f;
        }
    }
    // |x_j|^2 of a tile's 16 candidate rows of this lane, requested one tile ahead (a dependent L2 round trip per tile
    // otherwise: the scan cannot start before it)
    float sqj[16], sqn[16];
    auto load_sq = [&](int c0, float (&dst)[16]) {
        if (c0 + 32 <= N && sq_vec) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(sqb + c0 + 8 * g + 4 * lk);
                dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = c0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                dst[r] = j < N ? sqb[j] : 0.f;
            }
        }
    };
    load_sq(w * 32, sqj);
#pragma unroll
    for (int r = 0; r < 16; ++r) sqn[r] = 0.f;
    for (int t = w; t < ntile; t += NW) {
        const int c0 = t * 32;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bq[kk], acc, 0, 0, 0);
        if (t + NW < ntile) {                      // next tile's candidate operand is in flight during the selection
            const int ci = c0 + 32 * NW + lr;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int d = 2 * kk + lk;
                af[kk] = (ci < N && d < D) ? xb[(int64_t)d * N + ci] : 0.f;
            }
        }
        const bool full = c0 + 32 <= N;
        if (t + NW < ntile) load_sq(c0 + 32 * NW, sqn);  // ... and its |x|^2 (the tile's own arrived during the previous tile)
        auto scan = [&](auto whole) {              // whole tile inside the cloud: no per-candidate bound test
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // two scores per packed instruction (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, same roundings)
                const f32x2 a2 = {acc[r], acc[r + 1]}, s2 = {sqj[r], sqj[r + 1]};
                const f32x2 inner = a2 * -2.0f;
                const f32x2 tt = (-s2) - inner;
                const f32x2 sc2 = tt - sq_i;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = c0 + ((r + h) & 3) + 8 * ((r + h) >> 2) + 4 * lk;
                    const float sc = h ? sc2.y : sc2.x;
                    if ((decltype(whole)::value || j < N) && sc >= thr_f) {
                        *reinterpret_cast<int*>(ringb + wofs) = __float_as_int(sc);
                        *reinterpret_cast<int*>(ringb + wofs + 4) = j;
                        wofs += 512;
                    }
                }
            }
        };
        if (full) scan(std::true_type{});
        else scan(std::false_type{});
        if (__any(wofs >= 17 * 512)) merge16();    // some lane holds > 16; afterwards every lane holds <= 16: room for 16 more
#pragma unroll
        for (int r = 0; r < 16; ++r) sqj[r] = sqn[r];
    }
    while (__any(wofs >= 512)) merge16();
    // 2 * NW sorted lists per query (lanes l, l + 32 of every wave) -> LDS -> one thread per query merges
    if (NW > 1) __syncthreads();                                       // every wave is done with its pending stack
    double (*lkey)[2 * NW][32] = reinterpret_cast<double (*)[2 * NW][32]>(&ring[0][0][0]);
#pragma unroll
    for (int m = 0; m < KK; ++m) lkey[lr][2 * w + lk][m] = lst[m];
    __syncthreads();
    if (threadIdx.x < MQ && q0 + (int)threadIdx.x < N) {
        const int q = threadIdx.x;
        int head[2 * NW];
#pragma unroll
        for (int u = 0; u < 2 * NW; ++u) head[u] = 0;
        int64_t* o = idx + ((int64_t)b * N + q0 + q) * k;
        for (int m = 0; m < k; ++m) {
            double bestk = 0.0;
            int bu = 0;
#pragma unroll
            for (int u = 0; u < 2 * NW; ++u) {
                const int h = head[u];
                const double v = h < KK ? lkey[q][u][h] : 0.0;
                if (v > bestk) { bestk = v; bu = u; }
            }
#pragma unroll
            for (int u = 0; u < 2 * NW; ++u) head[u] += (u == bu) ? 1 : 0;
            o[m] = (int64_t)knn_key_index(bestk);
        }
        if (redo && k > KK) {                                          // a list that ran dry may have owed more than KK entries
            bool exhausted = false;
#pragma unroll
            for (int u = 0; u < 2 * NW; ++u) exhausted = exhausted || head[u] >= KK;
            if (exhausted) redo[blockIdx.x] = 1;
        }
    }
}

template <int KSTEPS>
int launch_knn_mfma(const float* x, const float* sq, int B, int D, int N, int k, int64_t* idx, int* redo, hipStream_t st) {
    const int tiles_q = (N + MQ - 1) / MQ;
    const dim3 grid((unsigned)(B * tiles_q));
    // Measured (MI355X, us): k = 20, N = 1024     B=32 D=3   B=32 D=63   B=512 D=3   B=512 D=63
    //   sorting-network kernel (4 waves)            96        168         1118        2324
    //   per-score insertion kernel (4 waves)       156        243         1729        3500
    //   queued single-wave kernel                  167        229         1508        2274
    // and for N = 2048, k = 40 (lists longer than the 32-wide merge network): queued 390-500 us, insertion 817-894 us;
    // r02: sorting network on 32-wide lists + flagged exact repair (see knn_mfma_sort_kernel): measured below.
    // So: k <= 32 -> sorting network; 32 < k <= 40 -> sorting network with 32-wide lists + repair pass of the queued kernel.
    const int force = hpcs_tuning().knn_kernel;      // 0 auto | 1 tiled (sorting network) | 2 queue | 4 bq | 5 bq2 (A/B tests)
    const bool tiled = force == 1 || (force == 0 && k <= 32);
    // k <= 32: the batch-queue kernel (one wave per 32 queries) once there is a wave per SIMD, the 4-wave sorting-network
    // kernel for small batches (B=8, N=500 D=63: 31 vs 50 us; B=32, N=1024: 57 vs 40 (D=3), 129 vs 95 (D=63) with two waves per
    // query tile; B=4096: 4935 vs 2585 with one)
    const bool bq = force == 4 || force == 5 || (force == 0 && k <= 32 && (long)grid.x >= 1024);
    // two waves per query tile while the launch has fewer than two one-wave workgroups per SIMD (HPCS_KNN_KERNEL=bq2 forces it)
    const bool bq2 = force == 5 || (force == 0 && (long)grid.x < 2048);
    if (k > 32 && redo && (force == 0 || force == 1 || force == 5)) {
        if (const int zr = hpcs_zero_async(st, redo, sizeof(int) * grid.x)) return zr;
        // B=16, N=2048, k=40: two-wave batch-queue 78 us vs sorting network 154 us at D=3; 229 vs 233 us at D=63 (kept on the network)
        if (force == 5 || (force == 0 && KSTEPS == 2))
            hipLaunchKernelGGL((knn_mfma_bq_kernel<KSTEPS, 32, 2>), grid, dim3(128), 0, st, x, sq, D, N, k, idx, redo);
        else
            hipLaunchKernelGGL((knn_mfma_sort_kernel<KSTEPS, 32>), grid, dim3(MW * 64), 0, st, x, sq, D, N, k, idx, redo);
        hipLaunchKernelGGL((knn_mfma_queue_kernel<KSTEPS, 40, 1>), grid, dim3(64), 0, st, x, sq, D, N, k, idx, (const int*)redo);
        return HPCS_OK;
    }
#define HPCS_KNNM_CASE(KK)                                                                                   \
    if (k <= KK) {                                                                                            \
        if (bq && KK <= 32 && bq2)                                                                            \
            hipLaunchKernelGGL((knn_mfma_bq_kernel<KSTEPS, (KK <= 32 ? KK : 32), 2>), grid, dim3(128), 0, st, x, sq, D, N, k, idx, \
                               (int*)nullptr);                                                                \
        else if (bq && KK <= 32)                                                                              \
            hipLaunchKernelGGL((knn_mfma_bq_kernel<KSTEPS, (KK <= 32 ? KK : 32), 1>), grid, dim3(64), 0, st, x, sq, D, N, k, idx, \
                               (int*)nullptr);                                                                \
        else if (tiled && KK <= 32)                                                                           \
            hipLaunchKernelGGL((knn_mfma_sort_kernel<KSTEPS, (KK <= 32 ? KK : 32)>), grid, dim3(MW * 64), 0, st, x, sq, D, N, k, \
                               idx, (int*)nullptr);                                                           \
        else                                                                                                  \
            hipLaunchKernelGGL((knn_mfma_queue_kernel<KSTEPS, KK, 1>), grid, dim3(64), 0, st, x, sq, D, N, k, idx, \
                               (const int*)nullptr);                                                          \
        return HPCS_OK;                                                                                       \
    }
    HPCS_KNNM_CASE(4)
    HPCS_KNNM_CASE(8)
    HPCS_KNNM_CASE(10)
    HPCS_KNNM_CASE(16)
    HPCS_KNNM_CASE(20)
    HPCS_KNNM_CASE(32)
    HPCS_KNNM_CASE(40)
#undef HPCS_KNNM_CASE
    return HPCS_ERR_UNSUPPORTED;
}

template <int DT, int TC>
int launch_knn(const float* x, int B, int D, int N, int k, int64_t* idx, hipStream_t st) {
    const int tiles_q = (N + QT - 1) / QT;
    const dim3 grid((unsigned)(B * tiles_q));
    const size_t lds = ((size_t)(D + 1) * TC + (DT == 0 ? (size_t)D * QT : 0)) * sizeof(float);
    if (lds > 160 * 1024) return HPCS_ERR_UNSUPPORTED;
#define HPCS_KNN_CASE(KK)                                                                       \
    if (k <= KK) {                                                                              \
        if (lds > 64 * 1024)                                                                    \
            (void)hipFuncSetAttribute((const void*)knn_kernel<DT, KK, TC>,                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL((knn_kernel<DT, KK, TC>), grid, dim3(QT), lds, st, x, D, N, k, idx);     \
        return HPCS_OK;                                                                         \
    }
    HPCS_KNN_CASE(4)
    HPCS_KNN_CASE(8)
    HPCS_KNN_CASE(10)
    HPCS_KNN_CASE(16)
    HPCS_KNN_CASE(20)
    HPCS_KNN_CASE(32)
    HPCS_KNN_CASE(40)
    HPCS_KNN_CASE(64)
#undef HPCS_KNN_CASE
    return HPCS_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" size_t hpcs_knn_workspace(int B, int N) {
    if (B <= 0 || N <= 0) return 64;
    // |x|^2 per point + one repair flag per 32-query tile (k > 32)
    return (size_t)B * N * sizeof(float) + (size_t)B * ((N + MQ - 1) / MQ) * sizeof(int);
}

extern "C" int hpcs_knn(const float* x, int B, int D, int N, int k, int64_t* idx, void* ws, size_t ws_bytes,
                        hpcs_stream_t stream) {
    if (!x || !idx) return HPCS_ERR_NULL;
    if (B < 0 || D <= 0 || N <= 0 || k <= 0 || k > N) return HPCS_ERR_SHAPE;
    if (k > 64) return HPCS_ERR_UNSUPPORTED;
    if (B == 0) return HPCS_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((D == 3 || D == 63) && k <= 40 && ws && N <= (1 << 20)) {      // (the double-held keys carry 20 index bits)
        if (ws_bytes < hpcs_knn_workspace(B, N)) return HPCS_ERR_WORKSPACE;
        float* sq = (float*)ws;
        const int64_t total = (int64_t)B * N;
        hipLaunchKernelGGL(knn_sumsq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, D, N, total, sq);
        HPCS_CHECK_LAUNCH();
        int* redo = reinterpret_cast<int*>(sq + (size_t)B * N);
        rc = D == 3 ? launch_knn_mfma<2>(x, sq, B, D, N, k, idx, redo, st) : launch_knn_mfma<32>(x, sq, B, D, N, k, idx, redo, st);
    } else if (D == 3) {
        rc = launch_knn<3, 256>(x, B, D, N, k, idx, st);
    } else if (D == 63) {
        rc = launch_knn<63, 128>(x, B, D, N, k, idx, st);
    } else {
        rc = launch_knn<0, 128>(x, B, D, N, k, idx, st);
    }
    if (rc != HPCS_OK) return rc;
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
