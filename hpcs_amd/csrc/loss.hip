// a8-a14: class-balanced triplet sampling, margin filter with stable compaction, matrix-free HypHC loss,
// CosFace and triplet-margin losses -- forward and backward.
//
// Replaces get_balanced_random_triplet_indices (hpcs/miner/loss_and_miner_utils.py:7-75),
// RandomTripletMarginMiner.mine (hpcs/miner/triplet_margin_miner.py:13-38),
// MetricHyperbolicLoss.compute_hyp (hpcs/loss/ultrametric_loss.py:57-93), PML CosFaceLoss (algorithm
// mirrored in hpcs/loss/hierarchical_cosface_loss.py:57-87) and TripletMarginLoss
// (hpcs/miner/triplet_margin_loss.py:34-65).
//
// Data layout: xn [M,E] = L2-normalised embedding rows (4 MB at M=32768,E=32: L2-resident); triplets are three
// int64 index arrays.  The reference's two (B*N)^2 cosine matrices (4.3 GB each) are never built: every
// similarity the path consumes is a gathered row dot, and HypHC's LCA distance for two points of equal norm s
// depends only on that similarity:  d = 2 artanh(m), m = 1/(C+sqrt(C^2-1)), C = (1+s^2)/(2 s sqrt(w)), w=(1+cos)/2.
// 8 lanes cooperate on one triplet (a 32-float row = one 128-byte segment per gather).
#include "common.h"

namespace {

constexpr int SUB = 8;
constexpr int TPB = 256;
constexpr int TRIP_PER_PASS = TPB / SUB;   // 32
constexpr int CHUNK = 2048;                // triplets per workgroup in the filter

__device__ __forceinline__ float sub_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// ---- 8-lane group data movement on the VALU (DPP): no LDS round trip, unlike __shfl's ds_bpermute ----------------
// All of these must run with every lane of the wave active (uniform control flow).
template <int C>   // value held by lane C (0..3) of each 8-lane group -> all 8 lanes of the group
__device__ __forceinline__ int grp_bcast(int x) {
    const int t = __builtin_amdgcn_update_dpp(x, x, C | (C << 2) | (C << 4) | (C << 6), 0xF, 0xF, false);   // quad_perm
    return __builtin_amdgcn_update_dpp(t, t, 0x114, 0xF, 0xA, false);        // row_shr:4 into the upper quads only
}
template <int C>
__device__ __forceinline__ float grp_bcast(float x) { return __int_as_float(grp_bcast<C>(__float_as_int(x))); }
// the other quad's value of the same position: lane s <- lane s ^ 4 (row_shr:4 into the upper quads, row_shl:4 into the lower)
__device__ __forceinline__ float grp_other_half(float x) {
    const int v = __float_as_int(x);
    int t = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xA, false);        // upper quads <- lane - 4
    t = __builtin_amdgcn_update_dpp(t, v, 0x104, 0xF, 0x5, false);            // lower quads <- lane + 4
    return __int_as_float(t);
}
// lane s <- lane s+1 (the group's last lane receives its neighbour group's first value or keeps its own: unused)
__device__ __forceinline__ int grp_shift_down(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x101, 0xF, 0xF, false); }
// sum over the 8 lanes of a group, result in all of them: xor-1, xor-2 inside quads, then the mirrored other quad
__device__ __forceinline__ float grp_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t bounded(uint64_t r, uint64_t n) { return __umul64hi(r, n); }
// 32-bit counter hash (two murmur3 finalizer rounds over a 64-bit counter): cheap on the quarter-rate integer multiplier
__device__ __forceinline__ unsigned hash32(unsigned a, unsigned b) {
    unsigned h = a * 0xCC9E2D51u + (b ^ 0x1B873593u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += b * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}

// ---------------------------------------------------------------------------------------------- class index
__global__ void hist_kernel(const int64_t* __restrict__ labels, int M, int C, int32_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t l = labels[i];
    if (l >= 0 && l < C) atomicAdd(counts + l, 1);
}
__global__ void offsets_kernel(const int32_t* __restrict__ counts, int C, int32_t* __restrict__ offsets) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < C; ++c) { offsets[c] = acc; acc += counts[c]; }
        offsets[C] = acc;
    }
}
// one workgroup per class: members in ascending index order (stable)
__global__ void class_fill_kernel(const int64_t* __restrict__ labels, int M, const int32_t* __restrict__ offsets,
                                  int32_t* __restrict__ order) {
    __shared__ int wave_cnt[TPB / 64];
    __shared__ int base_sh;
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_sh = offsets[c];
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += TPB) {
        const int i = i0 + threadIdx.x;
        const bool mine = i < M && labels[i] == c;
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) wave_cnt[w] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int j = 0; j < w; ++j) before += wave_cnt[j];
        int tot = 0;
        for (int j = 0; j < TPB / 64; ++j) tot += wave_cnt[j];
        const int base = base_sh;
        if (mine) order[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) base_sh = base + tot;
        __syncthreads();
    }
}

// Blocked form for C <= 512 (three short launches instead of global atomics on C counters and C workgroups that each
// scan all M labels): (1) per-1024-label-block class histograms in LDS, (2) one workgroup turns them into the class
// offsets and each block's first slot per class, (3) every block places its members: in-wave ranks by ballot over
// the classes present in the wave, the 16 waves ordered through an LDS prefix.  Stable (ascending index per class).
constexpr int CI_TPB = 1024;
constexpr int CI_MAXC = 512;
__global__ __launch_bounds__(CI_TPB) void class_hist_blocks_kernel(const int64_t* __restrict__ labels, int M, int C,
                                                                    int32_t* __restrict__ blk_cnt) {
    __shared__ int h[CI_MAXC];
    for (int c = threadIdx.x; c < C; c += CI_TPB) h[c] = 0;
    __syncthreads();
    const int i = blockIdx.x * CI_TPB + threadIdx.x;
    if (i < M) {
        const int64_t l = labels[i];
        if (l >= 0 && l < C) atomicAdd(&h[(int)l], 1);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += CI_TPB) blk_cnt[(int64_t)blockIdx.x * C + c] = h[c];
}
__global__ __launch_bounds__(CI_MAXC) void class_scan_blocks_kernel(int32_t* __restrict__ blk_cnt, int nblk, int C,
                                                                    int32_t* __restrict__ offsets) {
    __shared__ int tot[CI_MAXC + 1];
    const int c = threadIdx.x;
    if (c < C) {
        int t = 0;
        for (int b = 0; b < nblk; ++b) t += blk_cnt[(int64_t)b * C + c];
        tot[c] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int q = 0; q < C; ++q) { const int t = tot[q]; tot[q] = acc; offsets[q] = acc; acc += t; }
        offsets[C] = acc;
    }
    __syncthreads();
    if (c < C) {
        int run = tot[c];
        for (int b = 0; b < nblk; ++b) {
            const int t = blk_cnt[(int64_t)b * C + c];
            blk_cnt[(int64_t)b * C + c] = run;           // becomes: first slot of class c for block b
            run += t;
        }
    }
}
__global__ __launch_bounds__(CI_TPB) void class_fill_blocks_kernel(const int64_t* __restrict__ labels, int M, int C,
                                                                    const int32_t* __restrict__ blk_base,
                                                                    int32_t* __restrict__ order) {
    extern __shared__ int wcnt[];                        // [16][C]
    constexpr int NW = CI_TPB / 64;
    for (int q = threadIdx.x; q < NW * C; q += CI_TPB) wcnt[q] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * CI_TPB + threadIdx.x;
    int cls = -1;
    if (i < M) {
        const int64_t l = labels[i];
        if (l >= 0 && l < C) cls = (int)l;
    }
    int rank = 0;
    unsigned long long todo = __ballot(cls >= 0);
    while (todo) {                                       // one round per distinct class present in the wave
        const int src = __ffsll((long long)todo) - 1;
        const int c0 = __shfl(cls, src, 64);
        const unsigned long long m = __ballot(cls == c0);
        if (cls == c0) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == src) wcnt[w * C + c0] = __popcll(m);
        todo &= ~m;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += CI_TPB) {      // exclusive prefix over the waves, seeded with the block's first slot
        int run = blk_base[(int64_t)blockIdx.x * C + c];
        for (int j = 0; j < NW; ++j) {
            const int t = wcnt[j * C + c];
            wcnt[j * C + c] = run;
            run += t;
        }
    }
    __syncthreads();
    if (cls >= 0) order[wcnt[w * C + cls] + rank] = i;
}

// ---------------------------------------------------------------------------------------------- sampler
__global__ void sample_triplets_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ offsets,
                                       const int64_t* __restrict__ cls_toff, const int32_t* __restrict__ cls_k, int C, int M,
                                       int64_t T, const int64_t* __restrict__ draws_pos, const int64_t* __restrict__ draws_neg,
                                       uint64_t seed, int64_t* __restrict__ a, int64_t* __restrict__ p, int64_t* __restrict__ n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int lo = 0, hi = C;                       // largest c with cls_toff[c] <= t (classes with k=0 have empty ranges)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cls_toff[mid] <= t) lo = mid; else hi = mid;
    }
    while (lo + 1 < C && cls_toff[lo + 1] <= t) ++lo;
    const int c = lo;
    const int k = cls_k[c];
    const int64_t local = t - cls_toff[c];
    const int rank = (int)(local / k);
    const int off = offsets[c];
    const int n_a = offsets[c + 1] - off;
    const int32_t* mem = order + off;
    const int64_t rp = draws_pos ? draws_pos[t] : (int64_t)bounded(mix64(seed ^ mix64(2 * (uint64_t)t)), (uint64_t)(n_a - 1));
    const int64_t rn = draws_neg ? draws_neg[t] : (int64_t)bounded(mix64(seed ^ mix64(2 * (uint64_t)t + 1)), (uint64_t)(M - n_a));
    a[t] = mem[rank];
    p[t] = mem[rp + (rp >= rank ? 1 : 0)];
    // rn-th point NOT in the class, in ascending index order: rn + #{i : mem[i]-i <= rn}
    int l2 = 0, h2 = n_a;
    while (l2 < h2) {
        const int mid = (l2 + h2) >> 1;
        if ((int64_t)mem[mid] - mid <= rn) l2 = mid + 1; else h2 = mid;
    }
    n[t] = rn + l2;
}

// ---------------------------------------------------------------------------------------------- filter
__device__ __forceinline__ bool keep_rule(float tm, float margin, int kind) {
    if (kind == HPCS_TRIPLETS_EASY) return tm > margin;
    bool k = tm <= margin;
    if (kind == HPCS_TRIPLETS_HARD) k = k && (tm <= 0.f);
    else if (kind == HPCS_TRIPLETS_SEMIHARD) k = k && (tm > 0.f);
    return k;
}

__global__ void filter_flag_kernel(const float* __restrict__ xn, int E, const int64_t* __restrict__ a,
                                   const int64_t* __restrict__ p, const int64_t* __restrict__ n, int64_t T, float margin,
                                   int kind, uint8_t* __restrict__ flags, int32_t* __restrict__ block_cnt) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int s = threadIdx.x % SUB, g = threadIdx.x / SUB;
    const int64_t t0 = (int64_t)blockIdx.x * CHUNK;
    int local = 0;
    for (int pass = 0; pass < CHUNK / TRIP_PER_PASS; ++pass) {
        const int64_t t = t0 + (int64_t)pass * TRIP_PER_PASS + g;
        const bool live = t < T;
        const int64_t ia = live ? a[t] : 0, ip = live ? p[t] : 0, in = live ? n[t] : 0;
        float dp = 0.f, dn = 0.f;
        if (E % SUB == 0) {          // same element->lane mapping and summation order as hyphc_kernel's FAST path
            const int epl = E / SUB;
            for (int j = 0; j < epl; ++j) {
                const int e = s * epl + j;
                const float av = xn[ia * E + e];
                dp += av * xn[ip * E + e];
                dn += av * xn[in * E + e];
            }
        } else {
            for (int e = s; e < E; e += SUB) {
                const float av = xn[ia * E + e];
                dp += av * xn[ip * E + e];
                dn += av * xn[in * E + e];
            }
        }
        dp = sub_sum(dp);
        dn = sub_sum(dn);
        const float tm = 0.5f * (1.f + dp) - 0.5f * (1.f + dn);
        const bool keep = live && keep_rule(tm, margin, kind);
        if (live && s == 0) {
            flags[t] = keep ? 1 : 0;
            local += keep ? 1 : 0;
        }
    }
    if (local) atomicAdd(&cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = cnt;
}

__global__ void filter_scan_kernel(const int32_t* __restrict__ block_cnt, int nblocks, int64_t* __restrict__ block_off,
                                   int64_t* __restrict__ count) {
    // one workgroup: each thread owns a contiguous segment, workgroup-wide exclusive scan of the segment sums
    __shared__ int64_t wsum[TPB / 64];
    const int per = (nblocks + TPB - 1) / TPB;
    const int b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
    int64_t mine = 0;
    for (int i = b0; i < b1; ++i) mine += block_cnt[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int64_t before = 0, total = 0;
    for (int j = 0; j < TPB / 64; ++j) {
        if (j < w) before += wsum[j];
        total += wsum[j];
    }
    int64_t run = before + incl - mine;
    for (int i = b0; i < b1; ++i) {
        block_off[i] = run;
        run += block_cnt[i];
    }
    if (threadIdx.x == 0) *count = total;
}

__global__ void filter_scatter_kernel(const uint8_t* __restrict__ flags, const int64_t* __restrict__ block_off,
                                      const int64_t* __restrict__ a, const int64_t* __restrict__ p, const int64_t* __restrict__ n,
                                      int64_t T, int64_t* __restrict__ ao, int64_t* __restrict__ po, int64_t* __restrict__ no) {
    __shared__ int wsum[TPB / 64];
    constexpr int PER = CHUNK / TPB;      // 8 consecutive triplets per thread (keeps order)
    const int64_t t0 = (int64_t)blockIdx.x * CHUNK + (int64_t)threadIdx.x * PER;
    int f[PER], mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        f[i] = (t0 + i < T) ? flags[t0 + i] : 0;
        mine += f[i];
    }
    // inclusive scan of `mine` across the workgroup
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int before = 0;
    for (int j = 0; j < w; ++j) before += wsum[j];
    int64_t dst = block_off[blockIdx.x] + before + incl - mine;
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (f[i]) {
            ao[dst] = a[t0 + i];
            po[dst] = p[t0 + i];
            no[dst] = n[t0 + i];
            ++dst;
        }
}

// ---------------------------------------------------------------------------------------------- HypHC
// d(s, dot) = 2 artanh(m), m = 1/(C + sqrt(C^2-1)), C = (1+s^2)/(2 s sqrt(w)), w = (1+dot)/2, and its partials.
// fp32 with cancellation-free forms: C-1 = ((1-s)^2 + 2 s (1-w)/(1+sqrt w)) / (2 s sqrt w), C^2-1 = (C-1)(C+1).
struct LcaF { float d, dd_dw, dd_ds; };

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32, 1 ulp

// 2*artanh(m) for m in [0, 1-1e-5]: odd series below 0.25 (truncation < 5e-9 relative), log1p above
__device__ __forceinline__ float two_artanh(float m) {
    if (m < 0.25f) {
        const float m2 = m * m;
        const float p = 1.f + m2 * (1.f / 3.f + m2 * (1.f / 5.f + m2 * (1.f / 7.f + m2 * (1.f / 9.f + m2 * (1.f / 11.f)))));
        return 2.f * m * p;
    }
    return log1pf(m) - log1pf(-m);
}

__device__ __forceinline__ LcaF lca_of_dot(float s, float dot) {
    LcaF r;
    const float w = fmaxf(0.5f * (1.f + dot), 1e-12f);
    const float omw = fmaxf(0.5f * (1.f - dot), 0.f);
    const float sq = __builtin_amdgcn_sqrtf(w);
    const float iden = fast_rcp(2.f * s * sq);
    const float Cm1 = ((1.f - s) * (1.f - s) + 2.f * s * omw * fast_rcp(1.f + sq)) * iden;
    const float C = 1.f + Cm1;
    const float root = __builtin_amdgcn_sqrtf(Cm1 * (C + 1.f));
    const float m = fast_rcp(C + root);
    const float mc = fminf(m, 1.f - 1e-5f);                   // Artanh's input clamp
    r.d = two_artanh(mc);
    const float g = -2.f * m * fast_rcp((1.f - mc * mc) * fmaxf(root, 1e-20f));
    r.dd_dw = g * (-0.5f * C * fast_rcp(w));
    r.dd_ds = g * (C * (s * s - 1.f) * fast_rcp(s * (1.f + s * s)));
    return r;
}

constexpr int RUN = 16;          // consecutive triplets walked by one 8-lane group (they mostly share the anchor)

// EPL = ceil(E/8) elements per lane, rows live in registers.
// FAST: E == 8*EPL exactly (E = 16, 32, 64): lane s owns the EPL CONTIGUOUS elements [s*EPL, (s+1)*EPL) so a row is
// gathered with one vector load per lane, and backward adds whole rows per atomic instruction via an LDS transpose.
template <int EPL, bool FAST>
struct RowIO {
    static __device__ __forceinline__ int eidx(int s, int j) { return FAST ? s * EPL + j : s + SUB * j; }
    static __device__ __forceinline__ void load(const float* __restrict__ xn, int E, int row, int s, float (&v)[EPL]) {
        // FAST layouts only: 32-bit element offset (M*E < 2^31), one 16-byte load per 4 elements
        const unsigned o = (unsigned)(row * E + s * EPL);
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            const float4 u = *reinterpret_cast<const float4*>(xn + o + 4 * q);
            v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
        }
        if (EPL == 2) {
            const float2 u = *reinterpret_cast<const float2*>(xn + o);
            v[0] = u.x; v[1] = u.y;
        }
    }
    static __device__ __forceinline__ void load(const float* __restrict__ xn, int E, int64_t row, int s, float (&v)[EPL]) {
        if (FAST && EPL % 4 == 0) {
#pragma unroll
            for (int q = 0; q < EPL / 4; ++q) {
                const float4 u = *reinterpret_cast<const float4*>(xn + row * E + s * EPL + 4 * q);
                v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
            }
        } else if (FAST && EPL == 2) {
            const float2 u = *reinterpret_cast<const float2*>(xn + row * E + s * 2);
            v[0] = u.x; v[1] = u.y;
        } else {
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int e = eidx(s, j);
                v[j] = e < E ? xn[row * E + e] : 0.f;
            }
        }
    }
};

template <int EPL, bool BWD, bool FAST>
__global__ __launch_bounds__(TPB) void hyphc_kernel(const float* __restrict__ xn, int E, const int64_t* __restrict__ a,
                                                    const int64_t* __restrict__ p, const int64_t* __restrict__ n,
                                                    const int64_t* __restrict__ count, int64_t T_cap,
                                                    const float* __restrict__ scale, float temperature, float margin,
                                                    int kind, const int64_t* __restrict__ norm_count,
                                                    unsigned long long* __restrict__ kept_out,
                                                    const float* __restrict__ gloss, double* __restrict__ loss_sum,
                                                    float* __restrict__ gxn, double* __restrict__ gscale, unsigned perm_mul) {
    using IO = RowIO<EPL, FAST>;
    static_assert(RUN == 2 * SUB, "index prefetch assumes two index slots per lane");
    __shared__ double red[16];
    constexpr int LD = 8 * EPL + 8;
    __shared__ __attribute__((aligned(16))) float stage[(BWD && FAST) ? (TPB / 64) * 2 * 8 * LD : 4];
    const int64_t T = *count < T_cap ? *count : T_cap;
    const float sraw = *scale;
    const float sc = fminf(fmaxf(sraw, 1e-4f), 1.f);
    const bool s_live = sraw >= 1e-4f && sraw <= 1.f;         // torch.clamp passes the gradient inside the range only
    const float inv_t = 1.f / temperature;
    // kind >= 0: the miner's margin filter is applied inline (no compaction pass); the mean is over kept triplets
    float up = 0.f;
    if (BWD) {
        const int64_t Tn = norm_count ? *norm_count : T;
        up = Tn > 0 ? (float)((double)(*gloss) / (double)Tn) : 0.f;
    }
    unsigned kept = 0;
    const int lane = threadIdx.x & 63;
    const int s = threadIdx.x % SUB, g = threadIdx.x / SUB;
    const int sub0 = lane & ~(SUB - 1);
    // Triplets are ordered by class; co-resident workgroups would all add into the same few hundred positive rows.
    // A multiplicative permutation of the block id spreads concurrent workgroups over the whole triplet list.
    const unsigned pblock = (unsigned)(((uint64_t)blockIdx.x * perm_mul) % gridDim.x);
    const int64_t base = ((int64_t)pblock * TRIP_PER_PASS + g) * RUN;

    // the group's RUN triplet indices, fetched once (coalesced): lane s holds entries s and s+8
    int64_t ta[2], tp[2], tn[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t t = base + s + SUB * h;
        const bool ok = t < T;
        ta[h] = ok ? a[t] : 0;
        tp[h] = ok ? p[t] : 0;
        tn[h] = ok ? n[t] : 0;
    }
    auto fetch = [&](int i, int64_t& ia, int64_t& ip, int64_t& in, float (&av)[EPL], float (&pv)[EPL], float (&nv)[EPL]) {
        const int src = sub0 + (i & (SUB - 1));
        ia = __shfl(i < SUB ? ta[0] : ta[1], src, 64);
        ip = __shfl(i < SUB ? tp[0] : tp[1], src, 64);
        in = __shfl(i < SUB ? tn[0] : tn[1], src, 64);
        IO::load(xn, E, ia, s, av);
        IO::load(xn, E, ip, s, pv);
        IO::load(xn, E, in, s, nv);
    };

    double acc = 0.0, acc_loss = 0.0;
    float acc_a[EPL];
    int64_t cur = -1;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc_a[j] = 0.f;

    int64_t ia, ip, in, ia2 = 0, ip2 = 0, in2 = 0;
    float av[EPL], pv[EPL], nv[EPL], av2[EPL], pv2[EPL], nv2[EPL];
    fetch(0, ia, ip, in, av, pv, nv);
    for (int i = 0; i < RUN; ++i) {
        if (i + 1 < RUN) fetch(i + 1, ia2, ip2, in2, av2, pv2, nv2);      // next triplet's rows are in flight during the math
        bool live = base + i < T;
        float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            d1 += av[j] * pv[j];
            d2 += av[j] * nv[j];
            d3 += pv[j] * nv[j];
        }
        d1 = sub_sum(d1);
        d2 = sub_sum(d2);
        d3 = sub_sum(d3);
        if (kind >= 0) {
            const float tm = 0.5f * (1.f + d1) - 0.5f * (1.f + d2);
            live = live && keep_rule(tm, margin, kind);
            if (live && s == 0) ++kept;
        }
        // lane s (<3) of the group evaluates similarity s; lanes >= 3 shadow similarity 2
        const float my_dot = s == 0 ? d1 : (s == 1 ? d2 : d3);
        const LcaF L = lca_of_dot(sc, my_dot);
        const float dsc = L.d * inv_t;
        const float q0 = __shfl(dsc, sub0, 64), q1 = __shfl(dsc, sub0 + 1, 64), q2 = __shfl(dsc, sub0 + 2, 64);
        const float mx = fmaxf(q0, fmaxf(q1, q2));
        const float ex = __expf(dsc - mx);
        const float e0 = __shfl(ex, sub0, 64), e1 = __shfl(ex, sub0 + 1, 64), e2 = __shfl(ex, sub0 + 2, 64);
        const float w0 = 0.5f * (1.f + d1), w1 = 0.5f * (1.f + d2), w2 = 0.5f * (1.f + d3);
        const float den = e0 + e1 + e2;
        const float wbar = (e0 * w0 + e1 * w1 + e2 * w2) / den;
        if (!BWD) {
            if (live && s == 0) acc += (double)((w0 + w1 + w2) - wbar);
        } else {
            if (live && s == 0) acc_loss += (double)((w0 + w1 + w2) - wbar);
            const float my_w = s == 0 ? w0 : (s == 1 ? w1 : w2);
            const float sm = ex / den;
            const float gd = -(sm * (my_w - wbar)) * inv_t;                    // d loss_t / d d_s
            const float gw = up * ((1.f - sm) + gd * L.dd_dw);
            if (live && s < 3 && s_live) acc += (double)(up * gd * L.dd_ds);
            const float c0 = 0.5f * __shfl(gw, sub0, 64), c1 = 0.5f * __shfl(gw, sub0 + 1, 64),
                        c2 = 0.5f * __shfl(gw, sub0 + 2, 64);
            if (live) {
                if (ia != cur) {                       // anchor changed: flush the register accumulator
                    if (cur >= 0) {
#pragma unroll
                        for (int j = 0; j < EPL; ++j)
                            if (IO::eidx(s, j) < E) atomicAdd(gxn + cur * E + IO::eidx(s, j), acc_a[j]);
                    }
                    cur = ia;
#pragma unroll
                    for (int j = 0; j < EPL; ++j) acc_a[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < EPL; ++j) acc_a[j] += c0 * pv[j] + c1 * nv[j];
            }
            if (FAST) {
                // positive / negative rows are random: give every atomic wave-instruction whole contiguous rows
                // (64/E rows x E floats = 256 B) by transposing the 8 triplets' updates through a wave-private LDS
                // slab -- the full-rate shape of MI355X_MICROARCH "Global float atomics".
                float* slab = stage + (threadIdx.x >> 6) * (2 * 8 * LD);
                const int gi = (lane >> 3);
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    slab[gi * LD + s * EPL + j] = live ? c0 * av[j] + c2 * nv[j] : 0.f;
                    slab[8 * LD + gi * LD + s * EPL + j] = live ? c1 * av[j] + c2 * pv[j] : 0.f;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the slab is wave-private, no barrier needed
                constexpr int RPI = 64 / (8 * EPL);          // rows per instruction
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    const int tl = r * RPI + lane / (8 * EPL), e = lane % (8 * EPL);
                    const int64_t rp = __shfl(ip, tl * SUB, 64);
                    const int64_t rn = __shfl(in, tl * SUB, 64);
                    const bool lv = __shfl((int)live, tl * SUB, 64) != 0;
                    if (lv) {
                        atomicAdd(gxn + rp * E + e, slab[tl * LD + e]);
                        atomicAdd(gxn + rn * E + e, slab[8 * LD + tl * LD + e]);
                    }
                }
            } else if (live) {
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    const int e = IO::eidx(s, j);
                    if (e < E) {
                        atomicAdd(gxn + ip * E + e, c0 * av[j] + c2 * nv[j]);
                        atomicAdd(gxn + in * E + e, c1 * av[j] + c2 * pv[j]);
                    }
                }
            }
        }
        ia = ia2; ip = ip2; in = in2;
#pragma unroll
        for (int j = 0; j < EPL; ++j) { av[j] = av2[j]; pv[j] = pv2[j]; nv[j] = nv2[j]; }
    }
    if (BWD && cur >= 0) {
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (IO::eidx(s, j) < E) atomicAdd(gxn + cur * E + IO::eidx(s, j), acc_a[j]);
    }
    const double tot = hpcs_block_sum<double>(acc, red);
    if (threadIdx.x == 0 && tot != 0.0) atomicAdd(BWD ? gscale : loss_sum, tot);
    if (BWD && loss_sum) {                 // one-pass mode: the loss itself comes out of the gradient pass
        const double lt = hpcs_block_sum<double>(acc_loss, red);
        if (threadIdx.x == 0 && lt != 0.0) atomicAdd(loss_sum, lt);
    }
    if (kept_out) {
        const double kt = hpcs_block_sum<double>((double)kept, red);
        if (threadIdx.x == 0 && kt != 0.0) atomicAdd(kept_out, (unsigned long long)kt);
    }
}

// ---------------------------------------------------------------------------------------------- HypHC, sampled in place
// Training-path kernel for rng = "device": the class-balanced triplets are GENERATED inside the loss kernel from the
// counter RNG (no 3 x T int64 index arrays, no sampler / filter passes), the miner's margin rule is applied inline,
// and loss, kept count and the unscaled gradient come out of one sweep.
//   triplet t -> class c (cls_toff), anchor rank = (t - toff_c) / k_c, positive = uniform over the class minus the
//   anchor, negative = uniform over the other classes by rejection on the label (identical distribution to the
//   reference's "index into the list of non-members", loss_and_miner_utils.py:61-62).
// Work layout: a 1024-thread workgroup owns TB consecutive triplets (class-major order, so almost always one class);
// each 8-lane group walks SEG consecutive triplets (one anchor for long stretches: its gradient lives in registers).
// Positive rows belong to the anchor's class: their gradients accumulate in an LDS table indexed by the member's
// rank (ds_add_f32) and are flushed once per workgroup -- ~10x fewer global atomics than one per triplet; negative
// rows are random over the cloud and use the row-contiguous global atomic shape.
// Float add into an LDS table, EPL elements at stride SUB.  NOT ds_add_f32: on gfx950 that instruction retires one LANE per
// 3 cycles -- 193 cycles per 64-lane wave-instruction whatever the addresses, against 5.5 for ds_add_u32 and 17 for this
// compare-and-swap loop (scratch microbenchmark, r03: 16 waves per CU, the table's access pattern).  The positive table's
// 8.2 M ds_add_f32 per launch were 2.5 ms of LDS-pipe time per CU in a 2.9 ms kernel (SQ_LDS_IDX_ACTIVE = 73 % of the
// kernel's cycles; "87 % of the wait cycles are LDS waits").  All EPL reads, then all EPL swaps, are in flight together;
// a lane whose swap lost (two groups of the wave-instruction on one row) retries that element only.
template <int EPL>
__device__ __forceinline__ void lds_add_row(float* base, const float (&v)[EPL]) {
    int* ip = reinterpret_cast<int*>(base);
    int cur[EPL], got[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) cur[j] = ip[j * SUB];
#pragma unroll
    for (int j = 0; j < EPL; ++j) got[j] = atomicCAS(ip + j * SUB, cur[j], __float_as_int(__int_as_float(cur[j]) + v[j]));
    bool lost = false;
#pragma unroll
    for (int j = 0; j < EPL; ++j) lost = lost || (got[j] != cur[j]);
    if (__builtin_amdgcn_ballot_w64(lost) != 0ull && lost) {   // rare (< 1 % of the updates): wave-uniform test, retries out of line
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            int seen = got[j];
            if (seen != cur[j]) {
                int expect;
                do {
                    expect = seen;
                    seen = atomicCAS(ip + j * SUB, expect, __float_as_int(__int_as_float(expect) + v[j]));
                } while (seen != expect);
            }
        }
    }
}

constexpr int SB_THREADS = 1024;
constexpr int SB_GROUPS = SB_THREADS / SUB;      // 128
constexpr int SB_SEG_MAX = 256;                  // triplets per 8-lane group: chosen per launch (multiple of 8) so that the
constexpr int SB_SEG_MIN = 16;                   // workgroup count fills whole rounds of one-workgroup-per-CU residency

__device__ __forceinline__ int class_of(const int64_t* __restrict__ toff, int C, int64_t t) {
    int lo = 0, hi = C;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (toff[mid] <= t) lo = mid; else hi = mid;
    }
    while (lo + 1 < C && toff[lo + 1] <= t) ++lo;
    return lo;
}

template <int EPL, bool GRAD>
__global__ __launch_bounds__(SB_THREADS) void hyphc_sampled_kernel(
    const float* __restrict__ xn, const int64_t* __restrict__ labels, const int32_t* __restrict__ order,
    const int32_t* __restrict__ offsets, const int64_t* __restrict__ cls_toff, const int32_t* __restrict__ cls_k, int C,
    int M, int64_t T_arg, uint64_t seed, const uint64_t* __restrict__ seed_dev, const float* __restrict__ scale,
    float temperature, float margin, int kind, int RC,
    double* __restrict__ loss_sum, unsigned long long* __restrict__ kept_out, float* __restrict__ gxn,
    double* __restrict__ gscale, int64_t* __restrict__ a_dbg, int64_t* __restrict__ p_dbg, int64_t* __restrict__ n_dbg,
    unsigned perm_mul, int seg_len) {
    constexpr int E = 8 * EPL;
    using IO = RowIO<EPL, true>;
    constexpr int LD = E + 8;
    // positive table: row stride TS = E + 8 and element (lane s, j) at j*8 + s, so one ds_add_f32 wave-instruction
    // (8 groups x 8 lanes, same j) spreads over the banks instead of stacking 8 deep on every fourth one
    constexpr int TS = E + 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cache = smem;                                            // [RC][TS]  (GRAD only)
    float* stage = smem + (GRAD ? (size_t)RC * TS : 0);             // [waves][16][LD] negative-row ring + [waves][16] row ids
    __shared__ double red[16];
    __shared__ int cb_sh[3];

    // The number of sampled triplets is the plan's last prefix entry; T_arg only sized the grid (>= T: a replayed graph
    // keeps the grid of the largest batch seen so far, surplus workgroups fall through the bounds tests below).
    const int64_t T = cls_toff[C];
    (void)T_arg;
    if (seed_dev) seed = *seed_dev;
    const float sraw = *scale;
    const float sc = fminf(fmaxf(sraw, 1e-4f), 1.f);
    const bool s_live = sraw >= 1e-4f && sraw <= 1.f;
    const float inv_t = 1.f / temperature;
    const unsigned seed_lo = (unsigned)seed, seed_hi = (unsigned)(seed >> 32);
    const int lane = threadIdx.x & 63;
    const int s = threadIdx.x % SUB, g = threadIdx.x / SUB;
    const int sub0 = lane & ~(SUB - 1);
    const unsigned pblock = (unsigned)(((uint64_t)blockIdx.x * perm_mul) % gridDim.x);
    const int64_t block_base = (int64_t)pblock * ((int64_t)SB_GROUPS * seg_len);
    if (block_base >= T) return;             // surplus workgroup of a grid sized for a larger batch (uniform per block)

    if (threadIdx.x == 0) {
        const int cb = class_of(cls_toff, C, block_base < T ? block_base : T - 1);
        cb_sh[0] = cb;
        cb_sh[1] = offsets[cb];
        cb_sh[2] = offsets[cb + 1] - offsets[cb];
    }
    if (GRAD)
        for (int i = threadIdx.x; i < RC * TS; i += SB_THREADS) cache[i] = 0.f;
    __syncthreads();
    const int cb = cb_sh[0];

    const int64_t seg = block_base + (int64_t)g * seg_len;
    // lane-private class cursor (all 8 lanes of a group are within a few triplets of each other)
    int c = class_of(cls_toff, C, seg < T ? seg : T - 1);
    int64_t c_t0 = cls_toff[c], c_t1 = cls_toff[c + 1];
    int c_k = cls_k[c], c_off = offsets[c], c_na = offsets[c + 1] - c_off;
    double acc_loss = 0.0, acc_gs = 0.0;
    unsigned kept = 0;
    float acc_a[EPL];
    int cur = -1;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc_a[j] = 0.f;

    // wave-private ring of pending negative-row gradients (GRAD only)
    float* qrows = stage + (threadIdx.x >> 6) * (16 * LD);
    int* qid = reinterpret_cast<int*>(stage + (SB_THREADS / 64) * (16 * LD)) + (threadIdx.x >> 6) * 16;
    int qh = 0, qn = 0;
    for (int b8 = 0; b8 < seg_len / SUB; ++b8) {
        // ---- lane s generates triplet (seg + 8*b8 + s): eight index chains in flight per group --------------------
        const int64_t tl = seg + b8 * SUB + s;
        const bool lv_own = tl < T;
        const int64_t tt = lv_own ? tl : T - 1;
        while (tt >= c_t1 && c + 1 < C) {
            ++c;
            c_t0 = c_t1;
            c_t1 = cls_toff[c + 1];
            if (c_t1 > c_t0) { c_k = cls_k[c]; c_off = offsets[c]; c_na = offsets[c + 1] - c_off; }
        }
        const int64_t local = tt - c_t0;
        const unsigned kq = (unsigned)(c_k > 0 ? c_k : 1);
        const int rank = local < 0x7fffffffLL ? (int)((unsigned)local / kq) : (int)(local / kq);
        const int ia_own = order[c_off + rank];
        // counter RNG: two 32-bit streams per triplet from (seed, t)
        const unsigned hp = hash32((unsigned)tt ^ seed_lo, (unsigned)(tt >> 32) ^ seed_hi);
        const int rp = (int)__umulhi(hp, (unsigned)(c_na > 1 ? c_na - 1 : 1));
        const int pr_own = rp + (rp >= rank ? 1 : 0);
        const int ip_own = order[c_off + (pr_own < c_na ? pr_own : 0)];
        int in_own;
        {   // negative: uniform over the M - n_a points of the other classes, by rank in the class-major `order`
            // (ranks below the class's run map straight, the others skip the run): exactly the reference's
            // n_inds = where(labels != label) draw (loss_and_miner_utils.py:43-61), one lookup, no rejection loop
            const unsigned h = hash32(hp ^ 0x9E3779B9u, (unsigned)tt + seed_hi);
            const int n_other = M - c_na;
            const int rn = (int)__umulhi(h, (unsigned)(n_other > 0 ? n_other : 1));
            const int slot = rn + (rn >= c_off ? c_na : 0);
            in_own = order[slot < M ? slot : M - 1];
        }
        const int cpos_own = (c == cb && pr_own < RC) ? pr_own : -1;       // slot in the LDS positive table, or -1

        // ---- phase A: the group walks its 8 triplets for the three dot products; lane j keeps those of triplet j.
        // Per-triplet values travel down the group by one lane per step, so every broadcast has a constant source lane;
        // the rows of the next triplet are in flight during the math of the current one.
        float av[EPL], pv[EPL], nv[EPL], av2[EPL], pv2[EPL], nv2[EPL], pv3[EPL], nv3[EPL];
        float D1 = 0.f, D2 = 0.f, D3 = 0.f;
        // The anchor changes once per ~k_c consecutive triplets: its row is fetched only when it does (the kernel is bound
        // by these 128-byte L2 gathers: 3 rows per triplet per phase).
        {
            int ia_sh = ia_own, ip_sh = ip_own, in_sh = in_own;
            int ia_c = grp_bcast<0>(ia_sh);
            IO::load(xn, E, ia_c, s, av);
            IO::load(xn, E, grp_bcast<0>(ip_sh), s, pv);
            IO::load(xn, E, grp_bcast<0>(in_sh), s, nv);
            IO::load(xn, E, grp_bcast<1>(ip_sh), s, pv2);              // two triplets ahead: the gathers' L2 latency is
            IO::load(xn, E, grp_bcast<1>(in_sh), s, nv2);              // longer than one triplet's arithmetic
#pragma unroll 1
            for (int j8 = 0; j8 < SUB; ++j8) {
                bool a_new = false;
                if (j8 + 1 < SUB) {
                    const int ia_n = grp_bcast<1>(ia_sh);
                    a_new = ia_n != ia_c;
                    ia_c = ia_n;
                    if (a_new) IO::load(xn, E, ia_n, s, av2);
                }
                if (j8 + 2 < SUB) {
                    IO::load(xn, E, grp_bcast<2>(ip_sh), s, pv3);
                    IO::load(xn, E, grp_bcast<2>(in_sh), s, nv3);
                }
                float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    d1 += av[j] * pv[j];
                    d2 += av[j] * nv[j];
                    d3 += pv[j] * nv[j];
                }
                d1 = grp_sum(d1);
                d2 = grp_sum(d2);
                d3 = grp_sum(d3);
                const bool mine = s == j8;
                D1 = mine ? d1 : D1;
                D2 = mine ? d2 : D2;
                D3 = mine ? d3 : D3;
                ia_sh = grp_shift_down(ia_sh); ip_sh = grp_shift_down(ip_sh); in_sh = grp_shift_down(in_sh);
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    av[j] = a_new ? av2[j] : av[j];
                    pv[j] = pv2[j]; nv[j] = nv2[j];
                    pv2[j] = pv3[j]; nv2[j] = nv3[j];
                }
            }
        }
        // ---- phase B: every lane does the scalar part of ITS triplet (three LCA distances, the softmax, the weights)
        bool live = lv_own;
        if (kind >= 0) live = live && keep_rule(0.5f * (1.f + D1) - 0.5f * (1.f + D2), margin, kind);
        if (live) ++kept;
        // debug export (tests): the sampled triplet; a triplet the miner's rule dropped carries ~anchor (negative)
        if (a_dbg && lv_own) { a_dbg[tl] = live ? ia_own : ~ia_own; p_dbg[tl] = ip_own; n_dbg[tl] = in_own; }
        const LcaF L1 = lca_of_dot(sc, D1), L2 = lca_of_dot(sc, D2), L3 = lca_of_dot(sc, D3);
        const float q0 = L1.d * inv_t, q1 = L2.d * inv_t, q2 = L3.d * inv_t;
        const float mx = fmaxf(q0, fmaxf(q1, q2));
        const float e0 = __expf(q0 - mx), e1 = __expf(q1 - mx), e2 = __expf(q2 - mx);
        const float w0 = 0.5f * (1.f + D1), w1 = 0.5f * (1.f + D2), w2 = 0.5f * (1.f + D3);
        const float den = e0 + e1 + e2;
        const float inv_den = fast_rcp(den);                          // v_rcp_f32 (1 ulp) once, not four IEEE divisions
        const float wbar = (e0 * w0 + e1 * w1 + e2 * w2) * inv_den;
        if (live) acc_loss += (double)((w0 + w1 + w2) - wbar);
        if (GRAD) {
            const float sm0 = e0 * inv_den, sm1 = e1 * inv_den, sm2 = e2 * inv_den;
            const float gd0 = -(sm0 * (w0 - wbar)) * inv_t, gd1 = -(sm1 * (w1 - wbar)) * inv_t,
                        gd2 = -(sm2 * (w2 - wbar)) * inv_t;
            // unscaled weights of the three similarity gradients: upstream/kept are applied later
            const float c0_own = 0.5f * ((1.f - sm0) + gd0 * L1.dd_dw), c1_own = 0.5f * ((1.f - sm1) + gd1 * L2.dd_dw),
                        c2_own = 0.5f * ((1.f - sm2) + gd2 * L3.dd_dw);
            if (live && s_live)
                acc_gs += (double)(gd0 * L1.dd_ds) + (double)(gd1 * L2.dd_ds) + (double)(gd2 * L3.dd_ds);

            // ---- phase C: the group walks the 8 triplets again for the row gradients (rows re-read through L2); rows are
            // fetched only for kept triplets, and the anchor's only when it changes.  (Requesting the first rows BEFORE the
            // scalar math above was measured: the longer live ranges spill 6 VGPRs at the 128-register cap, 2452 vs 2378 us.)
            int ia_sh = ia_own, ip_sh = ip_own, in_sh = in_own;
            int cp_sh = live ? cpos_own : -2;                            // -2: dropped triplet, -1: positive not tabled
            int ia = grp_bcast<0>(ia_sh), ip = grp_bcast<0>(ip_sh), in = grp_bcast<0>(in_sh);
            int ia_have = -1;                                             // anchor whose row sits in av
            if (grp_bcast<0>(cp_sh) > -2) {
                IO::load(xn, E, ia, s, av);
                IO::load(xn, E, ip, s, pv);
                IO::load(xn, E, in, s, nv);
                ia_have = ia;
            }
            if (grp_bcast<1>(cp_sh) > -2) {                               // positive / negative rows run TWO triplets ahead,
                IO::load(xn, E, grp_bcast<1>(ip_sh), s, pv2);             // like phase A: an L2 gather outlasts one triplet's
                IO::load(xn, E, grp_bcast<1>(in_sh), s, nv2);             // arithmetic and atomics (2474 -> 2378 us)
            }
            int c0_sh = __float_as_int(c0_own), c1_sh = __float_as_int(c1_own), c2_sh = __float_as_int(c2_own);
#pragma unroll 1
            for (int j8 = 0; j8 < SUB; ++j8) {
                int ia2 = 0, ip2 = 0, in2 = 0;
                bool a_new = false, pn_new = false;
                if (j8 + 1 < SUB) {
                    ia2 = grp_bcast<1>(ia_sh);
                    ip2 = grp_bcast<1>(ip_sh);
                    in2 = grp_bcast<1>(in_sh);
                    pn_new = grp_bcast<1>(cp_sh) > -2;
                    a_new = pn_new && ia2 != ia_have;
                    if (a_new) IO::load(xn, E, ia2, s, av2);
                }
                if (j8 + 2 < SUB && grp_bcast<2>(cp_sh) > -2) {
                    IO::load(xn, E, grp_bcast<2>(ip_sh), s, pv3);
                    IO::load(xn, E, grp_bcast<2>(in_sh), s, nv3);
                }
                const int cpos = grp_bcast<0>(cp_sh);
                const bool lv = cpos > -2;
                const float c0 = __int_as_float(grp_bcast<0>(c0_sh)), c1 = __int_as_float(grp_bcast<0>(c1_sh)),
                            c2 = __int_as_float(grp_bcast<0>(c2_sh));
                if (lv) {
                    if (ia != cur) {
                        if (cur >= 0) {
#pragma unroll
                            for (int j = 0; j < EPL; ++j) atomicAdd(gxn + (unsigned)(cur * E + s * EPL + j), acc_a[j]);
                        }
                        cur = ia;
#pragma unroll
                        for (int j = 0; j < EPL; ++j) acc_a[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < EPL; ++j) acc_a[j] += c0 * pv[j] + c1 * nv[j];
                    // positive row: LDS table of the workgroup's class, else straight to memory
                    if (cpos >= 0) {
                        float gp[EPL];
#pragma unroll
                        for (int j = 0; j < EPL; ++j) gp[j] = c0 * av[j] + c2 * nv[j];
                        lds_add_row<EPL>(cache + cpos * TS + s, gp);
                    }
                }
                // a positive beyond the table leaves through the ring, like a negative row (see hyphc_sampled4_kernel)
                bool ov_flush = false;
                {
                    const bool ov = cpos == -1;
                    const unsigned long long obal = __ballot(ov);
                    if (obal != 0ull) {
                        if (ov) {
                            const int rank = __popcll(obal & ((1ull << (lane & ~7)) - 1ull)) >> 3;
                            const int slot = (qh + qn + rank) & 15;
                            float* qr = qrows + slot * LD + s * EPL;
#pragma unroll
                            for (int j = 0; j < EPL; ++j) qr[j] = c0 * av[j] + c2 * nv[j];
                            if (s == 0) qid[slot] = ip;
                        }
                        qn += __popcll(obal) >> 3;
                        ov_flush = qn >= 8;
                    }
                }
                if (ov_flush) {
                    constexpr int RPI = 64 / E;
#pragma unroll
                    for (int r = 0; r < EPL; ++r) {
                        const int slot = (qh + r * RPI + lane / E) & 15, e = lane % E;
                        atomicAdd(gxn + (unsigned)(qid[slot] * E + e), qrows[slot * LD + e]);
                    }
                    qh = (qh + 8) & 15;
                    qn -= 8;
                }
                // negative rows: a CU issues one 256-byte atomic wave-instruction per ~120 cycles, and this kernel is bound by
                // exactly that.  Kept rows are therefore packed into a wave-private LDS ring (16 rows) and leave in groups
                // of eight: 64 / E rows per instruction, every instruction full (pairing the rows of the current step
                // left a third of the instructions half empty).
                {
                    const unsigned long long lbal = __ballot(lv);                 // all 8 lanes of a kept group are set
                    const int n_live = __popcll(lbal) >> 3;
                    if (lv) {
                        const int rank = __popcll(lbal & ((1ull << (lane & ~7)) - 1ull)) >> 3;
                        const int slot = (qh + qn + rank) & 15;
                        float* qr = qrows + slot * LD + s * EPL;
#pragma unroll
                        for (int j = 0; j < EPL; ++j) qr[j] = c1 * av[j] + c2 * pv[j];
                        if (s == 0) qid[slot] = in;
                    }
                    qn += n_live;
                    if (qn >= 8) {
                        constexpr int RPI = 64 / E;
#pragma unroll
                        for (int r = 0; r < EPL; ++r) {
                            const int slot = (qh + r * RPI + lane / E) & 15, e = lane % E;
                            atomicAdd(gxn + (unsigned)(qid[slot] * E + e), qrows[slot * LD + e]);
                        }
                        qh = (qh + 8) & 15;
                        qn -= 8;
                    }
                }
                ia = ia2; ip = ip2; in = in2;
                ia_sh = grp_shift_down(ia_sh); ip_sh = grp_shift_down(ip_sh); in_sh = grp_shift_down(in_sh);
                cp_sh = grp_shift_down(cp_sh);
                c0_sh = grp_shift_down(c0_sh); c1_sh = grp_shift_down(c1_sh); c2_sh = grp_shift_down(c2_sh);
                if (a_new) ia_have = ia2;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    av[j] = a_new ? av2[j] : av[j];
                    pv[j] = pn_new ? pv2[j] : pv[j];
                    nv[j] = pn_new ? nv2[j] : nv[j];
                    pv2[j] = pv3[j];
                    nv2[j] = nv3[j];
                }
            }
        }
    }
    if (GRAD && qn > 0) {                                               // the ring's last (fewer than eight) rows
        constexpr int RPI = 64 / E;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int t_ = r * RPI + lane / E, e = lane % E;
            const int slot = (qh + t_) & 15;
            if (t_ < qn) atomicAdd(gxn + (unsigned)(qid[slot] * E + e), qrows[slot * LD + e]);
        }
    }
    if (GRAD && cur >= 0) {
#pragma unroll
        for (int j = 0; j < EPL; ++j) atomicAdd(gxn + (unsigned)(cur * E + s * EPL + j), acc_a[j]);
    }
    const double lt = hpcs_block_sum<double>(acc_loss, red);
    if (threadIdx.x == 0 && lt != 0.0) atomicAdd(loss_sum, lt);
    const double kt = hpcs_block_sum<double>((double)kept, red);
    if (threadIdx.x == 0 && kt != 0.0) atomicAdd(kept_out, (unsigned long long)kt);
    if (GRAD) {
        const double gt = hpcs_block_sum<double>(acc_gs, red);
        if (threadIdx.x == 0 && gt != 0.0) atomicAdd(gscale, gt);
        __syncthreads();
        // flush the positive table: row r of the class -> point order[off_cb + r]; 64/E rows per wave instruction
        const int rows = min(RC, cb_sh[2]);
        const int off_cb = cb_sh[1];
        constexpr int RPW = 64 / E;
        const int wave = threadIdx.x >> 6;
        for (int r0 = wave * RPW; r0 < rows; r0 += (SB_THREADS / 64) * RPW) {
            const int r = r0 + lane / E, e = lane % E;
            if (r < rows) {
                const float v = cache[r * TS + (e % EPL) * SUB + e / EPL];
                if (v != 0.f) atomicAdd(gxn + (int64_t)order[off_cb + r] * E + e, v);
            }
        }
    }
}

template <int EPL, bool GRAD>
__global__ __launch_bounds__(SB_THREADS) void hyphc_sampled4_kernel(
    const float* __restrict__ xn, const int64_t* __restrict__ labels, const int32_t* __restrict__ order,
    const int32_t* __restrict__ offsets, const int64_t* __restrict__ cls_toff, const int32_t* __restrict__ cls_k, int C,
    int M, int64_t T_arg, uint64_t seed, const uint64_t* __restrict__ seed_dev, const float* __restrict__ scale,
    float temperature, float margin, int kind, int RC,
    double* __restrict__ loss_sum, unsigned long long* __restrict__ kept_out, float* __restrict__ gxn,
    double* __restrict__ gscale, int64_t* __restrict__ a_dbg, int64_t* __restrict__ p_dbg, int64_t* __restrict__ n_dbg,
    unsigned perm_mul, int seg_len) {
    constexpr int E = 8 * EPL;
    using IO = RowIO<EPL, true>;
    constexpr int LD = E + 8;
    // positive table: row stride TS = E + 8 and element (lane s, j) at j*8 + s, so one ds_add_f32 wave-instruction
    // (8 groups x 8 lanes, same j) spreads over the banks instead of stacking 8 deep on every fourth one
    constexpr int TS = E + 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cache = smem;                                            // [RC][TS]  (GRAD only)
    float* stage = smem + (GRAD ? (size_t)RC * TS : 0);             // [waves][16][LD] negative-row ring + [waves][16] row ids
    __shared__ double red[16];
    __shared__ int cb_sh[3];

    // The number of sampled triplets is the plan's last prefix entry; T_arg only sized the grid (>= T: a replayed graph
    // keeps the grid of the largest batch seen so far, surplus workgroups fall through the bounds tests below).
    // 32-bit triplet counters (the launcher sends T >= 2^31 - 2^24 to the r02 kernel): six registers at the 128 cap
    const int T = (int)cls_toff[C];
    (void)T_arg;
    if (seed_dev) seed = *seed_dev;
    const float sraw = *scale;
    const float sc = fminf(fmaxf(sraw, 1e-4f), 1.f);
    const bool s_live = sraw >= 1e-4f && sraw <= 1.f;
    const float inv_t = 1.f / temperature;
    const unsigned seed_lo = (unsigned)seed, seed_hi = (unsigned)(seed >> 32);
    const int lane = threadIdx.x & 63;
    const int s = threadIdx.x % SUB, g = threadIdx.x / SUB;
    const int sub0 = lane & ~(SUB - 1);
    const unsigned pblock = (unsigned)(((uint64_t)blockIdx.x * perm_mul) % gridDim.x);
    const int block_base = (int)(pblock * (unsigned)(SB_GROUPS * seg_len));
    if (block_base >= T) return;             // surplus workgroup of a grid sized for a larger batch (uniform per block)

    if (threadIdx.x == 0) {
        const int cb = class_of(cls_toff, C, block_base < T ? block_base : T - 1);
        cb_sh[0] = cb;
        cb_sh[1] = offsets[cb];
        cb_sh[2] = offsets[cb + 1] - offsets[cb];
    }
    if (GRAD)
        for (int i = threadIdx.x; i < RC * TS; i += SB_THREADS) cache[i] = 0.f;
    __syncthreads();
    const int cb = cb_sh[0];

    const int seg = block_base + g * seg_len;
    // lane-private class cursor (all 8 lanes of a group are within a few triplets of each other)
    int c = class_of(cls_toff, C, seg < T ? seg : T - 1);
    int c_t0 = (int)cls_toff[c], c_t1 = (int)cls_toff[c + 1];
    int c_k = cls_k[c], c_off = offsets[c], c_na = offsets[c + 1] - c_off;
    double acc_loss = 0.0, acc_gs = 0.0;
    unsigned kept = 0;
    float acc_a[EPL];
    int cur = -1;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc_a[j] = 0.f;

    // wave-private ring of pending negative-row gradients (GRAD only)
    float* qrows = stage + (threadIdx.x >> 6) * (16 * LD);
    int* qid = reinterpret_cast<int*>(stage + (SB_THREADS / 64) * (16 * LD)) + (threadIdx.x >> 6) * 16;
    int qh = 0, qn = 0;
    // r03: FOUR triplets per 8-lane group and batch (lanes s and s + 4 generate the same triplet).  The rows of a batch --
    // four positive and four negative rows, 32 registers -- are requested together and STAY in registers through the
    // scalar phase, so the gradient phase gathers nothing a second time (round 2's kernel walked its 8 triplets twice and
    // fetched every kept triplet's rows again through L2: 0.67 ms of 2.06 at 50 % kept); the price is the per-triplet scalar
    // math (three LCA distances, softmax) evaluated by two lanes instead of one.
    int ia_have = -1;                                                   // anchor whose row sits in av
    float av[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) av[j] = 0.f;
    const bool owner = s < 4;                                           // lanes 4..7 shadow lanes 0..3
    for (int b4 = 0; b4 < seg_len / 4; ++b4) {
        const int tl = seg + b4 * 4 + (s & 3);
        const bool lv_own = tl < T;
        const int tt = lv_own ? tl : T - 1;
        while (tt >= c_t1 && c + 1 < C) {
            ++c;
            c_t0 = c_t1;
            c_t1 = (int)cls_toff[c + 1];
            if (c_t1 > c_t0) { c_k = cls_k[c]; c_off = offsets[c]; c_na = offsets[c + 1] - c_off; }
        }
        const unsigned kq = (unsigned)(c_k > 0 ? c_k : 1);
        // anchor rank = (t - t0_c) / k_c.  A class with fewer than 2^24 triplets: float reciprocal + one correction step
        // (x is exact in fp32, the quotient estimate is off by at most one) instead of the ~25-instruction 32-bit division
        int rank;
        {
            const unsigned x = (unsigned)(tt - c_t0);
            if ((unsigned)(c_t1 - c_t0) < (1u << 24) && kq < (1u << 24)) {
                unsigned qd = (unsigned)((float)x * fast_rcp((float)kq));
                const int rem = (int)(x - __umul24(qd, kq));
                qd += rem >= (int)kq ? 1u : 0u;
                qd -= rem < 0 ? 1u : 0u;
                rank = (int)qd;
            } else {
                rank = (int)(x / kq);
            }
        }
        const int ia_own = order[c_off + rank];
        const unsigned hp = hash32((unsigned)tt ^ seed_lo, seed_hi);          // (tt >> 32) == 0: the r02 kernel's stream
        const int rp = (int)__umulhi(hp, (unsigned)(c_na > 1 ? c_na - 1 : 1));
        const int pr_own = rp + (rp >= rank ? 1 : 0);
        const int ip_own = order[c_off + (pr_own < c_na ? pr_own : 0)];
        int in_own;
        {
            const unsigned h = hash32(hp ^ 0x9E3779B9u, (unsigned)tt + seed_hi);
            const int n_other = M - c_na;
            const int rn = (int)__umulhi(h, (unsigned)(n_other > 0 ? n_other : 1));
            const int slot = rn + (rn >= c_off ? c_na : 0);
            in_own = order[slot < M ? slot : M - 1];
        }
        const int cpos_own = (c == cb && pr_own < RC) ? pr_own : -1;       // slot in the LDS positive table, or -1

        // ---- phase A: all eight rows of the batch in flight at once; three dot products per triplet
        float P[4][EPL], Ng[4][EPL];
        // (indices are re-broadcast where they are needed -- two DPP moves -- instead of living in 12 registers across the
        // scalar phase: the kernel sits at the 128-register cap of four waves per SIMD)
#define BQ(v, q) ((q) == 0 ? grp_bcast<0>(v) : (q) == 1 ? grp_bcast<1>(v) : (q) == 2 ? grp_bcast<2>(v) : grp_bcast<3>(v))
        // eight ring rows (two per wave-instruction at E = 32) -> row-contiguous global atomics
#define HPCS_RING_FLUSH8                                                                                   \
        {                                                                                                  \
            constexpr int RPI_ = 64 / E;                                                                   \
            _Pragma("unroll") for (int r_ = 0; r_ < EPL; ++r_) {                                           \
                const int slot_ = (qh + r_ * RPI_ + lane / E) & 15, e_ = lane % E;                         \
                atomicAdd(gxn + (unsigned)(qid[slot_] * E + e_), qrows[slot_ * LD + e_]);                  \
            }                                                                                              \
            qh = (qh + 8) & 15;                                                                            \
            qn -= 8;                                                                                       \
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            IO::load(xn, E, BQ(ip_own, q), s, P[q]);
            IO::load(xn, E, BQ(in_own, q), s, Ng[q]);
        }
        float D1 = 0.f, D2 = 0.f, D3 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ia_q = BQ(ia_own, q);
            if (ia_q != ia_have) {                                      // the anchor changes once per ~k_c triplets
                IO::load(xn, E, ia_q, s, av);
                ia_have = ia_q;
            }
            float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                d1 += av[j] * P[q][j];
                d2 += av[j] * Ng[q][j];
                d3 += P[q][j] * Ng[q][j];
            }
            d1 = grp_sum(d1);
            d2 = grp_sum(d2);
            d3 = grp_sum(d3);
            const bool mine = (s & 3) == q;
            D1 = mine ? d1 : D1;
            D2 = mine ? d2 : D2;
            D3 = mine ? d3 : D3;
        }
        // ---- phase B: the scalar part of this lane's triplet (three LCA distances, the softmax, the weights)
        bool live = lv_own;
        if (kind >= 0) live = live && keep_rule(0.5f * (1.f + D1) - 0.5f * (1.f + D2), margin, kind);
        if (live && owner) ++kept;
        // debug export (tests): the sampled triplet; a triplet the miner's rule dropped carries ~anchor (negative)
        if (a_dbg && lv_own && owner) { a_dbg[tl] = live ? ia_own : ~ia_own; p_dbg[tl] = ip_own; n_dbg[tl] = in_own; }
        // Lanes s and s + 4 hold the same triplet: the lower quad evaluates the (a, p) distance, the upper quad the (p, n)
        // distance, then they swap -- two LCA evaluations per lane instead of three.
        LcaF L1, L3;
        {
            const LcaF Lx = lca_of_dot(sc, owner ? D1 : D3);
            const float od = grp_other_half(Lx.d), ow = grp_other_half(Lx.dd_dw), os = grp_other_half(Lx.dd_ds);
            L1.d = owner ? Lx.d : od; L1.dd_dw = owner ? Lx.dd_dw : ow; L1.dd_ds = owner ? Lx.dd_ds : os;
            L3.d = owner ? od : Lx.d; L3.dd_dw = owner ? ow : Lx.dd_dw; L3.dd_ds = owner ? os : Lx.dd_ds;
        }
        const LcaF L2 = lca_of_dot(sc, D2);
        const float q0 = L1.d * inv_t, q1 = L2.d * inv_t, q2 = L3.d * inv_t;
        const float mx = fmaxf(q0, fmaxf(q1, q2));
        const float e0 = __expf(q0 - mx), e1 = __expf(q1 - mx), e2 = __expf(q2 - mx);
        const float w0 = 0.5f * (1.f + D1), w1 = 0.5f * (1.f + D2), w2 = 0.5f * (1.f + D3);
        const float den = e0 + e1 + e2;
        const float inv_den = fast_rcp(den);                          // v_rcp_f32 (1 ulp) once, not four IEEE divisions
        const float wbar = (e0 * w0 + e1 * w1 + e2 * w2) * inv_den;
        if (live && owner) acc_loss += (double)((w0 + w1 + w2) - wbar);
        if (GRAD) {
            const float sm0 = e0 * inv_den, sm1 = e1 * inv_den, sm2 = e2 * inv_den;
            const float gd0 = -(sm0 * (w0 - wbar)) * inv_t, gd1 = -(sm1 * (w1 - wbar)) * inv_t,
                        gd2 = -(sm2 * (w2 - wbar)) * inv_t;
            const float c0_own = 0.5f * ((1.f - sm0) + gd0 * L1.dd_dw), c1_own = 0.5f * ((1.f - sm1) + gd1 * L2.dd_dw),
                        c2_own = 0.5f * ((1.f - sm2) + gd2 * L3.dd_dw);
            if (live && owner && s_live)
                acc_gs += (double)(gd0 * L1.dd_ds) + (double)(gd1 * L2.dd_ds) + (double)(gd2 * L3.dd_ds);
            // ---- phase C: row gradients from the rows still in registers.  The kernel is instruction-issue bound (r03 PMC:
            // some unit issues in 86 % of the cycles, four waves per SIMD), and a divergent `if` costs ~5 scalar
            // instructions whether or not it is taken: dropped triplets carry ZERO coefficients instead of being branched
            // around, and the rare events (anchor change, a positive outside the table, a lost swap) sit behind wave-uniform
            // tests.
            const int cp_own = live ? cpos_own : -2;                        // -2: dropped triplet, -1: positive not tabled
            const float c0_z = live ? c0_own : 0.f, c1_z = live ? c1_own : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cpos = BQ(cp_own, q);
                const bool lv = cpos > -2;
                const float c0 = BQ(c0_z, q), c1 = BQ(c1_z, q), c2 = BQ(c2_own, q);
                const int ia = BQ(ia_own, q), in = BQ(in_own, q);
                const bool moved = lv && (ia != cur || ia != ia_have);
                if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {           // once per ~k_c triplets and group
                    // a finished anchor's gradient leaves through the ring too (whole rows per atomic instruction)
                    const bool fl = moved && ia != cur && cur >= 0;
                    const unsigned long long fbal = __ballot(fl);
                    if (fbal != 0ull) {
                        if (fl) {
                            const int rank_q = __popcll(fbal & ((1ull << (lane & ~7)) - 1ull)) >> 3;
                            const int slot = (qh + qn + rank_q) & 15;
                            float* qr = qrows + slot * LD + s * EPL;
#pragma unroll
                            for (int j = 0; j < EPL; ++j) qr[j] = acc_a[j];
                            if (s == 0) qid[slot] = cur;
                        }
                        qn += __popcll(fbal) >> 3;
                        if (qn >= 8) { HPCS_RING_FLUSH8 }
                    }
                    if (moved) {
                        if (ia != ia_have) {                                // (the anchor changed inside the batch)
                            IO::load(xn, E, ia, s, av);
                            ia_have = ia;
                        }
                        if (ia != cur) {
                            cur = ia;
#pragma unroll
                            for (int j = 0; j < EPL; ++j) acc_a[j] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < EPL; ++j) acc_a[j] += c0 * P[q][j] + c1 * Ng[q][j];     // zeros for a dropped triplet
                {
                    const unsigned long long lbal = __ballot(lv);                 // all 8 lanes of a kept group are set
                    const int n_live = __popcll(lbal) >> 3;
                    if (lv) {
                        const int rank_q = __popcll(lbal & ((1ull << (lane & ~7)) - 1ull)) >> 3;
                        const int slot = (qh + qn + rank_q) & 15;
                        float* qr = qrows + slot * LD + s * EPL;
#pragma unroll
                        for (int j = 0; j < EPL; ++j) qr[j] = c1 * av[j] + c2 * P[q][j];
                        if (s == 0) qid[slot] = in;
                    }
                    qn += n_live;
                    if (qn >= 8) {
                        constexpr int RPI = 64 / E;
#pragma unroll
                        for (int r = 0; r < EPL; ++r) {
                            const int slot = (qh + r * RPI + lane / E) & 15, e = lane % E;
                            atomicAdd(gxn + (unsigned)(qid[slot] * E + e), qrows[slot * LD + e]);
                        }
                        qh = (qh + 8) & 15;
                        qn -= 8;
                    }
                }
                // positive row last (LDS executes in order per wave: the ring's reads above do not queue behind the table update)
                if (cpos >= 0) {
                    float gp[EPL];
#pragma unroll
                    for (int j = 0; j < EPL; ++j) gp[j] = c0 * av[j] + c2 * Ng[q][j];
                    lds_add_row<EPL>(cache + cpos * TS + s, gp);
                }
                // A positive beyond the table (classes with more members than its RC rows: 11.5 % of the triplets of the bench's
                // imbalanced labels, whose four largest classes hold 2600-3800 points against RC = 556): through the RING, like a
                // negative row -- whole 128-byte rows per atomic instruction.  (It used to add its 4 elements per lane straight
                // from the registers: four instructions of 4-byte pieces 16 bytes apart, four times the 32-byte segments of a
                // contiguous row -- about +45 % on the kernel's global-atomic work.)
                {
                    const bool ov = cpos == -1;
                    const unsigned long long obal = __ballot(ov);
                    if (obal != 0ull) {
                        const int ip = BQ(ip_own, q);
                        if (ov) {
                            const int rank_q = __popcll(obal & ((1ull << (lane & ~7)) - 1ull)) >> 3;
                            const int slot = (qh + qn + rank_q) & 15;
                            float* qr = qrows + slot * LD + s * EPL;
#pragma unroll
                            for (int j = 0; j < EPL; ++j) qr[j] = c0 * av[j] + c2 * Ng[q][j];
                            if (s == 0) qid[slot] = ip;
                        }
                        qn += __popcll(obal) >> 3;
                        if (qn >= 8) {
                            constexpr int RPI = 64 / E;
#pragma unroll
                            for (int r = 0; r < EPL; ++r) {
                                const int slot = (qh + r * RPI + lane / E) & 15, e = lane % E;
                                atomicAdd(gxn + (unsigned)(qid[slot] * E + e), qrows[slot * LD + e]);
                            }
                            qh = (qh + 8) & 15;
                            qn -= 8;
                        }
                    }
                }
            }
        }
    }
    if (GRAD) {                                                         // every group's last anchor joins the ring
        const bool fl = cur >= 0;
        const unsigned long long fbal = __ballot(fl);
        if (fbal != 0ull) {
            if (fl) {
                const int rank_q = __popcll(fbal & ((1ull << (lane & ~7)) - 1ull)) >> 3;
                const int slot = (qh + qn + rank_q) & 15;
                float* qr = qrows + slot * LD + s * EPL;
#pragma unroll
                for (int j = 0; j < EPL; ++j) qr[j] = acc_a[j];
                if (s == 0) qid[slot] = cur;
            }
            qn += __popcll(fbal) >> 3;
            if (qn >= 8) { HPCS_RING_FLUSH8 }
        }
    }
    if (GRAD && qn > 0) {                                               // the ring's last (fewer than eight) rows
        constexpr int RPI = 64 / E;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int t_ = r * RPI + lane / E, e = lane % E;
            const int slot = (qh + t_) & 15;
            if (t_ < qn) atomicAdd(gxn + (unsigned)(qid[slot] * E + e), qrows[slot * LD + e]);
        }
    }
    const double lt = hpcs_block_sum<double>(acc_loss, red);
    if (threadIdx.x == 0 && lt != 0.0) atomicAdd(loss_sum, lt);
    const double kt = hpcs_block_sum<double>((double)kept, red);
    if (threadIdx.x == 0 && kt != 0.0) atomicAdd(kept_out, (unsigned long long)kt);
    if (GRAD) {
        const double gt = hpcs_block_sum<double>(acc_gs, red);
        if (threadIdx.x == 0 && gt != 0.0) atomicAdd(gscale, gt);
        __syncthreads();
        // flush the positive table: row r of the class -> point order[off_cb + r]; 64/E rows per wave instruction
        const int rows = min(RC, cb_sh[2]);
        const int off_cb = cb_sh[1];
        constexpr int RPW = 64 / E;
        const int wave = threadIdx.x >> 6;
        for (int r0 = wave * RPW; r0 < rows; r0 += (SB_THREADS / 64) * RPW) {
            const int r = r0 + lane / E, e = lane % E;
            if (r < rows) {
                const float v = cache[r * TS + (e % EPL) * SUB + e / EPL];
                if (v != 0.f) atomicAdd(gxn + (int64_t)order[off_cb + r] * E + e, v);
            }
        }
    }
}
#undef BQ
#undef HPCS_RING_FLUSH8

template <bool BWD>
int launch_hyphc(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n, const int64_t* count,
                 int64_t T_cap, const float* scale, float temperature, float margin, int kind, const int64_t* norm_count,
                 unsigned long long* kept_out, const float* gloss, double* loss_sum, float* gxn, double* gscale,
                 hipStream_t st) {
    const int64_t per_block = (int64_t)TRIP_PER_PASS * RUN;
    const dim3 grid((unsigned)((T_cap + per_block - 1) / per_block));
    unsigned perm_mul = 1;
    if (BWD) {
        static const unsigned primes[] = {7919u, 7907u, 7901u, 7883u, 7879u, 7877u};
        for (unsigned pm : primes)
            if (grid.x % pm != 0) { perm_mul = pm; break; }
    }
#define HPCS_HYPHC_CASE(EPL)                                                                                          \
    if (E <= 8 * EPL) {                                                                                               \
        if (E == 8 * EPL && EPL >= 2)                                                                                 \
            hipLaunchKernelGGL((hyphc_kernel<EPL, BWD, true>), grid, dim3(TPB), 0, st, xn, E, a, p, n, count, T_cap, \
                               scale, temperature, margin, kind, norm_count, kept_out, gloss, loss_sum, gxn, gscale, perm_mul); \
        else                                                                                                          \
            hipLaunchKernelGGL((hyphc_kernel<EPL, BWD, false>), grid, dim3(TPB), 0, st, xn, E, a, p, n, count, T_cap,\
                               scale, temperature, margin, kind, norm_count, kept_out, gloss, loss_sum, gxn, gscale, perm_mul); \
        return HPCS_OK;                                                                                               \
    }
    HPCS_HYPHC_CASE(1)
    HPCS_HYPHC_CASE(2)
    HPCS_HYPHC_CASE(4)
    HPCS_HYPHC_CASE(8)
#undef HPCS_HYPHC_CASE
    return HPCS_ERR_UNSUPPORTED;       // embedding sizes above 64 (the reference uses 4..50)
}

// ---------------------------------------------------------------------------------------------- CosFace
// logits = scale * (cos - margin*onehot); loss_row = sum over levels of -log( sum_{c in branch_l(y)} p_c ).
// A plain CosFace is one level whose branches are the single classes (branch_id == nullptr); the PartNet
// hierarchical variant (hierarchical_cosface_loss.py:9-28) passes branch_id [n_levels, C].
constexpr int MAX_LEVELS = 8;

__device__ __forceinline__ bool same_branch(const int32_t* __restrict__ bid, int level, int C, int c, int lab) {
    return bid ? bid[level * C + c] == bid[level * C + lab] : c == lab;
}

__global__ void cosface_fwd_kernel(const float* __restrict__ xn, const float* __restrict__ wn, const int64_t* __restrict__ labels,
                                   const int32_t* __restrict__ bid, int n_levels, int M, int E, int C, float margin,
                                   float scale, float* __restrict__ logits, float* __restrict__ loss_rows,
                                   float* __restrict__ lse_rows) {
    extern __shared__ float wsh[];          // [C][E]
    for (int i = threadIdx.x; i < C * E; i += blockDim.x) wsh[i] = wn[i];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const int lab = (int)labels[row];
    const float* xr = xn + (int64_t)row * E;
    float* lr = logits + (int64_t)row * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
        float cs = 0.f;
        for (int e = 0; e < E; ++e) cs = __builtin_fmaf(xr[e], wsh[c * E + e], cs);
        if (c == lab) {
            const float diff = (cs - margin) - cs;
            cs = cs + diff;
        }
        const float lg = cs * scale;
        lr[c] = lg;
        mx = fmaxf(mx, lg);
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lr[c] - mx);
    const float lse = mx + logf(se);
    float loss = 0.f;
    for (int l = 0; l < n_levels; ++l) {
        float sb = 0.f;
        for (int c = 0; c < C; ++c)
            if (same_branch(bid, l, C, c, lab)) sb += expf(lr[c] - mx);
        loss += lse - (mx + logf(sb));
    }
    loss_rows[row] = loss;
    lse_rows[row] = lse;
}

// d loss_row / d logit_c = n_levels * p_c - sum_l [c in branch_l(y)] p_c / S_l,   S_l = sum_{branch_l(y)} p
__device__ __forceinline__ void branch_mass(const float* __restrict__ lr, float lse, const int32_t* __restrict__ bid,
                                            int n_levels, int C, int lab, float (&S)[MAX_LEVELS]) {
    for (int l = 0; l < n_levels; ++l) {
        float sb = 0.f;
        for (int c = 0; c < C; ++c)
            if (same_branch(bid, l, C, c, lab)) sb += __expf(lr[c] - lse);
        S[l] = sb;
    }
}

__device__ __forceinline__ float dlogit(const float* __restrict__ lr, float lse, const int32_t* __restrict__ bid,
                                        int n_levels, int C, int lab, int c, const float (&S)[MAX_LEVELS]) {
    const float pc = __expf(lr[c] - lse);
    float d = (float)n_levels * pc;
    for (int l = 0; l < n_levels; ++l)
        if (same_branch(bid, l, C, c, lab)) d -= pc / S[l];
    return d;
}

// grad wrt xn rows; one thread per row.  The row's d(logit) vector is staged once in LDS ([row][C+1], conflict-free
// for the per-thread walk) so exp() is evaluated C times per row, not C*E times.
__global__ void cosface_bwd_x_kernel(const float* __restrict__ wn, const int64_t* __restrict__ labels,
                                     const float* __restrict__ logits, const float* __restrict__ lse_rows,
                                     const int32_t* __restrict__ bid, int n_levels, int M, int E, int C, float scale,
                                     const float* __restrict__ gloss, float* __restrict__ gxn) {
    extern __shared__ float wsh[];               // [C][E] weights, then [blockDim][C+1] d(logit)
    float* dl = wsh + C * E + threadIdx.x * (C + 1);
    const float gmean = *gloss / (float)M;
    for (int i = threadIdx.x; i < C * E; i += blockDim.x) wsh[i] = wn[i];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const int lab = (int)labels[row];
    const float* lr = logits + (int64_t)row * C;
    const float lse = lse_rows[row];
    float S[MAX_LEVELS];
    branch_mass(lr, lse, bid, n_levels, C, lab, S);
    for (int c = 0; c < C; ++c) dl[c] = dlogit(lr, lse, bid, n_levels, C, lab, c, S) * gmean * scale;
    for (int e = 0; e < E; ++e) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc = __builtin_fmaf(dl[c], wsh[c * E + e], acc);
        gxn[(int64_t)row * E + e] += acc;
    }
}

// grad wrt wn [C,E]: each workgroup takes a slab of rows, stages d(logit) [rows][C] in LDS, then thread o=(c,e)
// reduces the slab and issues ONE atomic per output.
__global__ void cosface_bwd_w_kernel(const float* __restrict__ xn, const int64_t* __restrict__ labels,
                                     const float* __restrict__ logits, const float* __restrict__ lse_rows,
                                     const int32_t* __restrict__ bid, int n_levels, int M, int E, int C, float scale,
                                     const float* __restrict__ gloss, int rows_per_block, float* __restrict__ gwn) {
    extern __shared__ float dl[];        // [rows_per_block][C]
    const float gmean = *gloss / (float)M;
    const int r0 = blockIdx.x * rows_per_block;
    const int nr = min(rows_per_block, M - r0);
    for (int r = threadIdx.x; r < nr; r += blockDim.x) {
        const int row = r0 + r;
        const int lab = (int)labels[row];
        const float* lr = logits + (int64_t)row * C;
        const float lse = lse_rows[row];
        float S[MAX_LEVELS];
        branch_mass(lr, lse, bid, n_levels, C, lab, S);
        for (int c = 0; c < C; ++c) dl[r * C + c] = dlogit(lr, lse, bid, n_levels, C, lab, c, S) * gmean * scale;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < C * E; o += blockDim.x) {
        const int c = o / E, e = o % E;
        float acc = 0.f;
        for (int r = 0; r < nr; ++r) acc = __builtin_fmaf(dl[r * C + c], xn[(int64_t)(r0 + r) * E + e], acc);
        atomicAdd(gwn + o, acc);
    }
}

// ---- tiled CosFace (the training path): 64 rows per step, everything staged through LDS, coalesced global traffic ----
// forward: thread (r = t % 64, g = t / 64) owns the classes g, g+4, ... of row r; the 64 x C logits tile is written back
// coalesced; threads 0..63 then finish their row (lse, per-level branch masses, loss).
constexpr int CF_ROWS = 64;
__global__ __launch_bounds__(256) void cosface_fwd_tile_kernel(const float* __restrict__ xn, const float* __restrict__ wn,
                                                              const int64_t* __restrict__ labels, const int32_t* __restrict__ bid,
                                                              int n_levels, int M, int E, int C, float margin, float scale,
                                                              float* __restrict__ logits, float* __restrict__ loss_rows,
                                                              float* __restrict__ lse_rows) {
    extern __shared__ float sm[];
    float* wsh = sm;                                 // [C][E]
    float* xs = wsh + C * E;                         // [64][E+1]
    float* lg = xs + CF_ROWS * (E + 1);              // [64][C+1]
    const int r0 = blockIdx.x * CF_ROWS;
    const int nr = min(CF_ROWS, M - r0);
    for (int i = threadIdx.x; i < C * E; i += 256) wsh[i] = wn[i];
    for (int i = threadIdx.x; i < nr * E; i += 256) xs[(i / E) * (E + 1) + i % E] = xn[(int64_t)r0 * E + i];
    __syncthreads();
    const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (r < nr) {
        const int lab = (int)labels[r0 + r];
        const float* xr = xs + r * (E + 1);
        for (int c = g; c < C; c += 4) {
            float cs = 0.f;
            for (int e = 0; e < E; ++e) cs = __builtin_fmaf(xr[e], wsh[c * E + e], cs);
            if (c == lab) {
                const float diff = (cs - margin) - cs;
                cs = cs + diff;
            }
            lg[r * (C + 1) + c] = cs * scale;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * C; i += 256) logits[(int64_t)r0 * C + i] = lg[(i / C) * (C + 1) + i % C];
    if (threadIdx.x < nr) {
        const int row = r0 + threadIdx.x;
        const int lab = (int)labels[row];
        const float* lr = lg + threadIdx.x * (C + 1);
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, lr[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(lr[c] - mx);
        const float lse = mx + logf(se);
        float loss = 0.f;
        for (int l = 0; l < n_levels; ++l) {
            float sb = 0.f;
            for (int c = 0; c < C; ++c)
                if (same_branch(bid, l, C, c, lab)) sb += expf(lr[c] - mx);
            loss += lse - (mx + logf(sb));
        }
        loss_rows[row] = loss;
        lse_rows[row] = lse;
    }
}

// backward, both gradients from ONE pass over the rows: a workgroup walks CF_SUB sub-tiles of 64 rows; per sub-tile
// the d(logit) tile is built in LDS (one thread per row), gxn rows are finished (thread = row x quarter of E) and the
// workgroup's part of gwn [C,E] accumulates in registers (thread = up to 7 of the C*E outputs), flushed with one
// atomic per output at the end (128 workgroups at M = 32768 instead of one atomic set per 64 rows).
constexpr int CF_SUB = 1;      // measured: 1 -> 49 us, 2 -> 66 us, 4 -> 122-144 us (M = 32768: parallelism beats fewer atomics)
constexpr int CF_MAXOUT = 8;                         // C*E <= 2048
__global__ __launch_bounds__(256) void cosface_bwd_tile_kernel(const float* __restrict__ xn, const float* __restrict__ wn,
                                                              const int64_t* __restrict__ labels,
                                                              const float* __restrict__ logits, const float* __restrict__ lse_rows,
                                                              const int32_t* __restrict__ bid, int n_levels, int M, int E, int C,
                                                              float scale, const float* __restrict__ gloss,
                                                              float* __restrict__ gxn, float* __restrict__ gwn) {
    extern __shared__ float sm[];
    float* wsh = sm;                                 // [C][E]
    float* xs = wsh + C * E;                         // [64][E+1]
    float* dl = xs + CF_ROWS * (E + 1);              // [64][C+1]: logits, then p, then d(logit) in place
    __shared__ int rlab[CF_ROWS];
    __shared__ float rlse[CF_ROWS];
    __shared__ float rS[CF_ROWS * MAX_LEVELS];
    const float gmean = *gloss / (float)M;
    for (int i = threadIdx.x; i < C * E; i += 256) wsh[i] = wn[i];
    float gacc[CF_MAXOUT];
#pragma unroll
    for (int q = 0; q < CF_MAXOUT; ++q) gacc[q] = 0.f;
    const int nout = C * E;
    for (int st = 0; st < CF_SUB; ++st) {
        const int r0 = (blockIdx.x * CF_SUB + st) * CF_ROWS;
        if (r0 >= M) break;
        const int nr = min(CF_ROWS, M - r0);
        __syncthreads();                             // previous sub-tile fully consumed (and wsh written)
        for (int i = threadIdx.x; i < nr * E; i += 256) xs[(i / E) * (E + 1) + i % E] = xn[(int64_t)r0 * E + i];
        for (int i = threadIdx.x; i < nr * C; i += 256) dl[(i / C) * (C + 1) + i % C] = logits[(int64_t)r0 * C + i];
        __syncthreads();
        // d(logit) in three flat steps so that all 256 threads work: p = exp(logit - lse) in place; the row's branch
        // masses S_l (one thread per row, adds only); d = (n_levels p - sum_l [c in branch_l(y)] p / S_l) * upstream
        if (threadIdx.x < nr) {
            rlab[threadIdx.x] = (int)labels[r0 + threadIdx.x];
            rlse[threadIdx.x] = lse_rows[r0 + threadIdx.x];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr * C; i += 256) {
            const int r = i / C, c = i % C;
            dl[r * (C + 1) + c] = __expf(dl[r * (C + 1) + c] - rlse[r]);
        }
        __syncthreads();
        if (threadIdx.x < nr) {
            const float* pr = dl + threadIdx.x * (C + 1);
            const int lab = rlab[threadIdx.x];
            for (int l = 0; l < n_levels; ++l) {
                float sb = 0.f;
                for (int c = 0; c < C; ++c)
                    if (same_branch(bid, l, C, c, lab)) sb += pr[c];
                rS[threadIdx.x * MAX_LEVELS + l] = sb;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nr * C; i += 256) {
            const int r = i / C, c = i % C;
            const float pc = dl[r * (C + 1) + c];
            const int lab = rlab[r];
            float d = (float)n_levels * pc;
            for (int l = 0; l < n_levels; ++l)
                if (same_branch(bid, l, C, c, lab)) d -= pc / rS[r * MAX_LEVELS + l];
            dl[r * (C + 1) + c] = d * gmean * scale;
        }
        __syncthreads();
        {   // gxn: thread (r, quarter of E)
            const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
            if (r < nr) {
                const float* dr = dl + r * (C + 1);
                for (int e = g; e < E; e += 4) {
                    float acc = 0.f;
                    for (int c = 0; c < C; ++c) acc = __builtin_fmaf(dr[c], wsh[c * E + e], acc);
                    gxn[(int64_t)(r0 + r) * E + e] += acc;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < CF_MAXOUT; ++q) {
            const int o = threadIdx.x + q * 256;
            if (o < nout) {
                const int c = o / E, e = o % E;
                float acc = gacc[q];
                for (int r = 0; r < nr; ++r) acc = __builtin_fmaf(dl[r * (C + 1) + c], xs[r * (E + 1) + e], acc);
                gacc[q] = acc;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CF_MAXOUT; ++q) {
        const int o = threadIdx.x + q * 256;
        if (o < nout) atomicAdd(gwn + o, gacc[q]);
    }
}

// ---- in-step metrics (next row f3): what base_hyp_hc.py:88-99,126-130 hands to torchmetrics every training step -------
// softmax is monotone, so the predicted class is the arg-max logit (first maximum on ties, like torch.argmax).  One
// thread per row; per-workgroup LDS counters [3][C] = {true positives, predictions, targets} per class, flushed with
// one integer atomic per touched counter.
__global__ __launch_bounds__(256) void class_metrics_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                           int M, int C, unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int cm[];              // [3][C]
    for (int i = threadIdx.x; i < 3 * C; i += 256) cm[i] = 0u;
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row < M) {
        const float* lr = logits + (int64_t)row * C;
        float bv = lr[0];
        int bc = 0;
        for (int c = 1; c < C; ++c) {
            const float v = lr[c];
            if (v > bv) { bv = v; bc = c; }
        }
        const int64_t y = labels[row];
        atomicAdd(&cm[C + bc], 1u);
        if (y >= 0 && y < C) {
            atomicAdd(&cm[2 * C + (int)y], 1u);
            if ((int)y == bc) atomicAdd(&cm[bc], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += 256)
        if (cm[i]) atomicAdd(counts + i, (unsigned long long)cm[i]);
}
// acc = sum TP / M (micro, top-1); iou = mean over the classes present in predictions or targets of TP / (P + T - TP)
__global__ void class_metrics_finalize_kernel(const unsigned long long* __restrict__ counts, int M, int C,
                                              float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double tp = 0.0, iou = 0.0;
    int present = 0;
    for (int c = 0; c < C; ++c) {
        const double t = (double)counts[c], p = (double)counts[C + c], y = (double)counts[2 * C + c];
        tp += t;
        if (p + y > 0.0) {
            ++present;
            iou += t / (p + y - t);
        }
    }
    out[0] = M > 0 ? (float)(tp / (double)M) : 0.f;
    out[1] = present > 0 ? (float)(iou / (double)present) : 0.f;
}

__device__ __forceinline__ void triplet_sims(const float* __restrict__ xn, int E, int64_t ia, int64_t ip, int64_t in, int s,
                                             float& wij, float& wik, float& wjk) {
    float d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int e = s; e < E; e += SUB) {
        const float av = xn[ia * E + e], pv = xn[ip * E + e], nv = xn[in * E + e];
        d1 += av * pv;
        d2 += av * nv;
        d3 += pv * nv;
    }
    wij = 0.5f * (1.f + sub_sum(d1));
    wik = 0.5f * (1.f + sub_sum(d2));
    wjk = 0.5f * (1.f + sub_sum(d3));
}

// ---------------------------------------------------------------------------------------------- triplet margin
__global__ void triplet_margin_fwd_kernel(const float* __restrict__ xn, int E, const int64_t* __restrict__ a,
                                          const int64_t* __restrict__ p, const int64_t* __restrict__ n,
                                          const int64_t* __restrict__ count, int64_t T_cap, float margin,
                                          double* __restrict__ sum_nnz) {
    __shared__ double red[16];
    const int64_t T = *count < T_cap ? *count : T_cap;
    const int s = threadIdx.x % SUB, g = threadIdx.x / SUB;
    double acc = 0.0, nz = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * TRIP_PER_PASS + g; t - g < T; t += (int64_t)gridDim.x * TRIP_PER_PASS) {
        const bool live = t < T;
        const int64_t ia = live ? a[t] : 0, ip = live ? p[t] : 0, in = live ? n[t] : 0;
        float wij, wik, wjk;
        triplet_sims(xn, E, ia, ip, in, s, wij, wik, wjk);
        const float viol = (wik - wij) + margin;
        if (live && s == 0 && viol > 0.f) { acc += (double)viol; nz += 1.0; }
    }
    const double t1 = hpcs_block_sum<double>(acc, red);
    const double t2 = hpcs_block_sum<double>(nz, red);
    if (threadIdx.x == 0 && t2 != 0.0) { atomicAdd(sum_nnz, t1); atomicAdd(sum_nnz + 1, t2); }
}

__global__ void triplet_margin_bwd_kernel(const float* __restrict__ xn, int E, const int64_t* __restrict__ a,
                                          const int64_t* __restrict__ p, const int64_t* __restrict__ n,
                                          const int64_t* __restrict__ count, int64_t T_cap, float margin,
                                          const double* __restrict__ sum_nnz, const float* __restrict__ gloss,
                                          float* __restrict__ gxn) {
    const int64_t T = *count < T_cap ? *count : T_cap;
    const double nnz = sum_nnz[1];
    if (nnz <= 0.0) return;
    const float up = 0.5f * (float)((double)(*gloss) / nnz);
    const int s = threadIdx.x % SUB, g = threadIdx.x / SUB;
    for (int64_t t = (int64_t)blockIdx.x * TRIP_PER_PASS + g; t - g < T; t += (int64_t)gridDim.x * TRIP_PER_PASS) {
        const bool live = t < T;
        const int64_t ia = live ? a[t] : 0, ip = live ? p[t] : 0, in = live ? n[t] : 0;
        float wij, wik, wjk;
        triplet_sims(xn, E, ia, ip, in, s, wij, wik, wjk);
        const float viol = (wik - wij) + margin;
        if (live && viol > 0.f)
            for (int e = s; e < E; e += SUB) {
                const float av = xn[ia * E + e], pv = xn[ip * E + e], nv = xn[in * E + e];
                atomicAdd(gxn + ia * E + e, up * (nv - pv));
                atomicAdd(gxn + in * E + e, up * av);
                atomicAdd(gxn + ip * E + e, -up * av);
            }
    }
}

inline unsigned trip_grid(int64_t T_cap) {
    int64_t b = (T_cap + TRIP_PER_PASS - 1) / TRIP_PER_PASS;
    if (b > 8192) b = 8192;          // grid-stride the rest
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" size_t hpcs_class_index_workspace(int M, int C) {
    if (M < 0 || C <= 0) return 0;
    const size_t nblk = (size_t)hpcs_cdiv(M > 0 ? M : 1, CI_TPB);
    return sizeof(int32_t) * (C <= CI_MAXC ? nblk * (size_t)C : (size_t)C);
}

extern "C" int hpcs_class_index(const int64_t* labels, int M, int C, int32_t* order, int32_t* offsets, int32_t* counts_ws,
                                size_t ws_bytes, hpcs_stream_t stream) {
    if (!labels || !order || !offsets || !counts_ws) return HPCS_ERR_NULL;
    if (M < 0 || C <= 0) return HPCS_ERR_SHAPE;
    if (ws_bytes < hpcs_class_index_workspace(M, C)) return HPCS_ERR_WORKSPACE;
    if (C <= CI_MAXC && M > 0) {
        const int nblk = hpcs_cdiv(M, CI_TPB);
        hipLaunchKernelGGL(class_hist_blocks_kernel, dim3(nblk), dim3(CI_TPB), 0, ST, labels, M, C, counts_ws);
        HPCS_CHECK_LAUNCH();
        hipLaunchKernelGGL(class_scan_blocks_kernel, dim3(1), dim3(CI_MAXC), 0, ST, counts_ws, nblk, C, offsets);
        HPCS_CHECK_LAUNCH();
        hipLaunchKernelGGL(class_fill_blocks_kernel, dim3(nblk), dim3(CI_TPB), sizeof(int) * (CI_TPB / 64) * (size_t)C, ST, labels,
                           M, C, (const int32_t*)counts_ws, order);
        HPCS_CHECK_LAUNCH();
        return HPCS_OK;
    }
    if (const int zr = hpcs_zero_async(ST, counts_ws, sizeof(int32_t) * C)) return zr;
    if (M > 0) {
        hipLaunchKernelGGL(hist_kernel, dim3(hpcs_cdiv(M, TPB)), dim3(TPB), 0, ST, labels, M, C, counts_ws);
        HPCS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(64), 0, ST, counts_ws, C, offsets);
    HPCS_CHECK_LAUNCH();
    if (M > 0) {
        hipLaunchKernelGGL(class_fill_kernel, dim3(C), dim3(TPB), 0, ST, labels, M, offsets, order);
        HPCS_CHECK_LAUNCH();
    }
    return HPCS_OK;
}

extern "C" int hpcs_sample_triplets(const int32_t* order, const int32_t* offsets, const int64_t* cls_toff,
                                    const int32_t* cls_k, int C, int M, int64_t T, const int64_t* draws_pos,
                                    const int64_t* draws_neg, uint64_t seed, int64_t* a, int64_t* p, int64_t* n,
                                    hpcs_stream_t stream) {
    if (!order || !offsets || !cls_toff || !cls_k || !a || !p || !n) return HPCS_ERR_NULL;
    if ((draws_pos == nullptr) != (draws_neg == nullptr)) return HPCS_ERR_NULL;
    if (C <= 0 || M <= 0 || T < 0) return HPCS_ERR_SHAPE;
    if (T == 0) return HPCS_OK;
    hipLaunchKernelGGL(sample_triplets_kernel, dim3(hpcs_cdiv(T, TPB)), dim3(TPB), 0, ST, order, offsets, cls_toff, cls_k, C, M,
                       T, draws_pos, draws_neg, seed, a, p, n);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" size_t hpcs_triplet_filter_workspace(int64_t T) {
    if (T <= 0) return 64;
    const int64_t nb = (T + CHUNK - 1) / CHUNK;
    return (size_t)(((T + 15) / 16) * 16 + nb * sizeof(int32_t) + 16 + nb * sizeof(int64_t) + 64);
}

extern "C" int hpcs_triplet_filter(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n, int64_t T,
                                   float margin, int kind, int64_t* a_out, int64_t* p_out, int64_t* n_out, int64_t* count,
                                   void* ws, size_t ws_bytes, hpcs_stream_t stream) {
    if (!xn || !a || !p || !n || !a_out || !p_out || !n_out || !count || !ws) return HPCS_ERR_NULL;
    if (E <= 0 || T < 0 || kind < 0 || kind > 3) return HPCS_ERR_SHAPE;
    if (ws_bytes < hpcs_triplet_filter_workspace(T)) return HPCS_ERR_WORKSPACE;
    if (T == 0) {
        return hpcs_zero_async(ST, count, sizeof(int64_t));
    }
    const int64_t nb = (T + CHUNK - 1) / CHUNK;
    uint8_t* flags = (uint8_t*)ws;
    char* q = (char*)ws + ((T + 15) / 16) * 16;
    int32_t* block_cnt = (int32_t*)q;
    q += ((nb * sizeof(int32_t) + 15) / 16) * 16;
    int64_t* block_off = (int64_t*)q;
    hipLaunchKernelGGL(filter_flag_kernel, dim3((unsigned)nb), dim3(TPB), 0, ST, xn, E, a, p, n, T, margin, kind, flags,
                       block_cnt);
    HPCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(filter_scan_kernel, dim3(1), dim3(TPB), 0, ST, block_cnt, (int)nb, block_off, count);
    HPCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(filter_scatter_kernel, dim3((unsigned)nb), dim3(TPB), 0, ST, flags, block_off, a, p, n, T, a_out, p_out,
                       n_out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hyphc_fwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                              const int64_t* count, int64_t T_cap, const float* scale, float temperature, float margin,
                              int kind, double* loss_sum, int64_t* kept, hpcs_stream_t stream) {
    if (!xn || !a || !p || !n || !count || !scale || !loss_sum) return HPCS_ERR_NULL;
    if (E <= 0 || T_cap < 0 || !(temperature > 0.f) || kind > 3 || (kind >= 0 && !kept)) return HPCS_ERR_SHAPE;
    if (const int zr = hpcs_zero_async(ST, loss_sum, sizeof(double), kept, sizeof(int64_t))) return zr;
    if (T_cap == 0) return HPCS_OK;
    const int rc = launch_hyphc<false>(xn, E, a, p, n, count, T_cap, scale, temperature, margin, kind, nullptr,
                                       (unsigned long long*)kept, nullptr, loss_sum, nullptr, nullptr, ST);
    if (rc) return rc;
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hyphc_bwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                              const int64_t* count, int64_t T_cap, const float* scale, float temperature, float margin,
                              int kind, const int64_t* norm_count, const float* gloss, float* gxn, double* gscale,
                              double* loss_sum, int64_t* kept, hpcs_stream_t stream) {
    if (!xn || !a || !p || !n || !count || !scale || !gloss || !gxn || !gscale) return HPCS_ERR_NULL;
    if (E <= 0 || T_cap < 0 || !(temperature > 0.f) || kind > 3 || (kind >= 0 && !norm_count)) return HPCS_ERR_SHAPE;
    if (const int zr = hpcs_zero_async(ST, gscale, sizeof(double), loss_sum, sizeof(double), kept, sizeof(int64_t))) return zr;
    if (T_cap == 0) return HPCS_OK;
    const int rc = launch_hyphc<true>(xn, E, a, p, n, count, T_cap, scale, temperature, margin, kind, norm_count,
                                      (unsigned long long*)kept, gloss, loss_sum, gxn, gscale, ST);
    if (rc) return rc;
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_cosface_fwd(const float* xn, const float* wn, const int64_t* labels, const int32_t* branch_id,
                                int n_levels, int M, int E, int C, float margin, float scale, float* logits,
                                float* loss_rows, float* lse_rows, hpcs_stream_t stream) {
    if (!xn || !wn || !labels || !logits || !loss_rows || !lse_rows) return HPCS_ERR_NULL;
    if (M < 0 || E <= 0 || C <= 0 || (size_t)C * E * 4 > 64 * 1024) return HPCS_ERR_SHAPE;
    if (n_levels < 1 || n_levels > MAX_LEVELS || (n_levels > 1 && !branch_id)) return HPCS_ERR_SHAPE;
    if (M == 0) return HPCS_OK;
    const size_t tile_lds = ((size_t)C * E + CF_ROWS * (E + 1) + CF_ROWS * (C + 1)) * 4;
    if (tile_lds <= 64 * 1024)
        hipLaunchKernelGGL(cosface_fwd_tile_kernel, dim3(hpcs_cdiv(M, CF_ROWS)), dim3(256), tile_lds, ST, xn, wn, labels,
                           branch_id, n_levels, M, E, C, margin, scale, logits, loss_rows, lse_rows);
    else
        hipLaunchKernelGGL(cosface_fwd_kernel, dim3(hpcs_cdiv(M, 64)), dim3(64), (size_t)C * E * 4, ST, xn, wn, labels, branch_id,
                           n_levels, M, E, C, margin, scale, logits, loss_rows, lse_rows);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_cosface_bwd(const float* xn, const float* wn, const int64_t* labels, const float* logits,
                                const float* lse_rows, const int32_t* branch_id, int n_levels, int M, int E, int C,
                                float scale, const float* gloss, float* gxn, float* gwn, hpcs_stream_t stream) {
    if (!xn || !wn || !labels || !logits || !lse_rows || !gloss || !gxn || !gwn) return HPCS_ERR_NULL;
    if (M < 0 || E <= 0 || C <= 0 || (size_t)C * E * 4 > 64 * 1024) return HPCS_ERR_SHAPE;
    if (n_levels < 1 || n_levels > MAX_LEVELS || (n_levels > 1 && !branch_id)) return HPCS_ERR_SHAPE;
    if (const int zr = hpcs_zero_async(ST, gwn, sizeof(float) * C * E, gxn, sizeof(float) * (size_t)M * E)) return zr;
    if (M == 0) return HPCS_OK;
    const size_t tile_lds = ((size_t)C * E + CF_ROWS * (E + 1) + CF_ROWS * (C + 1)) * 4;
    if (tile_lds <= 64 * 1024 && C * E <= 256 * CF_MAXOUT) {
        hipLaunchKernelGGL(cosface_bwd_tile_kernel, dim3(hpcs_cdiv(M, CF_ROWS * CF_SUB)), dim3(256), tile_lds, ST, xn, wn, labels,
                           logits, lse_rows, branch_id, n_levels, M, E, C, scale, gloss, gxn, gwn);
        HPCS_CHECK_LAUNCH();
        return HPCS_OK;
    }
    hipLaunchKernelGGL(cosface_bwd_x_kernel, dim3(hpcs_cdiv(M, 64)), dim3(64), ((size_t)C * E + 64 * (C + 1)) * 4, ST, wn, labels,
                       logits, lse_rows, branch_id, n_levels, M, E, C, scale, gloss, gxn);
    HPCS_CHECK_LAUNCH();
    int rpb = (48 * 1024) / (4 * C);
    if (rpb > 256) rpb = 256;
    if (rpb < 1) return HPCS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cosface_bwd_w_kernel, dim3(hpcs_cdiv(M, rpb)), dim3(TPB), (size_t)rpb * C * 4, ST, xn, labels, logits,
                       lse_rows, branch_id, n_levels, M, E, C, scale, gloss, rpb, gwn);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

// hierarchical_loss(probabilities, targets, hierarchy_list) as a stand-alone op (hierarchical_cosface_loss.py:9-28):
// loss_row = sum_l -log( sum_{c in branch_l(y)} prob_c ); one thread per row (API op, not on the training path: the
// training path fuses this into hpcs_cosface_fwd/bwd).
__global__ void hier_nll_fwd_kernel(const float* __restrict__ prob, const int64_t* __restrict__ labels,
                                    const int32_t* __restrict__ bid, int n_levels, int M, int C, float* __restrict__ loss_rows) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const int lab = (int)labels[row];
    const float* pr = prob + (int64_t)row * C;
    float loss = 0.f;
    for (int l = 0; l < n_levels; ++l) {
        float sb = 0.f;
        for (int c = 0; c < C; ++c)
            if (same_branch(bid, l, C, c, lab)) sb += pr[c];
        loss -= logf(sb);
    }
    loss_rows[row] = loss;
}

__global__ void hier_nll_bwd_kernel(const float* __restrict__ prob, const int64_t* __restrict__ labels,
                                    const int32_t* __restrict__ bid, int n_levels, int M, int C, const float* __restrict__ gloss,
                                    float* __restrict__ gprob) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const int lab = (int)labels[row];
    const float* pr = prob + (int64_t)row * C;
    float invS[MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) {
        float sb = 0.f;
        for (int c = 0; c < C; ++c)
            if (same_branch(bid, l, C, c, lab)) sb += pr[c];
        invS[l] = 1.f / sb;
    }
    const float g = -gloss[0] / (float)M;
    for (int c = 0; c < C; ++c) {
        float d = 0.f;
        for (int l = 0; l < n_levels; ++l)
            if (same_branch(bid, l, C, c, lab)) d += invS[l];
        gprob[(int64_t)row * C + c] = g * d;
    }
}

extern "C" int hpcs_hier_nll_fwd(const float* prob, const int64_t* labels, const int32_t* branch_id, int n_levels, int M, int C,
                                 float* loss_rows, hpcs_stream_t stream) {
    if (!prob || !labels || !loss_rows) return HPCS_ERR_NULL;
    if (M < 0 || C <= 0) return HPCS_ERR_SHAPE;
    if (n_levels < 1 || n_levels > MAX_LEVELS || (n_levels > 1 && !branch_id)) return HPCS_ERR_SHAPE;
    if (M == 0) return HPCS_OK;
    hipLaunchKernelGGL(hier_nll_fwd_kernel, dim3(hpcs_cdiv(M, 64)), dim3(64), 0, ST, prob, labels, branch_id, n_levels, M, C,
                       loss_rows);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hier_nll_bwd(const float* prob, const int64_t* labels, const int32_t* branch_id, int n_levels, int M, int C,
                                 const float* gloss, float* gprob, hpcs_stream_t stream) {
    if (!prob || !labels || !gloss || !gprob) return HPCS_ERR_NULL;
    if (M < 0 || C <= 0) return HPCS_ERR_SHAPE;
    if (n_levels < 1 || n_levels > MAX_LEVELS || (n_levels > 1 && !branch_id)) return HPCS_ERR_SHAPE;
    if (M == 0) return HPCS_OK;
    hipLaunchKernelGGL(hier_nll_bwd_kernel, dim3(hpcs_cdiv(M, 64)), dim3(64), 0, ST, prob, labels, branch_id, n_levels, M, C,
                       gloss, gprob);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_class_metrics(const float* logits, const int64_t* labels, int M, int C, int64_t* counts, float* acc_iou,
                                  hpcs_stream_t stream) {
    if (!logits || !labels || !counts || !acc_iou) return HPCS_ERR_NULL;
    if (M < 0 || C <= 0 || C > 4096) return HPCS_ERR_SHAPE;
    if (const int zr = hpcs_zero_async(ST, counts, sizeof(int64_t) * 3 * C)) return zr;
    if (M > 0) {
        hipLaunchKernelGGL(class_metrics_kernel, dim3(hpcs_cdiv(M, 256)), dim3(256), sizeof(unsigned) * 3 * C, ST, logits, labels,
                           M, C, (unsigned long long*)counts);
        HPCS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(class_metrics_finalize_kernel, dim3(1), dim3(64), 0, ST, (const unsigned long long*)counts, M, C,
                       acc_iou);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_triplet_margin_fwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                                       const int64_t* count, int64_t T_cap, float margin, double* sum_and_nnz,
                                       hpcs_stream_t stream) {
    if (!xn || !a || !p || !n || !count || !sum_and_nnz) return HPCS_ERR_NULL;
    if (E <= 0 || T_cap < 0) return HPCS_ERR_SHAPE;
    if (const int zr = hpcs_zero_async(ST, sum_and_nnz, 2 * sizeof(double))) return zr;
    if (T_cap == 0) return HPCS_OK;
    hipLaunchKernelGGL(triplet_margin_fwd_kernel, dim3(trip_grid(T_cap)), dim3(TPB), 0, ST, xn, E, a, p, n, count, T_cap,
                       margin, sum_and_nnz);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_triplet_margin_bwd(const float* xn, int E, const int64_t* a, const int64_t* p, const int64_t* n,
                                       const int64_t* count, int64_t T_cap, float margin, const double* sum_and_nnz,
                                       const float* gloss, float* gxn, hpcs_stream_t stream) {
    if (!xn || !a || !p || !n || !count || !sum_and_nnz || !gloss || !gxn) return HPCS_ERR_NULL;
    if (E <= 0 || T_cap < 0) return HPCS_ERR_SHAPE;
    if (T_cap == 0) return HPCS_OK;
    hipLaunchKernelGGL(triplet_margin_bwd_kernel, dim3(trip_grid(T_cap)), dim3(TPB), 0, ST, xn, E, a, p, n, count, T_cap,
                       margin, sum_and_nnz, gloss, gxn);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hyphc_sampled(const float* xn, const int64_t* labels, const int32_t* order, const int32_t* offsets,
                                  const int64_t* cls_toff, const int32_t* cls_k, int C, int M, int E, int64_t T,
                                  uint64_t seed, const uint64_t* seed_dev, const float* scale, float temperature,
                                  float margin, int kind,
                                  double* loss_sum, int64_t* kept, float* gxn, double* gscale, int64_t* a_dbg,
                                  int64_t* p_dbg, int64_t* n_dbg, hpcs_stream_t stream) {
    if (!xn || !labels || !order || !offsets || !cls_toff || !cls_k || !scale || !loss_sum || !kept) return HPCS_ERR_NULL;
    if ((gxn == nullptr) != (gscale == nullptr)) return HPCS_ERR_NULL;
    if ((a_dbg || p_dbg || n_dbg) && !(a_dbg && p_dbg && n_dbg)) return HPCS_ERR_NULL;
    if (C <= 0 || M <= 0 || T < 0 || !(temperature > 0.f) || kind > 3) return HPCS_ERR_SHAPE;
    if (E != 16 && E != 32 && E != 64) return HPCS_ERR_UNSUPPORTED;
    if (const int zr = hpcs_zero_async(ST, loss_sum, sizeof(double), kept, sizeof(int64_t), gscale, sizeof(double), gxn,
                                       gxn ? sizeof(float) * (size_t)M * E : 0))
        return zr;
    if (T == 0) return HPCS_OK;
    // Workgroups are resident one per CU (128 KB of LDS), so the grid runs in rounds of n_cu workgroups: pick the
    // segment length (multiple of 8) that wastes the least of the last round; ties go to the longer segment, which
    // amortises the positive table's zero/flush best.  The triplet stream depends on (seed, t) only, not on this.
    static const int n_cu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        return v;
    }();
    // kernel variant: 4 (default, r03: four triplets per group and batch, rows kept in registers) | 8 (r02: eight, rows
    // gathered twice; E = 64 stays on it: eight rows of 8 registers spill at the 128-register cap); tuning table
    const int variant = (hpcs_tuning().sampled_v == 8 || T >= (int64_t)0x7F000000 || E == 64) ? 8 : 4;
    const int seg_env = hpcs_tuning().sampled_seg;                   // override (multiple of 8 in [8, 256]); 0: chosen below
    int seg_len = SB_SEG_MAX;
    if (seg_env >= 8 && seg_env <= SB_SEG_MAX && seg_env % 8 == 0) {
        seg_len = seg_env;
    } else {
        int64_t best = INT64_MAX;
        for (int sl = SB_SEG_MAX; sl >= SB_SEG_MIN; sl -= 8) {
            const int64_t blocks = (T + (int64_t)SB_GROUPS * sl - 1) / ((int64_t)SB_GROUPS * sl);
            // rounds x (segment + a workgroup's fixed cost: zeroing and flushing the positive table, ~11 segment units -- fitted to
            // 104 vs 176 at T = 16.9 M: 1.258 vs 1.227 ms)
            const int64_t cost = ((blocks + n_cu - 1) / n_cu) * (sl + 11);
            if (cost < best) { best = cost; seg_len = sl; }
        }
    }
    const int64_t tb = (int64_t)SB_GROUPS * seg_len;
    const dim3 grid((unsigned)((T + tb - 1) / tb));
    unsigned perm_mul = 1;
    static const unsigned primes[] = {7919u, 7907u, 7901u, 7883u, 7879u, 7877u};
    for (unsigned pm : primes)
        if (grid.x % pm != 0) { perm_mul = pm; break; }
    const bool grad = gxn != nullptr;
    const int LD = E + 8;
    const size_t slab_bytes = (size_t)(SB_THREADS / 64) * (16 * LD + 16) * sizeof(float);     // negative-row rings + ids
    // one 1024-thread workgroup per CU (forcing 8 waves/SIMD spills: measured 2.4x slower)
    // 156 of the CU's 160 KB (gfx950): 731 table rows at E = 32 instead of 556 at the 128 KB this was sized for -- fewer positives
    // take the global-atomic route (-1.2 % kernel time on the bench's imbalanced labels)
    // The budget is clamped to what the device offers per workgroup (queried once; gfx950: 160 KB) minus the kernel's static
    // LDS, and to at least the rings plus one table row (HPCS_SAMPLED_LDS_KB was used unchecked).
    static const size_t lds_budget = [] {
        int dev = 0, cap = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || cap <= 0)
            cap = 64 * 1024;
        const size_t room = (size_t)cap > 4096 ? (size_t)cap - 4096 : (size_t)cap;      // static LDS of the kernel + slack
        const int kb = hpcs_tuning().sampled_lds_kb;
        size_t want = (size_t)(kb > 0 ? kb : 156) * 1024;
        return want < room ? want : room;
    }();
    const size_t row_bytes = (size_t)(E + 8) * sizeof(float);
    int RC = 0;                                                                          // rows of the positive table
    if (grad) {
        RC = lds_budget > slab_bytes + row_bytes ? (int)((lds_budget - slab_bytes) / row_bytes) : 1;
        if (RC > 1024) RC = 1024;
    }
    const size_t lds = (size_t)RC * row_bytes + slab_bytes;
#define HPCS_SAMPLED_LAUNCH(KERNEL, EPL)                                                                              \
        if (grad) {                                                                                                   \
            if (hipFuncSetAttribute((const void*)KERNEL<EPL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
                (void)hipGetLastError();                                                                              \
                return HPCS_ERR_UNSUPPORTED;                                                                          \
            }                                                                                                         \
            hipLaunchKernelGGL((KERNEL<EPL, true>), grid, dim3(SB_THREADS), lds, ST, xn, labels, order,               \
                               offsets, cls_toff, cls_k, C, M, T, seed, seed_dev, scale, temperature, margin, kind, RC, loss_sum, \
                               (unsigned long long*)kept, gxn, gscale, a_dbg, p_dbg, n_dbg, perm_mul, seg_len);                \
        } else {                                                                                                      \
            hipLaunchKernelGGL((KERNEL<EPL, false>), grid, dim3(SB_THREADS), lds, ST, xn, labels, order,              \
                               offsets, cls_toff, cls_k, C, M, T, seed, seed_dev, scale, temperature, margin, kind, RC, loss_sum, \
                               (unsigned long long*)kept, gxn, gscale, a_dbg, p_dbg, n_dbg, perm_mul, seg_len);                \
        }
#define HPCS_SAMPLED_CASE(EPL)                                                                                        \
    if (E == 8 * EPL) {                                                                                               \
        if (variant == 8) { HPCS_SAMPLED_LAUNCH(hyphc_sampled_kernel, EPL) }                                          \
        else { HPCS_SAMPLED_LAUNCH(hyphc_sampled4_kernel, EPL) }                                                      \
    }
    HPCS_SAMPLED_CASE(2)
    HPCS_SAMPLED_CASE(4)
    HPCS_SAMPLED_CASE(8)
#undef HPCS_SAMPLED_CASE
#undef HPCS_SAMPLED_LAUNCH
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
