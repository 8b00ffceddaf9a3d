// Test-time decode: complete-linkage dendrogram of every cloud's leaf embeddings and the
// cluster count that best matches the ground-truth parts.
//
//   hpcs_complete_linkage   replaces  BaseSimilarityHypHC._decode_linkage        hpcs/models/base_hyp_hc.py:81-86
//                           ( = scipy.cluster.hierarchy.linkage(leaves, method='complete', metric='cosine') per cloud )
//   hpcs_optimal_k          replaces  get_optimal_k(y, Z, 'iou' | 'ri')          hpcs/utils/scores.py:141-177
//                           ( = scipy fcluster(Z, k, 'maxclust') for k = 1..K_true+4 + the IoU / adjusted-Rand scoring )
//
// The arithmetic is third-party (scipy 1.15.3, scikit-learn 1.7.2; both are in the image, so the tests compare with the
// real routines).  It is integer / order work around ONE floating-point stage, the fp64 cosine distance, whose rounding
// sequence is reproduced exactly (this file is compiled with -ffp-contract=off):
//     dot(u, v): even and odd features accumulate in two lanes, lanes added, an odd last feature added after
//                (the 2-wide vectorised loop of the scipy wheel -- oracle/decode.py:_dot2);
//     d = 1 - clip(dot(u,v) / (sqrt(dot(u,u)) * sqrt(dot(v,v))), -1, 1).
// Every later step only compares, takes maxima of, and copies these doubles, so the linkage matrix is bit-identical.
//
// Mapping.  A cloud is one workgroup of 1024 threads (16 waves); B clouds run side by side on B CUs.  The distance
// matrix of a cloud (N*N fp64, 8 MB at N=1024) lives in HBM/L2 as a FULL symmetric matrix so that the chain's row scans
// are coalesced 8 KB reads.  The nearest-neighbour chain is latency-bound by construction (about 3N dependent row
// scans + N row merges per cloud): each step is one row read, a 64-lane DPP/shuffle arg-min on (distance bits, index)
// keys, one LDS exchange between the 16 waves and ONE barrier.  Distances are compared as their uint64 bit patterns
// (non-negative doubles order like integers), which also makes the loop terminate on NaN input (flagged in `status`).
#include "common.h"

namespace {

constexpr int DEC_T = 1024;        // threads per cloud
constexpr int DEC_MAXN = 2048;     // points per cloud (static LDS plans below)
constexpr int DEC_ITEMS = DEC_MAXN / DEC_T;
constexpr int OK_MAXT = 64;        // ground-truth parts per cloud
constexpr int OK_MAXP = OK_MAXT + 4;
constexpr int OK_MAXLAB = 1024;    // label values are in [0, num_labels), num_labels <= 1024

__device__ __forceinline__ double dot2(const float* __restrict__ u, const float* __restrict__ v, int E) {
    double a0 = 0.0, a1 = 0.0;
    int e = 0;
    for (; e + 1 < E; e += 2) {
        a0 = a0 + (double)u[e] * (double)v[e];
        a1 = a1 + (double)u[e + 1] * (double)v[e + 1];
    }
    double s = a0 + a1;
    if (e < E) s = s + (double)u[e] * (double)v[e];
    return s;
}

__global__ void leaf_norms_kernel(const float* __restrict__ X, int E, long rows, double* __restrict__ nrm) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) nrm[r] = sqrt(dot2(X + r * E, X + r * E, E));
}

// D[b][i][j] for a 32x32 tile per block of 256 threads; rows of both panels staged in LDS as doubles.
constexpr int CM_TILE = 32, CM_EC = 32;
__global__ __launch_bounds__(256) void cosine_matrix_kernel(const float* __restrict__ X, const double* __restrict__ nrm,
                                                           int N, int E, double* __restrict__ Dall,
                                                           int* __restrict__ status) {
    __shared__ double sa[CM_TILE][CM_EC + 1], sb[CM_TILE][CM_EC + 1];
    const int b = blockIdx.z, i0 = blockIdx.y * CM_TILE, j0 = blockIdx.x * CM_TILE;
    const int tj = threadIdx.x & 31, ti = threadIdx.x >> 5;       // rows ti, ti+8, ti+16, ti+24
    const float* Xb = X + (size_t)b * N * E;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, tail[4] = {0, 0, 0, 0};
    for (int e0 = 0; e0 < E; e0 += CM_EC) {
        const int ec = min(CM_EC, E - e0);
        __syncthreads();
        for (int t = threadIdx.x; t < CM_TILE * CM_EC; t += 256) {
            const int r = t / CM_EC, c = t % CM_EC;
            double va = 0.0, vb = 0.0;
            if (c < ec) {
                if (i0 + r < N) va = (double)Xb[(size_t)(i0 + r) * E + e0 + c];
                if (j0 + r < N) vb = (double)Xb[(size_t)(j0 + r) * E + e0 + c];
            }
            sa[r][c] = va;
            sb[r][c] = vb;
        }
        __syncthreads();
        for (int c = 0; c < ec; c += 2) {
            const bool pair = e0 + c + 1 < E;
            const double v0 = sb[tj][c], v1 = sb[tj][c + 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double u0 = sa[ti + 8 * r][c], u1 = sa[ti + 8 * r][c + 1];
                if (pair) {
                    a0[r] = a0[r] + u0 * v0;
                    a1[r] = a1[r] + u1 * v1;
                } else {
                    tail[r] = u0 * v0;          // odd E: the last feature joins after the lane sum
                }
            }
        }
    }
    const int j = j0 + tj;
    if (j >= N) return;
    const double nj = nrm[(size_t)b * N + j];
    double* D = Dall + (size_t)b * N * N;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ti + 8 * r;
        if (i >= N) continue;
        double s = a0[r] + a1[r];
        if (E & 1) s = s + tail[r];
        double c = s / (nrm[(size_t)b * N + i] * nj);
        if (fabs(c) > 1.0) c = copysign(1.0, c);
        const double d = i == j ? 0.0 : 1.0 - c;
        bad |= !(d == d) || fabs(d) > 1e300;
        D[(size_t)i * N + j] = d;
    }
    if (bad) atomicOr(status + b, 1);
}

struct Key {
    unsigned long long d;
    unsigned id;
};
__device__ __forceinline__ bool key_less(const Key& a, const Key& b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }
__device__ __forceinline__ Key wave_key_min(Key k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Key q;
        q.d = __shfl_xor(k.d, o, 64);
        q.id = __shfl_xor(k.id, o, 64);
        if (key_less(q, k)) k = q;
    }
    return k;
}

// Nearest-neighbour chain, complete linkage (scipy/cluster/_hierarchy.pyx nn_chain with new_dist = max; restated in
// oracle/decode.py:nn_chain_complete).  D is consumed.  Output: the n-1 merges in chain order.
__global__ __launch_bounds__(DEC_T) void nn_chain_kernel(double* __restrict__ Dall, int N, int* __restrict__ mx_out,
                                                        int* __restrict__ my_out, double* __restrict__ mh_out,
                                                        int* __restrict__ ms_out, int* __restrict__ status) {
    __shared__ int size[DEC_MAXN];
    __shared__ int chain[DEC_MAXN];
    __shared__ unsigned long long sk[2][DEC_T / 64];
    __shared__ unsigned si[2][DEC_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x;
    unsigned long long* D = reinterpret_cast<unsigned long long*>(Dall + (size_t)b * N * N);   // bit patterns
    for (int i = tid; i < N; i += DEC_T) size[i] = 1;
    __syncthreads();
    int clen = 0, first = 0, parity = 0;
    int budget = 8 * N + 64;       // the chain makes < 3N scans in total; a uniform bound so that no input can hang the GPU
    for (int k = 0; k < N - 1; ++k) {
        int x, prev;
        if (clen == 0) {
            while (first < N - 1 && size[first] == 0) ++first;   // the lowest live slot only moves up
            x = first;
            prev = -1;
            if (tid == 0) chain[0] = x;
            clen = 1;
        } else {
            x = chain[clen - 1];
            prev = clen > 1 ? chain[clen - 2] : -1;
        }
        int y;
        unsigned long long cur;
        while (true) {
            if (--budget < 0) {                           // uniform across the block: every thread holds the same state
                if (tid == 0) atomicOr(status + b, 4);
                return;
            }
            // arg-min over the live slots of row x; the previous chain element wins ties (id 0 sorts first), otherwise
            // the lowest index does -- exactly the sequential scan "start from prev, replace on strictly smaller"
            Key best{~0ull, ~0u};
            const unsigned long long* row = D + (size_t)x * N;
#pragma unroll
            for (int q = 0; q < DEC_ITEMS; ++q) {
                const int i = tid + q * DEC_T;
                if (i < N && i != x && size[i] > 0) {
                    Key c{row[i], i == prev ? 0u : (unsigned)i + 1u};
                    if (key_less(c, best)) best = c;
                }
            }
            best = wave_key_min(best);
            if (lane == 0) {
                sk[parity][wv] = best.d;
                si[parity][wv] = best.id;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < DEC_T / 64; ++w) {
                Key c{sk[parity][w], si[parity][w]};
                if (key_less(c, best)) best = c;
            }
            parity ^= 1;
            y = best.id == 0u ? prev : (int)best.id - 1;
            cur = best.d;
            if (clen > 1 && y == prev) break;
            if (tid == 0) chain[clen] = y;
            ++clen;
            prev = x;
            x = y;
        }
        clen -= 2;
        const int mx = min(x, y), my = max(x, y);
        // row/column `my` becomes the merged cluster: D(i, my) = max(D(i, mx), D(i, my))
        const unsigned long long* rx = D + (size_t)mx * N;
        unsigned long long* ry = D + (size_t)my * N;
#pragma unroll
        for (int q = 0; q < DEC_ITEMS; ++q) {
            const int i = tid + q * DEC_T;
            if (i < N && i != mx && i != my && size[i] > 0) {
                const unsigned long long a = rx[i], c = ry[i];
                const unsigned long long v = a > c ? a : c;
                ry[i] = v;
                D[(size_t)i * N + my] = v;
            }
        }
        __syncthreads();                      // distance updates visible; every thread is done reading size[]
        if (tid == 0) {
            const int nx = size[mx], ny = size[my];
            size[mx] = 0;
            size[my] = nx + ny;
            const size_t o = (size_t)b * N + k;
            mx_out[o] = mx;
            my_out[o] = my;
            mh_out[o] = __longlong_as_double((long long)cur);
            ms_out[o] = nx + ny;
        }
        __syncthreads();
    }
}

// Stable sort of the merges by height (bitonic on (height bits, chain position) keys) and scipy's `label` pass
// (union-find relabelling: row i creates cluster n + i).  One workgroup per cloud.
__global__ __launch_bounds__(DEC_T) void sort_label_kernel(const int* __restrict__ mx_in, const int* __restrict__ my_in,
                                                          const double* __restrict__ mh_in, int N,
                                                          double* __restrict__ Zall) {
    __shared__ unsigned long long key[DEC_MAXN];
    __shared__ int idx[DEC_MAXN];
    __shared__ int parent[2 * DEC_MAXN];
    __shared__ int usz[2 * DEC_MAXN];
    const int tid = threadIdx.x, b = blockIdx.x, M = N - 1;
    int P = 1;
    while (P < M) P <<= 1;
    const size_t o = (size_t)b * N;
    for (int i = tid; i < P; i += DEC_T) {
        key[i] = i < M ? (unsigned long long)__double_as_longlong(mh_in[o + i]) : ~0ull;
        idx[i] = i < M ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += DEC_T) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long ka = key[lo], kb = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool gt = ka > kb || (ka == kb && ia > ib);
                if (gt == up) {
                    key[lo] = kb; key[hi] = ka;
                    idx[lo] = ib; idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    double* Z = Zall + (size_t)b * M * 4;
    for (int i = tid; i < 2 * N - 1; i += DEC_T) {
        parent[i] = i;
        usz[i] = 1;
    }
    for (int i = tid; i < M; i += DEC_T) Z[(size_t)i * 4 + 2] = __longlong_as_double((long long)key[i]);
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < M; ++i) {
            const int src = idx[i];
            int xr = mx_in[o + src], yr = my_in[o + src];
            int p = xr;
            while (parent[xr] != xr) xr = parent[xr];
            while (parent[p] != xr) { const int nx = parent[p]; parent[p] = xr; p = nx; }
            p = yr;
            while (parent[yr] != yr) yr = parent[yr];
            while (parent[p] != yr) { const int nx = parent[p]; parent[p] = yr; p = nx; }
            const int s = usz[xr] + usz[yr];
            parent[xr] = parent[yr] = N + i;
            usz[N + i] = s;
            // reuse key[] as the output staging: (min root, max root) packed, size in idx[]
            key[i] = ((unsigned long long)(unsigned)min(xr, yr) << 32) | (unsigned)max(xr, yr);
            idx[i] = s;
        }
    }
    __syncthreads();
    for (int i = tid; i < M; i += DEC_T) {
        Z[(size_t)i * 4 + 0] = (double)(unsigned)(key[i] >> 32);
        Z[(size_t)i * 4 + 1] = (double)(unsigned)(key[i] & 0xffffffffu);
        Z[(size_t)i * 4 + 3] = (double)idx[i];
    }
}

// get_optimal_k: for k = 1 .. K_true + 4 cut the dendrogram into at most k flat clusters with scipy's numbering, score
// the cut against the ground truth, keep the best (first k on ties).  One workgroup per cloud; integer work in LDS.
// The cut: Z is sorted by height, so "height <= cutoff" is "row <= rmax(k)".  The numbering (cluster_monocrit): an Euler
// tour over the internal nodes; a flat cluster is created when its top node is entered (key 4*enter), a singleton leaf
// when its parent is left (key 4*exit + 1|2); ids are ranks of the distinct keys = a popcount prefix over a bitmap.
__global__ __launch_bounds__(DEC_T) void optimal_k_kernel(const double* __restrict__ Zall, const int64_t* __restrict__ yall,
                                                         int N, int num_labels, int index_mode,
                                                         int* __restrict__ best_pred, int* __restrict__ best_k_out,
                                                         double* __restrict__ best_score_out, int* __restrict__ status) {
    __shared__ unsigned short left[DEC_MAXN], parent[2 * DEC_MAXN], cnt[DEC_MAXN], enter[DEC_MAXN];
    __shared__ unsigned char ytrue[DEC_MAXN];
    __shared__ unsigned short labrank[OK_MAXLAB];
    __shared__ unsigned bitmap[(8 * DEC_MAXN) / 32];
    __shared__ int wprefix[(8 * DEC_MAXN) / 32];
    __shared__ int cont[OK_MAXT][OK_MAXP];
    __shared__ int ct[OK_MAXT], cp[OK_MAXP], ind[OK_MAXT], remap[OK_MAXP];
    __shared__ int s_rmax, s_m, s_kt, s_err;
    __shared__ int wsum[DEC_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x, M = N - 1;
    const double* Z = Zall + (size_t)b * M * 4;
    const int64_t* y = yall + (size_t)b * N;
    const int ROOT = 2 * N - 2;
    constexpr int NWORDS = (8 * DEC_MAXN) / 32;

    // ---- remap_labels: rank of each label among the labels present ----
    for (int i = tid; i < OK_MAXLAB; i += DEC_T) labrank[i] = 0;
    if (tid == 0) { s_err = 0; s_kt = 0; }
    for (int i = tid; i < OK_MAXT * OK_MAXP; i += DEC_T) (&cont[0][0])[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += DEC_T) {
        const int64_t v = y[i];
        if (v < 0 || v >= num_labels) s_err = 2;
        else labrank[v] = 1;
    }
    __syncthreads();
    if (tid == 0) {
        int r = 0;
        for (int v = 0; v < num_labels; ++v)
            if (labrank[v]) labrank[v] = (unsigned short)(++r);      // rank + 1
        s_kt = r;
        if (r > OK_MAXT || r + 4 >= N) s_err = 3;
    }
    __syncthreads();
    const int kt = s_kt;
    if (s_err) {
        if (tid == 0) { status[b] = s_err; best_k_out[b] = 0; best_score_out[b] = 0.0; }
        for (int i = tid; i < N; i += DEC_T) best_pred[(size_t)b * N + i] = -1;
        return;
    }
    for (int i = tid; i < OK_MAXT; i += DEC_T) ct[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += DEC_T) {
        const int t = labrank[y[i]] - 1;
        ytrue[i] = (unsigned char)t;
        atomicAdd(&ct[t], 1);
    }
    // ---- tree arrays ----
    for (int r = tid; r < M; r += DEC_T) {
        const int l = (int)Z[(size_t)r * 4 + 0], rr = (int)Z[(size_t)r * 4 + 1];
        left[r] = (unsigned short)l;
        cnt[r] = (unsigned short)(int)Z[(size_t)r * 4 + 3];
        parent[l] = (unsigned short)(N + r);
        parent[rr] = (unsigned short)(N + r);
    }
    if (tid == 0) parent[ROOT] = 0xffff;
    __syncthreads();
    // enter time of every internal node: walk up, adding this child's offset inside its parent
    for (int r = tid; r < M; r += DEC_T) {
        int t = 0, v = N + r;
        while (v != ROOT) {
            const int p = parent[v], pr = p - N, l = left[pr];
            t += 1;
            if (l != v && l >= N) t += 2 * ((int)cnt[l - N] - 1);   // right child: the whole left internal subtree first
            v = p;
        }
        enter[r] = (unsigned short)t;
    }
    __syncthreads();

    int node[DEC_ITEMS], bestT[DEC_ITEMS], T[DEC_ITEMS];
#pragma unroll
    for (int q = 0; q < DEC_ITEMS; ++q) { node[q] = tid + q * DEC_T; bestT[q] = -1; T[q] = 0; }
    int best_k = 0;
    double best_score = 0.0;
    const double n_d = (double)N;
    for (int k = kt + 4; k >= 1; --k) {
        // ---- rmax: last row whose height equals the cut height h[n-k-1] ----
        const int c = N - k - 1;
        const double hc = Z[(size_t)c * 4 + 2];
        if (tid == 0) { s_rmax = c; s_m = 0; }
        for (int i = tid; i < NWORDS; i += DEC_T) bitmap[i] = 0;
        for (int i = tid; i < OK_MAXP; i += DEC_T) { cp[i] = 0; remap[i] = 0; }
        __syncthreads();
        for (int r = c + 1 + tid; r < M; r += DEC_T)
            if (Z[(size_t)r * 4 + 2] == hc) atomicMax(&s_rmax, r);
        __syncthreads();
        const int rmax = s_rmax;
        // ---- each leaf climbs to the top node of its flat cluster (cuts only rise as k falls) ----
        int keyv[DEC_ITEMS];
#pragma unroll
        for (int q = 0; q < DEC_ITEMS; ++q) {
            const int i = tid + q * DEC_T;
            keyv[q] = -1;
            if (i < N) {
                int v = node[q];
                while (v != ROOT && (int)parent[v] - N <= rmax) v = parent[v];
                node[q] = v;
                if (v == i) {
                    const int pr = (int)parent[i] - N;
                    const int ex = (int)enter[pr] + 2 * ((int)cnt[pr] - 1) - 1;
                    keyv[q] = 4 * ex + (left[pr] == i ? 1 : 2);
                } else {
                    keyv[q] = 4 * (int)enter[v - N];
                }
                atomicOr(&bitmap[keyv[q] >> 5], 1u << (keyv[q] & 31));
            }
        }
        __syncthreads();
        // ---- popcount prefix over the bitmap words (NWORDS <= DEC_T: one word per thread) ----
        {
            const int pc = tid < NWORDS ? __popc(bitmap[tid]) : 0;
            int incl = pc;
#pragma unroll
            for (int o2 = 1; o2 < 64; o2 <<= 1) {
                const int t = __shfl_up(incl, o2, 64);
                if (lane >= o2) incl += t;
            }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wv; ++w) base += wsum[w];
            if (tid < NWORDS) wprefix[tid] = base + incl - pc;        // exclusive
        }
        __syncthreads();
        int kp = 0;
        for (int w = 0; w < DEC_T / 64; ++w) kp += wsum[w];
        // ---- cluster ids and the contingency table ----
#pragma unroll
        for (int q = 0; q < DEC_ITEMS; ++q) {
            const int i = tid + q * DEC_T;
            if (i < N) {
                const int kv = keyv[q];
                T[q] = wprefix[kv >> 5] + __popc(bitmap[kv >> 5] & ((1u << (kv & 31)) - 1u));   // 0-based = fcluster - 1
                atomicAdd(&cont[ytrue[i]][T[q]], 1);
                atomicAdd(&cp[T[q]], 1);
            }
        }
        __syncthreads();
        double score = 0.0;
        if (index_mode == 0) {
            // IoU matrix in float32 (a torch.zeros matrix in the reference), first arg-max per true part
            if (tid < kt) {
                float bv = -1.f;
                int bj = 0;
                for (int j = 0; j < kp; ++j) {
                    const int cij = cont[tid][j];
                    const float v = (float)((double)cij / (double)(ct[tid] + cp[j] - cij));
                    if (v > bv) { bv = v; bj = j; }
                }
                ind[tid] = bj;
            }
            __syncthreads();
            if (tid < kt) atomicMax(&remap[ind[tid]], tid + 1);       // later parts overwrite earlier ones
            __syncthreads();
            int mloc = 0;
#pragma unroll
            for (int q = 0; q < DEC_ITEMS; ++q) {
                const int i = tid + q * DEC_T;
                if (i < N && remap[T[q]] == (int)ytrue[i] + 1) ++mloc;
            }
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) mloc += __shfl_xor(mloc, o2, 64);
            if (lane == 0 && mloc) atomicAdd(&s_m, mloc);
            __syncthreads();
            const double m = (double)s_m;
            score = m / (2.0 * n_d - m);           // |and| / |or| of the two one-hot matrices
        } else {
            // adjusted Rand index from the pair confusion matrix (sklearn adjusted_rand_score), integers until the end
            long long ss = 0, sk2 = 0, sc2 = 0;
            for (int i = 0; i < kt; ++i) {
                sc2 += (long long)ct[i] * ct[i];
                for (int j = 0; j < kp; ++j) ss += (long long)cont[i][j] * cont[i][j];
            }
            for (int j = 0; j < kp; ++j) sk2 += (long long)cp[j] * cp[j];
            const long long nn = N;
            const long long tp = ss - nn, fp = sk2 - ss, fn = sc2 - ss, tn = nn * nn - fp - fn - ss;
            if (fn == 0 && fp == 0) score = 1.0;
            else score = 2.0 * (double)(tp * tn - fn * fp) / (double)((tp + fn) * (fn + tn) + (tp + fp) * (fp + tn));
        }
        // ascending-k semantics with a strict '>' = the smallest k among the best scores; k runs downwards here
        if (score > 0.0 && score >= best_score) {
            best_score = score;
            best_k = k;
#pragma unroll
            for (int q = 0; q < DEC_ITEMS; ++q) bestT[q] = T[q];
        }
        __syncthreads();
        // clear the table for the next k
        for (int i = tid; i < OK_MAXT * OK_MAXP; i += DEC_T) (&cont[0][0])[i] = 0;
    }
#pragma unroll
    for (int q = 0; q < DEC_ITEMS; ++q) {
        const int i = tid + q * DEC_T;
        if (i < N) best_pred[(size_t)b * N + i] = bestT[q];
    }
    if (tid == 0) {
        best_k_out[b] = best_k;
        best_score_out[b] = best_score;
        status[b] = 0;
    }
}

struct DecodeWs {
    size_t d, nrm, mh, mx, my, ms, total;
};
DecodeWs decode_ws(int B, int N) {
    DecodeWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    w.d = take((size_t)B * N * N * 8);
    w.nrm = take((size_t)B * N * 8);
    w.mh = take((size_t)B * N * 8);
    w.mx = take((size_t)B * N * 4);
    w.my = take((size_t)B * N * 4);
    w.ms = take((size_t)B * N * 4);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t hpcs_complete_linkage_workspace(int B, int N) {
    if (B <= 0 || N <= 1) return 0;
    return decode_ws(B, N).total;
}

extern "C" int hpcs_complete_linkage(const float* leaves, int B, int N, int E, double* Z, int32_t* status, void* ws,
                                     size_t ws_bytes, hpcs_stream_t stream) {
    if (!leaves || !Z || !status || !ws) return HPCS_ERR_NULL;
    if (B <= 0 || N < 2 || E <= 0) return HPCS_ERR_SHAPE;
    if (N > DEC_MAXN) return HPCS_ERR_UNSUPPORTED;
    const DecodeWs w = decode_ws(B, N);
    if (ws_bytes < w.total) return HPCS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)ws;
    double* D = (double*)(base + w.d);
    double* nrm = (double*)(base + w.nrm);
    if (const int zr = hpcs_zero_async(s, status, sizeof(int32_t) * B)) return zr;
    const long rows = (long)B * N;
    leaf_norms_kernel<<<hpcs_cdiv(rows, 256), 256, 0, s>>>(leaves, E, rows, nrm);
    HPCS_CHECK_LAUNCH();
    const int tiles = hpcs_cdiv(N, CM_TILE);
    cosine_matrix_kernel<<<dim3(tiles, tiles, B), 256, 0, s>>>(leaves, nrm, N, E, D, status);
    HPCS_CHECK_LAUNCH();
    nn_chain_kernel<<<B, DEC_T, 0, s>>>(D, N, (int*)(base + w.mx), (int*)(base + w.my), (double*)(base + w.mh),
                                        (int*)(base + w.ms), status);
    HPCS_CHECK_LAUNCH();
    sort_label_kernel<<<B, DEC_T, 0, s>>>((const int*)(base + w.mx), (const int*)(base + w.my),
                                          (const double*)(base + w.mh), N, Z);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_optimal_k(const double* Z, const int64_t* y, int B, int N, int num_labels, int index_mode,
                              int32_t* best_pred, int32_t* best_k, double* best_score, int32_t* status,
                              hpcs_stream_t stream) {
    if (!Z || !y || !best_pred || !best_k || !best_score || !status) return HPCS_ERR_NULL;
    if (B <= 0 || N < 6 || num_labels <= 0) return HPCS_ERR_SHAPE;
    if (N > DEC_MAXN || num_labels > OK_MAXLAB || (index_mode != 0 && index_mode != 1)) return HPCS_ERR_UNSUPPORTED;
    optimal_k_kernel<<<B, DEC_T, 0, (hipStream_t)stream>>>(Z, y, N, num_labels, index_mode, best_pred, best_k, best_score,
                                                          status);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
