// a1: fused pairwise-score + per-row top-k (no N x N matrix in HBM).
//
// Replaces knn() of hpcs/nn/dgcnn/utils/vn_dgcnn_util.py:4-10.  Bit-defined arithmetic
// (must equal oracle/knn_oracle.c, which equals torch-CPU on the build image):
//   gram   = fmaf chain over d (k-ordered, from +0)
//   sq(j)  = cascade sum of squares (groups of 16, see oracle)
//   score  = ((-sq_j) - (-2*gram)) - sq_i ;  order = (score desc, index asc)
// This file is compiled with -ffp-contract=off so only the explicit fmaf's fuse.
//
// Layout / mapping: one workgroup = one cloud x QT queries (one query per thread).  The cloud's
// [D,N] slab is walked in candidate tiles of TC points (256 for D=3, 128 otherwise) staged in LDS as [D+1][TC] (row D = sq):
// global reads are coalesced along N, LDS reads in the scoring loop are wave-uniform broadcasts.
// Each thread keeps its sorted top-KK list in registers (static indexing only).
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

extern "C" int hpcs_knn(const float* x, int B, int D, int N, int k, int64_t* idx, hpcs_stream_t stream) {
    if (!x || !idx) return HPCS_ERR_NULL;
    if (B < 0 || D <= 0 || N <= 0 || k <= 0 || k > N) return HPCS_ERR_SHAPE;
    if (k > 64) return HPCS_ERR_UNSUPPORTED;
    if (B == 0) return HPCS_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (D == 3)
        rc = launch_knn<3, 256>(x, B, D, N, k, idx, st);
    else if (D == 63)
        rc = launch_knn<63, 128>(x, B, D, N, k, idx, st);
    else
        rc = launch_knn<0, 128>(x, B, D, N, k, idx, st);
    if (rc != HPCS_OK) return rc;
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
