// fp32-accurate channel-mixing GEMMs on the BF16 matrix cores ("bf16x3").
//
// gfx950's fp32-input MFMA runs at the vector rate (157 TFLOP/s); the bf16 MFMA is 16x faster.  An fp32 number splits
// EXACTLY into three bf16 pieces, a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 3 x 8 = 24
// mantissa bits; the two subtractions are exact), a product of two pieces is exact in fp32, and the MFMA accumulates in fp32.
// So   a.b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 |a||b|)
// -- six bf16 MFMAs per fp32 product at 6/16 of the fp32 MFMA's cost.  The dropped terms (a2 b3, a3 b2, a3 b3) are below
// 2^-23 |a||b|, i.e. below the rounding of the fp32 accumulation that every fp32 GEMM (the reference's MKL sgemm, this
// library's fma-chain kernels) performs anyway; tests hold the result to the fp64 product as tightly as the fma-chain kernel.
// kNN keeps its bit-defined fp32 MFMA chain (csrc/knn.hip): only the dense VN layers / heads take this path.
//
// Replaces, like csrc/gemm.hip: nn.Linear-over-channels (map_to_feat / map_to_dir, vn_layers.py:14-24,48-77) and the
// Conv1d(k=1) heads (vn_dgcnn_partseg.py:40-57), forward and data gradient (hpcs_gemm_tn_x3: same arguments as hpcs_gemm_tn).
//
// Tile 128 x 128 x 16, four waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16 tiles).  Both operands are K-major in memory
// ([k][m], [k][n]); the MFMA wants 8 consecutive k per lane.  The loader thread owns one (row, 8 k) strip -- eight coalesced
// 4-byte loads, one per k -- splits it in registers and writes one 16-byte [row][k8] piece per split level to LDS (row pitch
// 48 bytes: conflict-free for the 16-byte fragment reads), so the transposition costs nothing and no split copy of any tensor
// ever exists in HBM.  Register-staged prefetch of the next k-step, LDS double-buffered, one barrier per step.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 16;
constexpr int X3_PITCH = 48;                        // bytes per [row][16 k] LDS row: 32 data + 16 pad (3 bank quads: odd)
constexpr int X3_PIECE = X3_BM * X3_PITCH;          // one split level of one operand tile: 6 KB
constexpr int X3_OPER = 3 * X3_PIECE;               // 18 KB
constexpr int X3_STAGE = 2 * X3_OPER;               // A + B: 36 KB
#ifndef X3_DB
#define X3_DB 0                                      // 1: double-buffered LDS (72 KB, two workgroups per CU, one barrier per step)
#endif
constexpr int X3_NBUF = X3_DB ? 2 : 1;
constexpr int X3_WPS = X3_DB ? 2 : 3;               // waves per SIMD the register budget is set for

// two fp32 -> packed pair of bf16 (round to nearest even), and back
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }

// eight fp32 (one row, consecutive k) -> three 16-byte pieces
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        const unsigned h = pk_bf16(a, b);
        const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);          // exact
        const unsigned m = pk_bf16(ra, rb);
        const float sa = ra - bf16_lo(m), sb = rb - bf16_hi(m);        // exact
        const unsigned l = pk_bf16(sa, sb);
        p1[q] = h;
        p2[q] = m;
        p3[q] = l;
    }
}

__global__ __launch_bounds__(256, X3_WPS) void gemm_tn_x3_kernel(
    const float* __restrict__ At, int lda, const float* __restrict__ Bm, int ldb, int64_t sBo, int64_t sBi,
    float* __restrict__ C, int ldc, int64_t sCo, int64_t sCi, int M, int Nc, int K, int n_inner,
    const float* __restrict__ rowbias, int accumulate, int tilesM, int tilesN, unsigned total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
    const unsigned logical = hpcs_xcd_remap(blockIdx.x, total);
    const int rowTile = logical % tilesM;
    const unsigned t1 = logical / tilesM;
    const int colTile = t1 % tilesN;
    const int batch = t1 / tilesN;
    const int bo = batch / n_inner, bi = batch % n_inner;
    const float* Bp = Bm + bo * sBo + bi * sBi;
    float* Cp = C + bo * sCo + bi * sCi;
    const float* bias = rowbias ? rowbias + (int64_t)batch * M : nullptr;
    const int m0 = rowTile * X3_BM, n0 = colTile * X3_BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // ---- loader: thread = (row, k half); rows beyond the matrix are clamped (their outputs are never stored)
    const int lrow = tid & 127, lh = tid >> 7;
    const float* Ag = At + min(m0 + lrow, M - 1);
    const float* Bg = Bp + min(n0 + lrow, Nc - 1);
    float ra[8], rb[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = min(k0 + 8 * lh + j, K - 1);                     // clamped; zeroed in lstore
            ra[j] = Ag[(int64_t)k * lda];
            rb[j] = Bg[(int64_t)k * ldb];
        }
    };
    auto lstore = [&](int buf, int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + 8 * lh + j >= K) { ra[j] = 0.f; rb[j] = 0.f; }
        u32x4 p1, p2, p3;
        unsigned char* base = x3_lds + buf * X3_STAGE + lrow * X3_PITCH + lh * 16;
        split8(ra, p1, p2, p3);
        *reinterpret_cast<u32x4*>(base) = p1;
        *reinterpret_cast<u32x4*>(base + X3_PIECE) = p2;
        *reinterpret_cast<u32x4*>(base + 2 * X3_PIECE) = p3;
        split8(rb, p1, p2, p3);
        *reinterpret_cast<u32x4*>(base + X3_OPER) = p1;
        *reinterpret_cast<u32x4*>(base + X3_OPER + X3_PIECE) = p2;
        *reinterpret_cast<u32x4*>(base + X3_OPER + 2 * X3_PIECE) = p3;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bool live[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) live[i][j] = (m0 + wm * 64 + i * 32 < M) && (n0 + wn * 64 + j * 32 < Nc);

    const int lr = lane & 31, lk = lane >> 5;
    const int fa = (wm * 64 + lr) * X3_PITCH + lk * 16;          // fragment offsets inside a piece
    const int fb = (wn * 64 + lr) * X3_PITCH + lk * 16;
    const int nk = (K + X3_BK - 1) / X3_BK;
    gload(0);
    lstore(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = X3_DB ? (kt & 1) : 0;
        if (kt + 1 < nk) gload((kt + 1) * X3_BK);
        const unsigned char* sa = x3_lds + buf * X3_STAGE + fa;
        const unsigned char* sb = sa - fa + X3_OPER + fb;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(sa + p * X3_PIECE + i * 32 * X3_PITCH);
                b[i][p] = *reinterpret_cast<const bf16x8*>(sb + p * X3_PIECE + i * 32 * X3_PITCH);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (live[i][j]) {
                    // smallest terms first: a3 b1 + a2 b2 + a1 b3, then a2 b1 + a1 b2, then a1 b1
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        if (!X3_DB) __syncthreads();                          // single stage: everybody has read its fragments
        if (kt + 1 < nk) lstore(X3_DB ? (buf ^ 1) : 0, (kt + 1) * X3_BK);
        __syncthreads();
    }
    // ---- epilogue: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < M && col < Nc) {
                    float v = acc[i][j][r];
                    if (bias) v += bias[row];
                    float* dst = Cp + (int64_t)row * ldc + col;
                    if (accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

// ---- weight gradients of the large layers: dW (M x Kx) = sum_b G[b] (M x Nc) X[b]^T (Kx x Nc), split over the column steps.
// Both operands have the contraction index contiguous in memory, so the loader thread takes eight consecutive columns of
// one row (two 16-byte loads), splits them and writes the three [row][k8] pieces; everything else is the tile above.  Same
// work split as gemm_nt_big_kernel (csrc/gemm.hip): ONE list of 16-column steps over all batches cut into equal ranges, plain
// partials [split][M][Kx] summed by split_reduce.
__device__ __forceinline__ void ld8_guard(const float* __restrict__ rowp, int col, int ncols, bool rowok, bool vec, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (!rowok) return;
    if (vec && col + 7 < ncols) {
        const float4 a = *reinterpret_cast<const float4*>(rowp + col), b = *reinterpret_cast<const float4*>(rowp + col + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (col + j < ncols) v[j] = rowp[col + j];
}

__global__ __launch_bounds__(256, X3_WPS) void gemm_nt_big_x3_kernel(
    const float* __restrict__ G, int ldg, int64_t sGo, int64_t sGi, const float* __restrict__ X, int ldx, int64_t sXo,
    int64_t sXi, int M, int Kx, int Nc, int n_inner, int n_batch, int nsplit, float* __restrict__ part, int tilesM, int tilesK,
    int vecG, int vecX) {
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
    const unsigned bid = hpcs_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid % (tilesM * tilesK);
    const int split = bid / (tilesM * tilesK);
    const int m0 = (tile % tilesM) * 128, x0 = (tile / tilesM) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int spb = (Nc + X3_BK - 1) / X3_BK;
    const int64_t T = (int64_t)n_batch * spb;
    int64_t g = (int64_t)split * T / nsplit;
    const int64_t g_end = (int64_t)(split + 1) * T / nsplit;
    const int lrow = tid >> 1, lh = tid & 1;               // loader: thread = (row, 8-column half of the step)
    float rg[8], rx[8];
    auto gload = [&](int64_t gs) {
        const int batch = (int)(gs / spb);
        const int c0 = (int)(gs % spb) * X3_BK + 8 * lh;
        const int bo = batch / n_inner, bi = batch % n_inner;
        ld8_guard(G + bo * sGo + bi * sGi + (int64_t)(m0 + lrow) * ldg, c0, Nc, m0 + lrow < M, vecG, rg);
        ld8_guard(X + bo * sXo + bi * sXi + (int64_t)(x0 + lrow) * ldx, c0, Nc, x0 + lrow < Kx, vecX, rx);
    };
    auto lstore = [&]() {
        u32x4 p1, p2, p3;
        unsigned char* base = x3_lds + lrow * X3_PITCH + lh * 16;
        split8(rg, p1, p2, p3);
        *reinterpret_cast<u32x4*>(base) = p1;
        *reinterpret_cast<u32x4*>(base + X3_PIECE) = p2;
        *reinterpret_cast<u32x4*>(base + 2 * X3_PIECE) = p3;
        split8(rx, p1, p2, p3);
        *reinterpret_cast<u32x4*>(base + X3_OPER) = p1;
        *reinterpret_cast<u32x4*>(base + X3_OPER + X3_PIECE) = p2;
        *reinterpret_cast<u32x4*>(base + X3_OPER + 2 * X3_PIECE) = p3;
    };
    bool live[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) live[i][j] = (m0 + wm * 64 + i * 32 < M) && (x0 + wn * 64 + j * 32 < Kx);
    const unsigned char* sa = x3_lds + (wm * 64 + lr) * X3_PITCH + lk * 16;
    const unsigned char* sb = x3_lds + X3_OPER + (wn * 64 + lr) * X3_PITCH + lk * 16;
    if (g < g_end) {
        gload(g);
        lstore();
    }
    __syncthreads();
    for (; g < g_end; ++g) {
        const bool have_next = g + 1 < g_end;
        if (have_next) gload(g + 1);
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(sa + p * X3_PIECE + i * 32 * X3_PITCH);
                b[i][p] = *reinterpret_cast<const bf16x8*>(sb + p * X3_PIECE + i * 32 * X3_PITCH);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (live[i][j]) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        __syncthreads();                                       // everybody has read its fragments
        if (have_next) lstore();
        __syncthreads();
    }
    float* P = part + (int64_t)split * M * Kx;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = x0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < M && col < Kx) P[(int64_t)row * Kx + col] = acc[i][j][r];
            }
        }
}

}  // namespace

// launcher for csrc/gemm.hip's weight-gradient dispatch (plain partials; the caller runs its split_reduce launches)
int hpcs_x3_nt_big_launch(const float* G, int ldg, int64_t sGo, int64_t sGi, const float* X, int ldx, int64_t sXo, int64_t sXi, int M,
                          int Kx, int Nc, int n_inner, int n_batch, int nsplit, float* part, int tilesM, int tilesK, int vecG, int vecX,
                          long blocks, hipStream_t st) {
    static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_nt_big_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    X3_STAGE) == hipSuccess;
    if (!attr_ok) return HPCS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gemm_nt_big_x3_kernel, dim3((unsigned)blocks), dim3(256), X3_STAGE, st, G, ldg, sGo, sGi, X, ldx, sXo, sXi, M, Kx,
                       Nc, n_inner, n_batch, nsplit, part, tilesM, tilesK, vecG, vecX);
    return HPCS_OK;
}

extern "C" int hpcs_gemm_tn_x3(const float* At, int lda, const float* Bm, int ldb, int64_t strideB_o, int64_t strideB_i, float* C,
                               int ldc, int64_t strideC_o, int64_t strideC_i, int M, int Nc, int K, int n_outer, int n_inner,
                               const float* rowbias, int accumulate, hpcs_stream_t stream) {
    if (!At || !Bm || !C) return HPCS_ERR_NULL;
    if (M <= 0 || Nc <= 0 || K <= 0 || n_outer < 0 || n_inner <= 0 || lda < M || ldb < Nc || ldc < Nc) return HPCS_ERR_SHAPE;
    if (n_outer == 0) return HPCS_OK;
    static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_tn_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    X3_NBUF * X3_STAGE) == hipSuccess;
    if (!attr_ok) return HPCS_ERR_UNSUPPORTED;
    const int tm = hpcs_cdiv(M, X3_BM), tn = hpcs_cdiv(Nc, X3_BN);
    const long total = (long)tm * tn * n_outer * n_inner;
    if (total > 0x7fffffffL) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3((unsigned)total), dim3(256), X3_NBUF * X3_STAGE, (hipStream_t)stream, At, lda, Bm, ldb,
                       strideB_o, strideB_i, C, ldc, strideC_o, strideC_i, M, Nc, K, n_inner, rowbias, accumulate, tm, tn,
                       (unsigned)total);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
