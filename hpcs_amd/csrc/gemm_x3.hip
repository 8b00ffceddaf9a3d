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
// 4-byte buffer loads, one per k, the row offset in a scalar register -- splits it in registers and writes one 16-byte
// [row][k8] piece per split level to LDS (row pitch 48 bytes: conflict-free for the 16-byte fragment reads), so the
// transposition costs nothing and no split copy of any tensor ever exists in HBM.  Register-staged prefetch two K steps ahead.
//
// The loop is priced in vector-issue slots, not in MFMA cycles: an MFMA holds the SIMD's issue port for 8 of its 32 cycles and
// every VALU instruction for 4, and on random data the chip lowers its clock as the loop gets denser (MI355X_MICROARCH.md
// "DVFS give-back").  The first form of this loader spent 185 VALU instructions per wave and K step (64-bit address
// arithmetic per load, min-clamps, tail selects) beside 24 MFMAs; buffer loads with scalar row offsets and the descriptor's
// range check (rows k >= K read as zeros) leave the 88 of the split itself: 141-157 -> 159-175 TFLOP/s on the VN shapes.
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

// The six piece products of one 32 x 32 tile, smallest terms first
#define X3_MMA6(c, a, b)                                                        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);        \
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);

// Raw buffer descriptor over one operand matrix: loads take a 32-bit per-lane byte offset plus a scalar one, and anything at
// or beyond `bytes` reads as zero -- the K tail needs no clamp and no select, and no 64-bit address arithmetic is left.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    const unsigned long long q = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)) << 32);
    return __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// Software pipeline of one workgroup (single LDS stage, two register sets for the raw operands):
//   step kt:  request the raw operands of step kt+2 | read the fragments of step kt from LDS | the 24 MFMAs of step kt
//             INTERLEAVED with the VALU split of step kt+1's raw operands (the matrix pipe runs under the ~100 split
//             instructions instead of after them: sched_group_barrier) | barrier | write the split pieces | barrier.
template <bool FULL>
__device__ __forceinline__ void x3_tn_tile(__amdgpu_buffer_rsrc_t rsA, unsigned voA, int lda, __amdgpu_buffer_rsrc_t rsB, unsigned voB,
                                           int ldb, int K, int lh,
                                           unsigned char* wbase, const unsigned char* sa, const unsigned char* sb,
                                           const bool (&live)[2][2], f32x16 (&acc)[2][2]) {
    // lh (the k half this wave's loader threads own) is wave-uniform: the row offsets below are scalar
    float ra[8], rb[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 8 * lh + j;                                 // rows k >= K are beyond the descriptors: zeros
            ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, voA, k * lda * 4, 0));
            rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, voB, k * ldb * 4, 0));
        }
    };
    auto lwrite = [&](const u32x4 (&pa)[3], const u32x4 (&pb)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<u32x4*>(wbase + p * X3_PIECE) = pa[p];
            *reinterpret_cast<u32x4*>(wbase + X3_OPER + p * X3_PIECE) = pb[p];
        }
    };
    const int nk = (K + X3_BK - 1) / X3_BK;
    u32x4 pa[3], pb[3];
    gload(0);
    split8(ra, pa[0], pa[1], pa[2]);
    split8(rb, pb[0], pb[1], pb[2]);
    if (nk > 1) gload(X3_BK);
    lwrite(pa, pb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        // (ra, rb) hold the raw operands of step kt+1 (requested one step ago); after their split the same registers take
        // the request for step kt+2, which has the barriers, the piece stores and the first MFMAs of the next step to land
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(sa + p * X3_PIECE + i * 32 * X3_PITCH);
                b[i][p] = *reinterpret_cast<const bf16x8*>(sb + p * X3_PIECE + i * 32 * X3_PITCH);
            }
        const bool have1 = kt + 1 < nk;
        if (FULL) {
            X3_MMA6(acc[0][0], a[0], b[0]) X3_MMA6(acc[0][1], a[0], b[1])
            split8(ra, pa[0], pa[1], pa[2]);
            X3_MMA6(acc[1][0], a[1], b[0]) X3_MMA6(acc[1][1], a[1], b[1])
            split8(rb, pb[0], pb[1], pb[2]);
        } else {
            if (live[0][0]) { X3_MMA6(acc[0][0], a[0], b[0]) }
            if (live[0][1]) { X3_MMA6(acc[0][1], a[0], b[1]) }
            if (live[1][0]) { X3_MMA6(acc[1][0], a[1], b[0]) }
            if (live[1][1]) { X3_MMA6(acc[1][1], a[1], b[1]) }
            split8(ra, pa[0], pa[1], pa[2]);
            split8(rb, pb[0], pb[1], pb[2]);
        }
        if (kt + 2 < nk) gload((kt + 2) * X3_BK);
        __syncthreads();                                       // everybody has read its fragments of step kt
        if (have1) lwrite(pa, pb);
        __syncthreads();
    }
}

// Epilogue of the tn kernels.  D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Through a wave-private LDS
// patch (the operand stages are dead after the loop's last barrier) every lane owns 4 CONSECUTIVE columns and the tile leaves
// as 16-byte stores, eight 128-byte row segments per instruction (vn1's forward writes 268 MB), as in csrc/gemm.hip.
__device__ __forceinline__ void x3_tn_store(unsigned char* x3_lds, const f32x16 (&acc)[2][2], float* __restrict__ C, float* __restrict__ Cp,
                                            int ldc, int64_t sCo, int64_t sCi, int M, int Nc, int m0, int n0, int wave, int lane,
                                            const float* __restrict__ bias, int accumulate) {
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = lane >> 5;
    const bool vecC = (ldc % 4 == 0) && (sCo % 4 == 0) && (sCi % 4 == 0) && ((((uintptr_t)C) & 15u) == 0) && (Nc % 4 == 0);
    constexpr int PP = 36;
    float* patch = reinterpret_cast<float*>(x3_lds) + wave * (32 * PP);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row0 = m0 + wm * 64 + i * 32, col0 = n0 + wn * 64 + j * 32;
            if (row0 >= M || col0 >= Nc) continue;               // wave-uniform
            if (vecC) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * PP + lr] = acc[i][j][r];
                __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0); the patch is wave-private: no barrier
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rr = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                    const int row = row0 + rr, col = col0 + c4;
                    if (row < M && col < Nc) {
                        float4 v = *reinterpret_cast<const float4*>(patch + rr * PP + c4);
                        if (bias) { const float bb = bias[row]; v.x += bb; v.y += bb; v.z += bb; v.w += bb; }
                        float* dst = Cp + (int64_t)row * ldc + col;
                        if (accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(dst);
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(dst) = v;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);              // the reads are done before the next tile overwrites
            } else {
                const int col = col0 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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
}

__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(
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
    // ---- loader: thread = (row, k half).  Columns beyond the matrix read finite neighbours (their outputs are never stored)
    // or, past the end of the operand, zeros.
    const int lrow = tid & 127;
    const int lh = __builtin_amdgcn_readfirstlane(tid >> 7);
    const __amdgpu_buffer_rsrc_t rsA = x3_rsrc(At, (unsigned)K * (unsigned)lda * 4u);
    const __amdgpu_buffer_rsrc_t rsB = x3_rsrc(Bp, ((unsigned)(K - 1) * (unsigned)ldb + (unsigned)Nc) * 4u);
    const unsigned voA = (unsigned)(m0 + lrow) * 4u, voB = (unsigned)(n0 + lrow) * 4u;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bool live[2][2];
    bool all_live = true;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            live[i][j] = (m0 + wm * 64 + i * 32 < M) && (n0 + wn * 64 + j * 32 < Nc);
            all_live = all_live && live[i][j];
        }
    const int lr = lane & 31, lk = lane >> 5;
    unsigned char* wbase = x3_lds + lrow * X3_PITCH + (tid >> 7) * 16;
    const unsigned char* sa = x3_lds + (wm * 64 + lr) * X3_PITCH + lk * 16;
    const unsigned char* sb = x3_lds + X3_OPER + (wn * 64 + lr) * X3_PITCH + lk * 16;
    // (the branch is per wave; both bodies hold the same barriers)
    if (all_live) x3_tn_tile<true>(rsA, voA, lda, rsB, voB, ldb, K, lh, wbase, sa, sb, live, acc);
    else x3_tn_tile<false>(rsA, voA, lda, rsB, voB, ldb, K, lh, wbase, sa, sb, live, acc);
    x3_tn_store(x3_lds, acc, C, Cp, ldc, sCo, sCi, M, Nc, m0, n0, wave, lane, bias, accumulate);
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

// FAST: 16-byte loadable operands and Nc % 16 == 0 (every shape of the step): straight-line loads from clamped rows (a row
// beyond the matrix reads the last row and is zeroed by a select), so that the split interleaves with the MFMAs.
template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_nt_big_x3_kernel(
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
    const bool okG = m0 + lrow < M, okX = x0 + lrow < Kx;
    const int64_t offG = (int64_t)min(m0 + lrow, M - 1) * ldg + 8 * lh, offX = (int64_t)min(x0 + lrow, Kx - 1) * ldx + 8 * lh;
    float rg[8], rx[8];
    // The loader walks the step list in order: (batch, column) advance incrementally, the batch's base pointers are
    // recomputed only when the batch changes (the division per step cost ~200 scalar instructions).
    int ld_batch = (int)(g / spb), ld_c0 = (int)(g % spb) * X3_BK;
    const float* ld_gp = G + (ld_batch / n_inner) * sGo + (ld_batch % n_inner) * sGi;
    const float* ld_xp = X + (ld_batch / n_inner) * sXo + (ld_batch % n_inner) * sXi;
    auto gload = [&]() {
        if (FAST) {
            // rows beyond the matrix are clamped to the last row: their products land in rows / columns that are never stored
            const float* gp = ld_gp + offG + ld_c0;
            const float* xp = ld_xp + offX + ld_c0;
            const float4 a0 = *reinterpret_cast<const float4*>(gp), a1 = *reinterpret_cast<const float4*>(gp + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(xp), b1 = *reinterpret_cast<const float4*>(xp + 4);
            rg[0] = a0.x; rg[1] = a0.y; rg[2] = a0.z; rg[3] = a0.w; rg[4] = a1.x; rg[5] = a1.y; rg[6] = a1.z; rg[7] = a1.w;
            rx[0] = b0.x; rx[1] = b0.y; rx[2] = b0.z; rx[3] = b0.w; rx[4] = b1.x; rx[5] = b1.y; rx[6] = b1.z; rx[7] = b1.w;
        } else {
            ld8_guard(ld_gp + (int64_t)(m0 + lrow) * ldg, ld_c0 + 8 * lh, Nc, okG, vecG, rg);
            ld8_guard(ld_xp + (int64_t)(x0 + lrow) * ldx, ld_c0 + 8 * lh, Nc, okX, vecX, rx);
        }
        ld_c0 += X3_BK;
        if (ld_c0 >= Nc) {
            ld_c0 = 0;
            ++ld_batch;
            ld_gp = G + (ld_batch / n_inner) * sGo + (ld_batch % n_inner) * sGi;
            ld_xp = X + (ld_batch / n_inner) * sXo + (ld_batch % n_inner) * sXi;
        }
    };
    unsigned char* wbase = x3_lds + lrow * X3_PITCH + lh * 16;
    auto lwrite = [&](const u32x4 (&pa)[3], const u32x4 (&pb)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<u32x4*>(wbase + p * X3_PIECE) = pa[p];
            *reinterpret_cast<u32x4*>(wbase + X3_OPER + p * X3_PIECE) = pb[p];
        }
    };
    bool live[2][2];
    bool all_live = true;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            live[i][j] = (m0 + wm * 64 + i * 32 < M) && (x0 + wn * 64 + j * 32 < Kx);
            all_live = all_live && live[i][j];
        }
    const unsigned char* sa = x3_lds + (wm * 64 + lr) * X3_PITCH + lk * 16;
    const unsigned char* sb = x3_lds + X3_OPER + (wn * 64 + lr) * X3_PITCH + lk * 16;
    u32x4 pa[3], pb[3];
    if (g < g_end) {
        gload();
        split8(rg, pa[0], pa[1], pa[2]);
        split8(rx, pb[0], pb[1], pb[2]);
        if (g + 1 < g_end) gload();
        lwrite(pa, pb);
    }
    __syncthreads();
    for (; g < g_end; ++g) {
        const bool have1 = g + 1 < g_end;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(sa + p * X3_PIECE + i * 32 * X3_PITCH);
                b[i][p] = *reinterpret_cast<const bf16x8*>(sb + p * X3_PIECE + i * 32 * X3_PITCH);
            }
        if (all_live) {
            X3_MMA6(acc[0][0], a[0], b[0]) X3_MMA6(acc[0][1], a[0], b[1])
                split8(rg, pa[0], pa[1], pa[2]);
            X3_MMA6(acc[1][0], a[1], b[0]) X3_MMA6(acc[1][1], a[1], b[1])
            split8(rx, pb[0], pb[1], pb[2]);
        } else {
            if (live[0][0]) { X3_MMA6(acc[0][0], a[0], b[0]) }
            if (live[0][1]) { X3_MMA6(acc[0][1], a[0], b[1]) }
            if (live[1][0]) { X3_MMA6(acc[1][0], a[1], b[0]) }
            if (live[1][1]) { X3_MMA6(acc[1][1], a[1], b[1]) }
                split8(rg, pa[0], pa[1], pa[2]);
            split8(rx, pb[0], pb[1], pb[2]);
        }
        if (g + 2 < g_end) gload();
        __syncthreads();                                       // everybody has read its fragments
        if (have1) lwrite(pa, pb);
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
    // The stage is 36 KB = three workgroups per CU = 456 of a SIMD's 512 VGPRs: beside them a streaming kernel of the main
    // stream (this kernel runs on the side stream under the VN backward's element-wise passes) finds room for ONE wave per
    // SIMD.  Requesting 56 KB holds the residency at two workgroups per CU: the kernel alone is slower, the step 0.07-0.08 ms
    // faster in three alternating runs (8.25 -> 8.17 ms; 90 KB = one per CU: the same).  HPCS_X3_NT_LDS_KB=36 restores three.
    const int want_kb = hpcs_tuning().x3_nt_lds_kb;
    const int lds = want_kb * 1024 > X3_STAGE ? (want_kb > 156 ? 156 * 1024 : want_kb * 1024) : X3_STAGE;
    static const bool attr_ok =
        hipFuncSetAttribute((const void*)gemm_nt_big_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess &&
        hipFuncSetAttribute((const void*)gemm_nt_big_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) == hipSuccess;
    if (!attr_ok) return HPCS_ERR_UNSUPPORTED;
    if (vecG && vecX && Nc % X3_BK == 0)
        hipLaunchKernelGGL(gemm_nt_big_x3_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, st, G, ldg, sGo, sGi, X, ldx, sXo, sXi,
                           M, Kx, Nc, n_inner, n_batch, nsplit, part, tilesM, tilesK, vecG, vecX);
    else
        hipLaunchKernelGGL(gemm_nt_big_x3_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, st, G, ldg, sGo, sGi, X, ldx, sXo, sXi,
                           M, Kx, Nc, n_inner, n_batch, nsplit, part, tilesM, tilesK, vecG, vecX);
    return HPCS_OK;
}

extern "C" int hpcs_gemm_tn_x3(const float* At, int lda, const float* Bm, int ldb, int64_t strideB_o, int64_t strideB_i, float* C,
                               int ldc, int64_t strideC_o, int64_t strideC_i, int M, int Nc, int K, int n_outer, int n_inner,
                               const float* rowbias, int accumulate, hpcs_stream_t stream) {
    if (!At || !Bm || !C) return HPCS_ERR_NULL;
    if (M <= 0 || Nc <= 0 || K <= 0 || n_outer < 0 || n_inner <= 0 || lda < M || ldb < Nc || ldc < Nc) return HPCS_ERR_SHAPE;
    if (n_outer == 0) return HPCS_OK;
    static const bool attr_ok = hipFuncSetAttribute((const void*)gemm_tn_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    X3_STAGE) == hipSuccess;
    if (!attr_ok) return HPCS_ERR_UNSUPPORTED;
    // the loader addresses one operand matrix with 32-bit byte offsets (csrc/gemm.hip routes larger operands to the fma chain)
    if ((int64_t)(K + X3_BK) * lda * 4 >= ((int64_t)1 << 31) || (int64_t)(K + X3_BK) * ldb * 4 >= ((int64_t)1 << 31))
        return HPCS_ERR_UNSUPPORTED;
    const int tm = hpcs_cdiv(M, X3_BM), tn = hpcs_cdiv(Nc, X3_BN);
    const long total = (long)tm * tn * n_outer * n_inner;
    if (total > 0x7fffffffL) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(gemm_tn_x3_kernel, dim3((unsigned)total), dim3(256), X3_STAGE, (hipStream_t)stream, At, lda, Bm, ldb,
                       strideB_o, strideB_i, C, ldc, strideC_o, strideC_i, M, Nc, K, n_inner, rowbias, accumulate, tm, tn,
                       (unsigned)total);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
