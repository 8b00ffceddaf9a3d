// Channel-mixing GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// Replaces nn.Linear-over-channels (map_to_feat / map_to_dir, vn_layers.py:14-24,48-77) and the
// Conv1d(k=1) heads (vn_dgcnn_partseg.py:40-57) of the reference, forward and backward:
//   gemm_tn : C[b] = At^T * B[b]          forward (At = W^T) and data-grad (At = W)
//   gemm_nt : dW   = sum_b G[b] * X[b]^T  weight-grad, deterministic split reduction
//
// Tiling (wave64): workgroup tile BM x BN, WM x WN waves, each wave owns (BM/WM) x (BN/WN) as 32x32
// MFMA tiles.  Both operands are staged K-major in LDS ([k][m] / [k][n]) so every MFMA operand fetch is one
// conflict-free ds_read_b32 per lane (lanes 0-31 = 32 consecutive floats of row k, lanes 32-63 = row k+1).
// Global->LDS goes through registers, double-buffered, one barrier per K step.
// Block ids are remapped so that the row tiles sharing one activation panel run on the same XCD (L2 reuse).
//
// The large shapes (gemm_tn with M > 64, weight gradients with both output sides >= 128) are routed to csrc/gemm_x3.hip by
// default: fp32 operands and results, every product formed from three exact bf16 pieces per operand on the bf16 matrix
// cores (hpcs_tuning().gemm_x3; HPCS_GEMM_X3=0 keeps them here).
#include <type_traits>

#include "common.h"

int hpcs_x3_nt_big_launch(const float* G, int ldg, int64_t sGo, int64_t sGi, const float* X, int ldx, int64_t sXo, int64_t sXi, int M,
                          int Kx, int Nc, int n_inner, int n_batch, int nsplit, float* part, int tilesM, int tilesK, int vecG, int vecX,
                          long blocks, hipStream_t st);      // csrc/gemm_x3.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // register image of a 16-byte load (a native vector: always SSA)

// -DHPCS_TN_STAMPS (scratch builds only): every workgroup of gemm_tn_kernel records shader-clock stamps at its phase
// boundaries (100 MHz wall clock: start | first tile staged | main loop done | epilogue done) + its XCC/CU id into g_tn_stamps[5 * logical].
#ifdef HPCS_TN_STAMPS
__device__ long long* g_tn_stamps = nullptr;
#define TN_STAMP(i)                                                                    \
    do {                                                                               \
        if (g_tn_stamps && threadIdx.x == 0) g_tn_stamps[5 * (size_t)blockIdx.x + (i)] = (long long)wall_clock64(); \
    } while (0)
#else
#define TN_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float4 ld4_guard(const float* __restrict__ rowp, int col, int ncols, bool rowok,
                                            bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!rowok) return v;
    if (vec && col + 3 < ncols) return *reinterpret_cast<const float4*>(rowp + col);
    if (col < ncols) v.x = rowp[col];
    if (col + 1 < ncols) v.y = rowp[col + 1];
    if (col + 2 < ncols) v.z = rowp[col + 2];
    if (col + 3 < ncols) v.w = rowp[col + 3];
    return v;
}

// One output tile of C = At^T * B: BM x BN, WM x WN waves of (BM/WM) x (BN/WN) each, BK-deep K steps, double-buffered LDS.
// FAST: both operands are 16-byte loadable everywhere (lda, ldb, Nc multiples of 4, aligned bases): the loader is
// straight-line code -- clamped addresses instead of guards (a clamped column only feeds output rows / columns that are never
// stored; A's rows beyond K are zeroed on their way to LDS, B's are finite copies of row K-1) -- so that the compiler can
// count the loads in flight (`vmcnt(n)`; behind guard branches it falls back to `vmcnt(0)` and a maze of exec-mask
// branches: 509 -> 487 us on vn1's forward, 57 -> 48 us on conv6's).  Measured on top of it and NOT kept: a second K step
// of prefetch distance (two register sets, 124 VGPRs, still four waves per SIMD): 484 vs 487 us.
template <int BM, int BN, int BK, int WM, int WN, bool FAST>
__device__ __forceinline__ void tn_tile(const float* __restrict__ At, int lda, const float* __restrict__ Bp, int ldb,
                                        float* __restrict__ Cp, int ldc, int M, int Nc, int K, const float* __restrict__ bias,
                                        int accumulate, int m0, int n0, int vecA, int vecB, int vecC, float* lds_raw) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_V4 = BK * BM / 4, B_V4 = BK * BN / 4;                  // 16-byte pieces of one operand tile
    constexpr int A_LD = (A_V4 + NT - 1) / NT, B_LD = (B_V4 + NT - 1) / NT;
    static_assert(A_V4 % NT == 0 && B_V4 % NT == 0, "operand tiles must be a whole number of passes");
    float (*As)[BK][BM] = reinterpret_cast<float (*)[BK][BM]>(lds_raw);
    float (*Bs)[BK][BN] = reinterpret_cast<float (*)[BK][BN]>(lds_raw + 2 * BK * BM);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rA[A_LD], rB[B_LD];
    int rowA[A_LD], c4A[A_LD], rowB[B_LD], c4B[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int f = tid + i * NT;
        rowA[i] = f / (BM / 4);
        c4A[i] = (f % (BM / 4)) * 4;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int f = tid + i * NT;
        rowB[i] = f / (BN / 4);
        c4B[i] = (f % (BN / 4)) * 4;
    }
    auto gload = [&](int k0) {
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                rA[i] = *reinterpret_cast<const f32x4*>(At + (int64_t)min(k0 + rowA[i], K - 1) * lda + min(m0 + c4A[i], lda - 4));
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                rB[i] = *reinterpret_cast<const f32x4*>(Bp + (int64_t)min(k0 + rowB[i], K - 1) * ldb + min(n0 + c4B[i], Nc - 4));
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const float4 q = ld4_guard(At + (int64_t)(k0 + rowA[i]) * lda, m0 + c4A[i], M, k0 + rowA[i] < K, vecA);
                rA[i] = f32x4{q.x, q.y, q.z, q.w};
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const float4 q = ld4_guard(Bp + (int64_t)(k0 + rowB[i]) * ldb, n0 + c4B[i], Nc, k0 + rowB[i] < K, vecB);
                rB[i] = f32x4{q.x, q.y, q.z, q.w};
            }
        }
    };
    auto lstore = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            f32x4 v = rA[i];
            if constexpr (FAST) {
                if (k0 + rowA[i] >= K) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4*>(&As[buf][rowA[i]][c4A[i]]) = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][rowB[i]][c4B[i]]) = rB[i];
    };

    const int nk = (K + BK - 1) / BK;
    gload(0);
    lstore(0, 0);
    __syncthreads();
    TN_STAMP(1);
    const int lr = lane & 31, lk = lane >> 5;
    // 32x32 sub-tiles that lie entirely in the padding of an edge tile issue no MFMA: the matrix pipe of that SIMD goes to
    // the other resident workgroups instead
    bool live[TM][TN];
    bool all_live = true;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            live[i][j] = (m0 + wm * (BM / WM) + i * 32 < M) && (n0 + wn * (BN / WN) + j * 32 < Nc);
            all_live = all_live && live[i][j];
        }
    // (Measured and NOT kept, r03: requesting the operands of k-pair kk + 1 before the MFMAs of pair kk issue -- two register
    // sets and scheduling barriers; the compiler's own order is read, wait, four MFMAs per pair -- vn1 forward 526 us against
    // 490-510, step +0.1 ms: with four waves per SIMD the other waves already fill the LDS round trips.)
    auto mma = [&](int buf, auto full) {
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf][kk * 2 + lk][wm * (BM / WM) + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk * 2 + lk][wn * (BN / WN) + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (decltype(full)::value || live[i][j])
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        if (all_live) mma(buf, std::true_type{});
        else mma(buf, std::false_type{});
        if (kt + 1 < nk) lstore(buf ^ 1, (kt + 1) * BK);
        __syncthreads();
    }

    TN_STAMP(2);
    if (vecC) {
        // Epilogue through a wave-private LDS patch: the MFMA result holds 4-row x 1-column slivers per lane, i.e. one
        // dword store per register (two 128-byte row segments per wave instruction).  Transposed through LDS each lane
        // owns 4 CONSECUTIVE columns instead and the tile leaves as 16-byte stores, eight 128-byte row segments per
        // instruction -- the shape that streams at the HBM rate (the EdgeConv mixes write 330 MB per launch).
        constexpr int PP = 36;                                   // patch pitch: 16-byte aligned rows, conflict-free
        static_assert(2 * BK * (BM + BN) >= NT / 64 * 32 * PP, "patch does not fit");
        float* patch = lds_raw + wave * (32 * PP);
        // (the main loop ended with a barrier: nobody reads As / Bs any more)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row0 = m0 + wm * (BM / WM) + i * 32, col0 = n0 + wn * (BN / WN) + j * 32;
                if (row0 >= M || col0 >= Nc) continue;           // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * PP + lr] = acc[i][j][r];
                __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0); the patch is wave-private: no barrier
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rr = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                    const int row = row0 + rr, col = col0 + c4;
                    if (row < M && col < Nc) {
                        float4 v = *reinterpret_cast<const float4*>(patch + rr * PP + c4);
                        if (bias) { const float bb = bias[row]; v.x += bb; v.y += bb; v.z += bb; v.w += bb; }
                        float* dst = Cp + (int64_t)row * ldc + col;
                        if (col + 3 < Nc) {
                            if (accumulate) {
                                const float4 o = *reinterpret_cast<const float4*>(dst);
                                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                            }
                            *reinterpret_cast<float4*>(dst) = v;
                        } else {
                            const float vv[4] = {v.x, v.y, v.z, v.w};
                            for (int q = 0; q < 4 && col + q < Nc; ++q) dst[q] = accumulate ? dst[q] + vv[q] : vv[q];
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);              // the reads are done before the next tile overwrites
            }
        TN_STAMP(3);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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

// Measured and NOT kept: a last row tile of 64 / 96 rows with its four waves side by side (M = 341 / 682 leave 43 / 86 dead
// rows in a 128-row last tile whose lower waves then issue no MFMA): 491 vs 488 us, 416 vs 421 us -- the other resident
// workgroups already use the matrix-pipe time the dead waves leave.
template <int BM, int BN, int BK, int WM, int WN, bool FAST>
__global__ __launch_bounds__(WM* WN * 64, 4) void gemm_tn_kernel(
    const float* __restrict__ At, int lda, const float* __restrict__ Bm, int ldb, int64_t sBo, int64_t sBi,
    float* __restrict__ C, int ldc, int64_t sCo, int64_t sCi, int M, int Nc, int K, int n_inner,
    const float* __restrict__ rowbias, int accumulate, int tilesM, int tilesN, unsigned total, int vecA,
    int vecB, int vecC) {
    // one LDS block: the two double-buffered operand tiles, reused as the epilogue's transpose patches
    __shared__ __attribute__((aligned(16))) float lds_raw[2 * BK * (BM + BN)];
    TN_STAMP(0);
    const unsigned logical = hpcs_xcd_remap(blockIdx.x, total);
    const int rowTile = logical % tilesM;
    const unsigned t1 = logical / tilesM;
    const int colTile = t1 % tilesN;
    const int batch = t1 / tilesN;
    const int bo = batch / n_inner, bi = batch % n_inner;
    const float* Bp = Bm + bo * sBo + bi * sBi;
    float* Cp = C + bo * sCo + bi * sCi;
    const float* bias = rowbias ? rowbias + (int64_t)batch * M : nullptr;
    const int m0 = rowTile * BM, n0 = colTile * BN;
    tn_tile<BM, BN, BK, WM, WN, FAST>(At, lda, Bp, ldb, Cp, ldc, M, Nc, K, bias, accumulate, m0, n0, vecA, vecB, vecC, lds_raw);
}

// The EdgeConv channel mix of a graph layer with its edge tensor formed in the operand loader: PD[b] (M x 3*N*k) = At^T E[b],
// E = get_graph_feature(feat, idx) (vn_dgcnn_util.py:10-37) -- row r < C is the difference channel x[r, a, idx[n, j]] - x[r, a, n],
// row r >= C the centre channel x[r - C, a, n], column (a, n, j) = (a * N + n) * k + j -- and E never exists in memory.
// One workgroup = one 64 x 128 output tile with the WHOLE reduction (K = 2C <= 64) staged at once.  The gathers read a
// POINT-major copy of the features (featT [b][a][n][Cp], Cp = pad4(C): feat_transpose_kernel, 8 MB at B = 32): a thread owns
// one column (one edge) and fetches FOUR channels of its neighbour / its centre per 16-byte load -- 12 loads per column.  (The
// channel-major layout costs one 4-byte gather per element: 4032 scattered loads per tile, and the texture path resolves one
// cache line per cycle -- that form measured 192-195 us whether the loads were issued per 16-deep step or all at once, against
// 101 + 159 us for gather + GEMM.)  MFMA operand mapping and k order are those of tn_tile<64,128,..>, so the product is
// bit-identical to hpcs_edge_gather_fwd + hpcs_gemm_tn.
constexpr int GM_BM = 64, GM_BN = 128, GM_KMAX = 64;
__global__ void feat_transpose_kernel(const float* __restrict__ feat, int C, int Cp, int N, float* __restrict__ featT) {
    // feat [b][c][a][n] -> featT [b][a][n][Cp]; one thread per (a, n, 4 channels)
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int g4 = Cp / 4;
    if (t >= 3 * N * g4) return;
    const int g = t % g4, n = (t / g4) % N, a = t / (g4 * N);
    const float* src = feat + (int64_t)b * C * 3 * N + (int64_t)a * N + n;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (4 * g + c < C) ? src[(int64_t)(4 * g + c) * 3 * N] : 0.f;
    *reinterpret_cast<float4*>(featT + ((int64_t)b * 3 * N + (int64_t)a * N + n) * Cp + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256, 4) void gemm_tn_graph_kernel(
    const float* __restrict__ At, int lda, const float* __restrict__ featT, const int64_t* __restrict__ idx, int C, int N, int k,
    float* __restrict__ PD, int M, int tilesN, unsigned total) {
    extern __shared__ __attribute__((aligned(16))) float gm_lds[];
    const int K = 2 * C, Kp = (K + 1) & ~1, Cp = (C + 3) & ~3;
    float (*As)[GM_BM] = reinterpret_cast<float (*)[GM_BM]>(gm_lds);                    // [Kp][64]
    float (*Bs)[GM_BN] = reinterpret_cast<float (*)[GM_BN]>(gm_lds + Kp * GM_BM);       // [Kp][128]
    const unsigned logical = hpcs_xcd_remap(blockIdx.x, total);
    const int colTile = logical % tilesN;
    const int b = logical / tilesN;
    const int Nc = 3 * N * k, n0 = colTile * GM_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- B: thread = (column, half of the channel groups); a column is one edge (a, n, j)
    const int cl = tid & 127, half = tid >> 7;
    const int col = min(n0 + cl, Nc - 1);                     // (a clamped column is never stored)
    const int e = col % (N * k);
    const int a = col / (N * k), nctr = e / k;
    const int nbr = (int)idx[(int64_t)b * N * k + e];
    const float* base = featT + ((int64_t)b * 3 * N + (int64_t)a * N) * Cp;
    constexpr int GMAX = GM_KMAX / 2 / 4 / 2;                 // channel groups per thread: <= 4 (C <= 32)
    float4 vn[GMAX], vc[GMAX];
#pragma unroll
    for (int i = 0; i < GMAX; ++i) {
        const int g = half + 2 * i;
        vn[i] = vc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * g < C) {
            vn[i] = *reinterpret_cast<const float4*>(base + (int64_t)nbr * Cp + 4 * g);
            vc[i] = *reinterpret_cast<const float4*>(base + (int64_t)nctr * Cp + 4 * g);
        }
    }
    // ---- A: At [K][lda] -> As[k][m], zero beyond K / M (lda % 4 == 0, 16-byte aligned)
    for (int f = tid; f < Kp * (GM_BM / 4); f += 256) {
        const int kr = f / (GM_BM / 4), m4 = (f % (GM_BM / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kr < K && m4 < lda) v = *reinterpret_cast<const float4*>(At + (int64_t)kr * lda + m4);
        *reinterpret_cast<float4*>(&As[kr][m4]) = v;
    }
#pragma unroll
    for (int i = 0; i < GMAX; ++i) {
        const int g = half + 2 * i;
        const float dn[4] = {vn[i].x - vc[i].x, vn[i].y - vc[i].y, vn[i].z - vc[i].z, vn[i].w - vc[i].w};
        const float ct[4] = {vc[i].x, vc[i].y, vc[i].z, vc[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * g + c;
            if (r < C) {
                Bs[r][cl] = dn[c];
                Bs[C + r][cl] = ct[c];
            }
        }
    }
    if (Kp > K && half == 0) Bs[K][cl] = 0.f;                  // odd K: the padding row of the last k pair
    __syncthreads();

    // ---- 64 x 32 per wave: two 32 x 32 accumulators, the k order of tn_tile
    const int lr = lane & 31, lk = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int kk = 0; kk < Kp / 2; ++kk) {
        const float bq = Bs[kk * 2 + lk][wave * 32 + lr];
        const float a0 = As[kk * 2 + lk][lr], a1 = As[kk * 2 + lk][32 + lr];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq, acc[1], 0, 0, 0);
    }
    __syncthreads();                                          // operand tiles dead: their LDS becomes the transpose patches

    // ---- epilogue: wave-private 32 x 36 patch, 16-byte row-contiguous stores (the shape that streams at the HBM rate)
    constexpr int PP = 36;
    float* patch = gm_lds + wave * (32 * PP);
    float* Cb = PD + (int64_t)b * M * Nc;
    const int col0 = n0 + wave * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = i * 32;
        if (row0 >= M || col0 >= Nc) continue;                // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * PP + lr] = acc[i][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0); the patch is wave-private
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            const int row = row0 + rr, cc = col0 + c4;
            if (row < M && cc + 3 < Nc)
                *reinterpret_cast<float4*>(Cb + (int64_t)row * Nc + cc) = *reinterpret_cast<const float4*>(patch + rr * PP + c4);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

// Measured and NOT kept: the VN batch-norm statistics in this GEMM's epilogue for the EdgeConv layers (one workgroup computing
// the same 64 x 128 tile of the x, y and z planes, |p| from the three accumulator sets): bit-identical product, 220 vs 254 us
// against GEMM + statistics pass on cold data, but 185 vs 71 + 40 us inside the step (the statistics pass reads the
// pre-activations from the infinity cache; three accumulator sets leave two workgroups per CU).  Splitting the tile body into
// main-loop and epilogue functions for it also cost the 64-row kernel its 72-VGPR allocation (128, i.e. 4 instead of 6
// workgroups per CU: 74 -> 98 us per launch in the step), so the body stays one function.

// Measured and NOT kept: a persistent form (min(tiles, 3-4 per CU) workgroups walk the tile list in the same XCD-aware order,
// the next tile's first K step requested before the current tile's epilogue): vn1 forward 505-516 vs 496 us, EdgeConv mix
// (K = 42) 150-156 vs 147 us -- four one-tile workgroups per CU already overlap each other's prologues and epilogues.

// ---------------------------------------------------------------------------------------------------------
// Small problems (a handful of MFMA tiles in total: row biases, [3B x C] frames): the tiled kernel would run on two
// or three workgroups and is pure latency.  A workgroup owns 8 output rows x 32 columns; eight lanes per column split
// the reduction (k = s, s+8, ...) and every B element fetched from L2 feeds the 8 rows; the partial sums meet in a
// 3-step butterfly.
constexpr int SMALL_RB = 8;
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(
    const float* __restrict__ At, int lda, const float* __restrict__ Bm, int ldb, int64_t sBo, int64_t sBi,
    float* __restrict__ C, int ldc, int64_t sCo, int64_t sCi, int M, int Nc, int K, int n_inner,
    const float* __restrict__ rowbias, int accumulate, int colBlocks, int rowBlocks) {
    const int cb = blockIdx.x % colBlocks;
    const int t1 = blockIdx.x / colBlocks;
    const int row0 = (t1 % rowBlocks) * SMALL_RB, batch = t1 / rowBlocks;
    const int s = threadIdx.x & 7;
    const int col = cb * 32 + (threadIdx.x >> 3);
    const bool live = col < Nc;
    const int bo = batch / n_inner, bi = batch % n_inner;
    const float* Bp = Bm + bo * sBo + bi * sBi + (live ? col : 0);
    float acc[SMALL_RB];
#pragma unroll
    for (int r = 0; r < SMALL_RB; ++r) acc[r] = 0.f;
    int k = s;
    for (; k + 8 < K; k += 16) {
        const float b0 = Bp[(int64_t)k * ldb], b1 = Bp[(int64_t)(k + 8) * ldb];
        const float* A0 = At + (int64_t)k * lda + row0;
        const float* A1 = A0 + (int64_t)8 * lda;
        float a0[SMALL_RB], a1[SMALL_RB];
#pragma unroll
        for (int r = 0; r < SMALL_RB; ++r) {
            const bool in = row0 + r < M;
            a0[r] = in ? A0[r] : 0.f;
            a1[r] = in ? A1[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < SMALL_RB; ++r) acc[r] = fmaf(a1[r], b1, fmaf(a0[r], b0, acc[r]));
    }
    for (; k < K; k += 8) {
        const float b0 = Bp[(int64_t)k * ldb];
        const float* A0 = At + (int64_t)k * lda + row0;
#pragma unroll
        for (int r = 0; r < SMALL_RB; ++r) acc[r] = fmaf(row0 + r < M ? A0[r] : 0.f, b0, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < SMALL_RB; ++r) {
        float v = acc[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        acc[r] = v;
    }
    if (!live) return;
    // lane s finishes row row0 + s
    float mine = acc[0];
#pragma unroll
    for (int r = 1; r < SMALL_RB; ++r) mine = s == r ? acc[r] : mine;
    const int row = row0 + s;
    if (row >= M) return;
    if (rowbias) mine += rowbias[(int64_t)batch * M + row];
    float* dst = C + bo * sCo + bi * sCi + (int64_t)row * ldc + col;
    if (accumulate) mine += *dst;
    *dst = mine;
}

// ---------------------------------------------------------------------------------------------------------
// In-launch split reduction of the weight-gradient GEMMs (r03; replaces the split_reduce_kernel launches that followed
// every hpcs_gemm_nt: 26 per training step, 1.0 ms of side-stream time).  Every split workgroup publishes its partial tile
// and takes a ticket; the last arriver of a SLICE of splits sums that slice's partials in split order, and -- two levels,
// when a tile has many splits (the EdgeConv gradients: one 42 x 42 tile over ~1900 splits) -- the last slice-reducer sums
// the slice partials in slice order into dW.  Fixed summation order: bit-reproducible.
// Visibility (per-XCD L2s are not coherent, MI355X_MICROARCH "inter-workgroup visibility"): the partials are stored
// WRITE-THROUGH (agent-scope relaxed atomic stores = `global_store_dword sc1`), every storing wave drains `vmcnt(0)`, the
// workgroup meets at a barrier, ONE lane adds to the ticket (agent-scope atomic), and the reducer reads every partial with
// `sc1` loads.  No release / acquire fence: `buffer_wbl2` would write back the whole XCD L2, i.e. the main stream's
// 330 MB gradient tensors in flight beside these side-stream GEMMs.  Tickets come zeroed from the caller (the step's
// fp64 arena: one fill per step); the last reducer leaves them zero again.
struct NtFuse {
    unsigned* tickets;        // [tiles * (slices + 1)]: per (tile, slice) arrivals, then per tile slice arrivals; nullptr: off
    float* stage;             // [slices][tiles][TE] slice sums (two-level only)
    float* dW;
    int lddw, accumulate, slice_len, slices;
    unsigned part_bytes, stage_bytes;     // extents of the two write-through buffers (buffer descriptors)
};
typedef unsigned int nt_v4u __attribute__((ext_vector_type(4)));
// 16-byte write-through store / L1-bypassing load (aux 16 = sc1) through a raw buffer descriptor built from wave-uniform values
__device__ __forceinline__ void nt_store16_wt(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float4 v) {
    nt_v4u u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, byte_off, 0, 16);
}
__device__ __forceinline__ float4 nt_load16_wt(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const nt_v4u u = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
// Called by EVERY thread of the workgroup once its finished TILE sits in LDS (`tl`: TR rows x TC columns, row-major, dense;
// rows / columns beyond M / Kx hold zeros).  Partials are TILE-MAJOR in the workspace -- part[split][tile][TR*TC] -- so that
// the publish and the reducer's reads are whole 16-byte accesses: the first version (element-major partials, one dword
// write-through store and one dword load per element, eight in flight) made the last arriver a 0.1-0.5 ms serial tail
// (gemm_nt 1.75 -> 3.2 ms per step).  `sh`: one int of LDS outside `tl`.
template <int NT, int TR, int TC>
__device__ __forceinline__ void nt_fused_reduce(const NtFuse& f, float* __restrict__ part, const float* tl, int nsplit, int split,
                                                int tile, int tiles, int M, int Kx, int m0, int x0, int* sh) {
    constexpr int TE = TR * TC, T4 = TE / 4;
    static_assert(TE % 4 == 0 && TC % 4 == 0, "whole float4 rows");
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)f.part_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(f.stage, 0, (int)f.stage_bytes, 0x00020000);
    const unsigned my_base = (unsigned)((split * tiles + tile) * TE) * 4u;
    for (int q = tid; q < T4; q += NT) nt_store16_wt(rp, my_base + 16u * q, *reinterpret_cast<const float4*>(tl + 4 * q));
    const int slice = split / f.slice_len;
    const int s0 = slice * f.slice_len, s1 = min(nsplit, s0 + f.slice_len);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through partial stores have been performed
    __syncthreads();
    if (tid == 0)
        *sh = __hip_atomic_fetch_add(f.tickets + (int64_t)tile * f.slices + slice, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(s1 - s0 - 1);
    __syncthreads();
    if (!*sh) return;
    const bool two = f.slices > 1;
    auto finish = [&](int q, float4 a) {                      // four consecutive columns of one tile row -> dW
        const int row = m0 + (4 * q) / TC, col = x0 + (4 * q) % TC;
        if (row >= M) return;
        float* d = f.dW + (int64_t)row * f.lddw + col;
        const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < Kx) d[e] = f.accumulate ? d[e] + v[e] : v[e];
    };
    // level 1: the slice's partials in split order, 8 independent 16-byte loads in flight per float4 of the tile
    for (int q = tid; q < T4; q += NT) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned o = (unsigned)(tile * TE) * 4u + 16u * q, pitch = (unsigned)(tiles * TE) * 4u;
        int sp = s0;
        for (; sp + 8 <= s1; sp += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = nt_load16_wt(rp, (unsigned)(sp + u) * pitch + o);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; sp < s1; ++sp) {
            const float4 v = nt_load16_wt(rp, (unsigned)sp * pitch + o);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        if (two) nt_store16_wt(rs, (unsigned)((slice * tiles + tile) * TE) * 4u + 16u * q, a);
        else finish(q, a);
    }
    if (tid == 0) __hip_atomic_store(f.tickets + (int64_t)tile * f.slices + slice, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!two) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
        *sh = __hip_atomic_fetch_add(f.tickets + (int64_t)tiles * f.slices + tile, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(f.slices - 1);
    __syncthreads();
    if (!*sh) return;
    for (int q = tid; q < T4; q += NT) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const unsigned o = (unsigned)(tile * TE) * 4u + 16u * q, pitch = (unsigned)(tiles * TE) * 4u;
        int sl = 0;
        for (; sl + 8 <= f.slices; sl += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = nt_load16_wt(rs, (unsigned)(sl + u) * pitch + o);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; sl < f.slices; ++sl) {
            const float4 v = nt_load16_wt(rs, (unsigned)sl * pitch + o);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        finish(q, a);
    }
    if (tid == 0) __hip_atomic_store(f.tickets + (int64_t)tiles * f.slices + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// weight-grad: both operands have the reduction index (columns) contiguous.
template <int BM, int BN, int BKC, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(
    const float* __restrict__ G, int ldg, int64_t sGo, int64_t sGi, const float* __restrict__ X, int ldx,
    int64_t sXo, int64_t sXi, int M, int Kx, int Nc, int n_inner, int n_batch, int chunk, int units_per_batch,
    int nsplit, float* __restrict__ part, int tilesM, int tilesK, int vecG, int vecX, NtFuse fuse) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // Row pitch BKC + 4 floats: rows start on 16 bytes, so staging is one ds_write_b128 per float4 and the MFMA operands
    // are fetched with ds_read_b128 (four consecutive columns = two k-steps per lane); at a 144-byte pitch eight
    // consecutive rows cover the eight 16-byte bank groups, so both are conflict-free.  (A BKC + 1 pitch with scalar
    // LDS traffic made the EdgeConv shapes LDS-issue bound: 238 -> 178 us.)
    // (BM = 128 is no longer instantiated: outputs from 300 x 300 up take gemm_nt_big_kernel.  Its branch here -- BKC + 1
    // pitch, scalar LDS traffic, single-buffered -- measured 906 / 427 us on the 682 x 341 / 340 x 341 gradients.)
    constexpr bool V4 = BM == 64;
    constexpr int LD = V4 ? BKC + 4 : BKC + 1;
    constexpr int G_LD = BM * BKC / 4 / NT, X_LD = BN * BKC / 4 / NT;
    // one LDS array (operand tiles; reused as the dense output tile + flag by the in-launch reduction)
    constexpr int SM_FLOATS = (BM + BN) * LD > BM * BN + 4 || BM > 64 ? (BM + BN) * LD : BM * BN + 4;
    __shared__ __attribute__((aligned(16))) float sm_all[SM_FLOATS];
    float (*Gs)[LD] = reinterpret_cast<float (*)[LD]>(sm_all);
    float (*Xs)[LD] = reinterpret_cast<float (*)[LD]>(sm_all + BM * LD);

    // (Measured and NOT kept, r03: hpcs_xcd_remap here, so that the tiles of one split -- which read the same column slice of
    // G and X -- share one XCD's L2: the 340 x 341 / 682 x 341 gradients went 379 -> 603 us and 649 -> 698 us standalone.)
    const int tile = blockIdx.x % (tilesM * tilesK);
    const int split = blockIdx.x / (tilesM * tilesK);
    const int rowTile = tile % tilesM, colTile = tile / tilesM;
    const int m0 = rowTile * BM, x0 = colTile * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lk = lane >> 5;

    const bool wave_live = (TM * TN > 1) || ((m0 + wm * (BM / WM) < M) && (x0 + wn * (BN / WN) < Kx));
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The reduction range of this workgroup is a list of (batch, column-chunk) units; its BKC-wide steps are walked
    // with a one-step register prefetch: the global loads of step i+1 are in flight while step i is multiplied.
    const int64_t units = (int64_t)n_batch * units_per_batch;
    float4 rg[G_LD], rx[X_LD];
    // iterator over (unit, c0): the unit's operand bases and column range are found (a 64-bit division) when the unit is
    // opened, not in every step
    const float* Gp_u = G;
    const float* Xp_u = X;
    int c0 = 0, c_end_u = 0;
    auto open_unit = [&](int64_t uu) {
        const int batch = (int)(uu / units_per_batch);
        c0 = (int)(uu % units_per_batch) * chunk;
        c_end_u = min(Nc, c0 + chunk);
        const int bo = batch / n_inner, bi = batch % n_inner;
        Gp_u = G + bo * sGo + bi * sGi;
        Xp_u = X + bo * sXo + bi * sXi;
    };
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < G_LD; ++i) {
            const int f = tid + i * NT, row = f / (BKC / 4), c4 = (f % (BKC / 4)) * 4;
            rg[i] = ld4_guard(Gp_u + (int64_t)(m0 + row) * ldg, c0 + c4, c_end_u, m0 + row < M, vecG);
        }
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            const int f = tid + i * NT, row = f / (BKC / 4), c4 = (f % (BKC / 4)) * 4;
            rx[i] = ld4_guard(Xp_u + (int64_t)(x0 + row) * ldx, c0 + c4, c_end_u, x0 + row < Kx, vecX);
        }
    };
    int64_t u = split;
    bool have = u < units;
    if (have) {
        open_unit(u);
        gload();
    }
    while (have) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < G_LD; ++i) {
            const int f = tid + i * NT, row = f / (BKC / 4), c4 = (f % (BKC / 4)) * 4;
            if (V4) *reinterpret_cast<float4*>(&Gs[row][c4]) = rg[i];
            else { Gs[row][c4] = rg[i].x; Gs[row][c4 + 1] = rg[i].y; Gs[row][c4 + 2] = rg[i].z; Gs[row][c4 + 3] = rg[i].w; }
        }
#pragma unroll
        for (int i = 0; i < X_LD; ++i) {
            const int f = tid + i * NT, row = f / (BKC / 4), c4 = (f % (BKC / 4)) * 4;
            if (V4) *reinterpret_cast<float4*>(&Xs[row][c4]) = rx[i];
            else { Xs[row][c4] = rx[i].x; Xs[row][c4 + 1] = rx[i].y; Xs[row][c4 + 2] = rx[i].z; Xs[row][c4 + 3] = rx[i].w; }
        }
        __syncthreads();
        // advance and prefetch
        c0 += BKC;
        if (c0 >= c_end_u) {
            u += nsplit;
            have = u < units;
            if (have) open_unit(u);
        }
        if (have) gload();
        if (!V4) {
#pragma unroll
            for (int kk = 0; kk < BKC / 2; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Gs[wm * (BM / WM) + i * 32 + lr][kk * 2 + lk];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Xs[wn * (BN / WN) + j * 32 + lr][kk * 2 + lk];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if (wave_live)              // a wave whose 32x32 block is all padding (Kx = 341: columns 352..383) idles
#pragma unroll
        for (int k4 = 0; k4 < BKC / 4; ++k4) {               // columns 4*k4 .. 4*k4+3 = k-steps 2*k4 and 2*k4+1
            float4 a4[TM], b4[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[i] = *reinterpret_cast<const float4*>(&Gs[wm * (BM / WM) + i * 32 + lr][k4 * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4*>(&Xs[wn * (BN / WN) + j * 32 + lr][k4 * 4]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(lk ? a4[i].y : a4[i].x, lk ? b4[j].y : b4[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(lk ? a4[i].w : a4[i].z, lk ? b4[j].w : b4[j].z, acc[i][j], 0, 0, 0);
                }
        }
    }
    if constexpr (BM <= 64) {
        if (fuse.tickets != nullptr) {
            // dense BM x BN tile through LDS (the operand tiles are dead), then 16-byte write-through publishes + tickets
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int cl = wn * (BN / WN) + j * 32 + lr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        sm_all[rl * BN + cl] = acc[i][j][r];
                    }
                }
            __syncthreads();
            nt_fused_reduce<NT, BM, BN>(fuse, part, sm_all, nsplit, split, tile, tilesM * tilesK, M, Kx, m0, x0,
                                        reinterpret_cast<int*>(sm_all + BM * BN));
            return;
        }
    }
    float* P = part + (int64_t)split * M * Kx;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = x0 + wn * (BN / WN) + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < M && col < Kx) P[(int64_t)row * Kx + col] = acc[i][j][r];
            }
        }
}

// EdgeConv weight gradients: one output tile with M, Kx <= 48 over millions of columns.  On the 64x64 tile of the
// kernel above the 42x42 result uses 43 % of every 32x32x2 MFMA; here the tile is 3x3 blocks of v_mfma_f32_16x16x4_f32
// (48x48: 77 %), the four waves split each 32-column step (eight columns = two k4-steps each, all nine blocks) and
// add their accumulators through LDS at the end.  Same split plan and partials as the general kernel.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NE_T = 48;
__global__ __launch_bounds__(256) void gemm_nt_edge_kernel(
    const float* __restrict__ G, int ldg, int64_t sGo, int64_t sGi, const float* __restrict__ X, int ldx,
    int64_t sXo, int64_t sXi, int M, int Kx, int Nc, int n_inner, int n_batch, int chunk, int units_per_batch,
    int nsplit, float* __restrict__ part, int vecG, int vecX, NtFuse fuse) {
    constexpr int BKC = 32, NT = 256, LD = BKC + 4;          // pitch 36: 16-byte row starts (ds_write_b128), operand fetch
                                                             // (16 rows x 4 columns per instruction) spreads over all banks
    constexpr int NLD = NE_T * (BKC / 4);                    // 384 float4 per operand tile
    __shared__ __attribute__((aligned(16))) float sm[2][NE_T][LD];      // Gs, Xs; reused for the final 48x48 reduction
    float (*Gs)[LD] = sm[0];
    float (*Xs)[LD] = sm[1];
    const int split = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;

    f32x4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    const int64_t units = (int64_t)n_batch * units_per_batch;
    float4 rg[2], rx[2];
    // the unit's operand bases and column range change once per unit (~chunk / 32 steps): the 64-bit division that finds them is
    // not in the per-step path (it cost ~200 scalar instructions per 32-column step)
    const float* Gp_u = G;
    const float* Xp_u = X;
    int c0 = 0, c_end_u = 0;
    auto open_unit = [&](int64_t uu) {
        const int batch = (int)(uu / units_per_batch);
        c0 = (int)(uu % units_per_batch) * chunk;
        c_end_u = min(Nc, c0 + chunk);
        const int bo = batch / n_inner, bi = batch % n_inner;
        Gp_u = G + bo * sGo + bi * sGi;
        Xp_u = X + bo * sXo + bi * sXi;
    };
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * NT, row = f / (BKC / 4), c4 = (f % (BKC / 4)) * 4;
            const bool in = f < NLD;
            rg[i] = ld4_guard(Gp_u + (int64_t)row * ldg, c0 + c4, c_end_u, in && row < M, vecG);
            rx[i] = ld4_guard(Xp_u + (int64_t)row * ldx, c0 + c4, c_end_u, in && row < Kx, vecX);
        }
    };
    int64_t u = split;
    bool have = u < units;
    if (have) {
        open_unit(u);
        gload();
    }
    while (have) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + i * NT, row = f / (BKC / 4), c4 = (f % (BKC / 4)) * 4;
            if (f < NLD) {
                *reinterpret_cast<float4*>(&Gs[row][c4]) = rg[i];
                *reinterpret_cast<float4*>(&Xs[row][c4]) = rx[i];
            }
        }
        __syncthreads();
        c0 += BKC;
        if (c0 >= c_end_u) {
            u += nsplit;
            have = u < units;
            if (have) open_unit(u);
        }
        if (have) gload();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = 4 * (2 * w + q) + lq;            // this lane's column of the k4-step
            float a[3], b[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a[i] = Gs[i * 16 + li][col];
                b[i] = Xs[i * 16 + li][col];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // add the four waves' accumulators in wave order through LDS, then store this split's partial
    float* tile = &sm[0][0][0];                              // 48 x 48 floats fit in the two staging tiles
    for (int ww = 0; ww < 4; ++ww) {
        __syncthreads();
        if (w == ww) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* t = tile + (i * 16 + 4 * lq + r) * NE_T + j * 16 + li;
                        *t = ww == 0 ? acc[i][j][r] : *t + acc[i][j][r];
                    }
        }
    }
    __syncthreads();
    float* P = part + (int64_t)split * M * Kx;
    if (fuse.tickets == nullptr) {
        for (int e = tid; e < M * Kx; e += NT) P[e] = tile[(e / Kx) * NE_T + e % Kx];
        return;
    }
    // (the 48 x 48 tile is dense in LDS after the wave reduction; its padding rows / columns are zeros)
    nt_fused_reduce<NT, NE_T, NE_T>(fuse, part, tile, nsplit, split, 0, 1, M, Kx, 0, 0, reinterpret_cast<int*>(tile + NE_T * NE_T));
}

// Sums the split partials in a FIXED order (deterministic).  64 outputs per workgroup, 16 waves each own every
// 16th split, the 16 partial sums are then added in wave order.  blockIdx.y = slice of `slice_len` splits: a launch with
// gridDim.y > 1 writes one partial per slice to `stage` ([slices][M*Kx]) instead of dW (first level of the two-level
// reduction used when few outputs face many splits: 42x42 EdgeConv gradients over ~1900 splits kept 28 workgroups busy
// for 55-100 us; two levels spread the same reads over ~900 workgroups).
__global__ __launch_bounds__(256) void split_reduce_kernel(const float* __restrict__ part, int nsplit, int slice_len, int M,
                                                           int Kx, float* __restrict__ dW, int lddw, int accumulate,
                                                           float* __restrict__ stage) {
    // A workgroup owns 256 consecutive outputs; wave w sums the splits s0 + w, s0 + w + 4, ... of its slice, four outputs per
    // lane (each load is one coalesced 256-byte row piece; eight in flight per lane), then the four wave sums are added in wave
    // order.  (The first form gave every 64 outputs a 1024-thread workgroup with at most two loads per thread: 58 k waves and
    // 105 us for vn1's 26 MB of partials.)  Fixed order: bit-reproducible.
    __shared__ float sh[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t e0 = (int64_t)blockIdx.x * 256 + lane;
    const int64_t ne = (int64_t)M * Kx;
    const int s0 = blockIdx.y * slice_len, s1 = min(nsplit, s0 + slice_len);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    bool in[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) in[j] = e0 + 64 * j < ne;
    int i = s0 + w;
    for (; i + 4 < s1; i += 8) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = in[j] ? part[(int64_t)i * ne + e0 + 64 * j] : 0.f;
            b[j] = in[j] ? part[(int64_t)(i + 4) * ne + e0 + 64 * j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (acc[j] + a[j]) + b[j];
    }
    if (i < s1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += in[j] ? part[(int64_t)i * ne + e0 + 64 * j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[w][lane + 64 * j] = acc[j];
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < ne) {
        const float t = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
        if (stage) {
            stage[(int64_t)blockIdx.y * ne + e] = t;
        } else {
            const int row = (int)(e / Kx), col = (int)(e % Kx);
            float* d = dW + (int64_t)row * lddw + col;
            *d = accumulate ? *d + t : t;
        }
    }
}

// Weight gradient of the LARGE layers (VNStdFeature's vn1 / vn2: 682 x 341 and 340 x 341 over 98 k columns, 46 + 23 GFLOP):
// 128 x 128 output tile, 64 x 64 per wave (four accumulators: every operand fetched from LDS feeds two MFMAs), 16-deep
// steps, double-buffered LDS.  Both operands are reduction-contiguous in memory ("NT"), the MFMA wants them reduction-MAJOR:
// the 16-byte global pieces are transposed on their way into LDS (four ds_write_b32, row pitch 132: the inherent two-way
// conflict of 64 lanes on 32 banks), after which the inner loop is the forward GEMM's (conflict-free ds_read_b32, operand
// mapping of tn_tile).  The 64 x 64 tile kernel below moves 16 flop per operand byte through L2 -> LDS and reached 60-70
// TFLOP/s on these two shapes while starving the main stream's streaming kernels next to it; this one moves 32.
// Split partials go to `part` (summed by split_reduce_kernel: two more launches for two GEMMs per step).
#ifndef HPCS_NB_BK
#define HPCS_NB_BK 16      // 32-deep steps (66 KB of LDS, two workgroups per CU) measured 593 vs 614 us on vn1 and the same step time
#endif
constexpr int NB_BK = HPCS_NB_BK, NB_P = 132;
constexpr int NB_PIECES = NB_BK / 4;          // 16-byte pieces per operand row and step
constexpr int NB_LD = 128 * NB_PIECES / 256;   // pieces per thread and operand
// FUSED: the in-launch split reduction (off by default) -- its eight 16-byte partial loads in flight next to the 64 live
// accumulators spill at 128 registers, so that instantiation is built for two workgroups per CU (256 registers).
template <bool FUSED>
__global__ __launch_bounds__(256, (NB_BK == 16 && !FUSED) ? 4 : 2) void gemm_nt_big_kernel(
    const float* __restrict__ G, int ldg, int64_t sGo, int64_t sGi, const float* __restrict__ X, int ldx, int64_t sXo,
    int64_t sXi, int M, int Kx, int Nc, int n_inner, int n_batch, int chunk, int units_per_batch, int nsplit,
    float* __restrict__ part, int tilesM, int tilesK, int vecG, int vecX, NtFuse fuse) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // 2 * NB_BK * 2 * NB_P floats (dynamic: 66 KB at 32-deep steps)
    float (*As)[NB_BK][NB_P] = reinterpret_cast<float (*)[NB_BK][NB_P]>(lds);
    float (*Bs)[NB_BK][NB_P] = reinterpret_cast<float (*)[NB_BK][NB_P]>(lds + 2 * NB_BK * NB_P);
    // XCD-aware order: the tiles of one split (same column range of G and X) share an L2 (541 vs 550 us, 282 vs 292 us)
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
    // The reduction is ONE list of 16-column steps over all batches (steps_per_batch = ceil(Nc / 16)); split s owns the
    // contiguous range [s T / nsplit, (s + 1) T / nsplit): equal work per workgroup whatever the batch count (the unit lists
    // of the 64 x 64 kernel gave 56 splits of 64 units one or two units each).  `chunk` / `units_per_batch` are unused here.
    (void)chunk; (void)units_per_batch;
    const int spb = (Nc + NB_BK - 1) / NB_BK;
    const int64_t T = (int64_t)n_batch * spb;
    int64_t g = (int64_t)split * T / nsplit;
    const int64_t g_end = (int64_t)(split + 1) * T / nsplit;
    // 128 rows x 16 columns per operand and step = 512 16-byte pieces: two per thread (row = piece / 4)
    float4 rg[NB_LD], rx[NB_LD];
    auto gload = [&](int64_t gs) {
        const int batch = (int)(gs / spb);
        const int c0 = (int)(gs % spb) * NB_BK;
        const int bo = batch / n_inner, bi = batch % n_inner;
        const float* Gp = G + bo * sGo + bi * sGi;
        const float* Xp = X + bo * sXo + bi * sXi;
#pragma unroll
        for (int i = 0; i < NB_LD; ++i) {
            const int f = tid + i * 256, row = f / NB_PIECES, c4 = (f % NB_PIECES) * 4;
            rg[i] = ld4_guard(Gp + (int64_t)(m0 + row) * ldg, c0 + c4, Nc, m0 + row < M, vecG);
            rx[i] = ld4_guard(Xp + (int64_t)(x0 + row) * ldx, c0 + c4, Nc, x0 + row < Kx, vecX);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NB_LD; ++i) {
            const int f = tid + i * 256, row = f / NB_PIECES, c4 = (f % NB_PIECES) * 4;
            As[buf][c4][row] = rg[i].x; As[buf][c4 + 1][row] = rg[i].y; As[buf][c4 + 2][row] = rg[i].z; As[buf][c4 + 3][row] = rg[i].w;
            Bs[buf][c4][row] = rx[i].x; Bs[buf][c4 + 1][row] = rx[i].y; Bs[buf][c4 + 2][row] = rx[i].z; Bs[buf][c4 + 3][row] = rx[i].w;
        }
    };
    // 32 x 32 blocks that lie entirely in the padding of an edge tile issue no MFMA (682 x 341 in 128 x 128 tiles: 242 of 288)
    bool live[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) live[i][j] = (m0 + wm * 64 + i * 32 < M) && (x0 + wn * 64 + j * 32 < Kx);
    if (g < g_end) {
        gload(g);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (; g < g_end; ++g) {
        const bool have_next = g + 1 < g_end;
        if (have_next) gload(g + 1);
#pragma unroll
        for (int kk = 0; kk < NB_BK / 2; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[buf][kk * 2 + lk][wm * 64 + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[buf][kk * 2 + lk][wn * 64 + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (live[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (have_next) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if constexpr (FUSED) {
        // In-launch split reduction (nt_fused_reduce), one 64 x 64 QUARTER of the tile at a time: quarter q is wave q's four
        // accumulators, staged dense in the (dead) operand LDS; partials, tickets and the final sum are per quarter tile
        // (tile * 4 + q of 4 * tiles).  Quarters that lie entirely in the padding are skipped by every workgroup alike.
        float* tl = lds;                                       // 64 x 64 floats + the arrival flag behind them
        for (int q = 0; q < 4; ++q) {
            const int qm = q >> 1, qn = q & 1;
            const int m0q = m0 + qm * 64, x0q = x0 + qn * 64;
            if (m0q >= M || x0q >= Kx) continue;               // (uniform over the workgroup)
            __syncthreads();                                   // main loop / previous quarter done with the LDS
            if (wave == q) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            tl[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + lr] = live[i][j] ? acc[i][j][r] : 0.f;
            }
            __syncthreads();
            nt_fused_reduce<256, 64, 64>(fuse, part, tl, nsplit, split, tile * 4 + q, tilesM * tilesK * 4, M, Kx, m0q, x0q,
                                         reinterpret_cast<int*>(tl + 64 * 64));
        }
    } else {
        (void)fuse;
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
}

static size_t nb_lds_bytes() {
    static const size_t bytes = [] {
        const size_t b = (size_t)2 * NB_BK * 2 * NB_P * sizeof(float);
        (void)hipFuncSetAttribute((const void*)gemm_nt_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
        return b;
    }();
    return bytes;
}

constexpr int SR_SLICES = 32;   // first-level slices of the two-level split reduction
inline bool sr_two_level(int nsplit, int64_t ne) { return nsplit >= 256 && ne <= 64 * 512; }

constexpr int NT_BKC = 32;      // 64-wide steps measured: +9 % on the HBM-bound EdgeConv shapes, -10 % on the large ones

struct NtPlan { int tile, tilesM, tilesK, chunk, units_per_batch, nsplit; };
static long nt_wg_target() {
    // 768 = three workgroups per CU: 2048 (eight) measured 1.3 % slower in the step, 10.25-10.29 vs 10.10-10.15 ms -- the
    // weight gradients run next to the main stream's kernels and fewer, longer workgroups leave those more of the chip
    const long t = hpcs_tuning().nt_wg_target > 0 ? hpcs_tuning().nt_wg_target : 768;
    return t < 64 ? 64 : t;
}
// (Measured and NOT kept, r03: capping this GEMM's residency with unused dynamic LDS -- 4 / 3 / 2 workgroups per CU instead of
// 8 -- so that the main stream's streaming kernels find free wave slots next to it: 10.61 -> 10.55 / 10.53 / 10.58 ms per step,
// i.e. nothing.  The side stream slows them through the memory system, not through occupancy.)
NtPlan nt_plan(int M, int Kx, int Nc, int n_batch) {
    NtPlan p;
    const long WG = nt_wg_target();
    // 128 x 128 tiles (gemm_nt_big_kernel) from 300 x 300 outputs up: VNStdFeature's vn1 / vn2 weight gradients
    // (r04: from 128 x 128 up once the 128-tile kernel runs on the bf16 matrix cores: the heads' 256 x 256 gradients too)
    const int big_min = hpcs_tuning().nt_big_min > 0 ? hpcs_tuning().nt_big_min : (hpcs_tuning().gemm_x3 ? 128 : 300);
    p.tile = (M >= big_min && Kx >= big_min) ? 128 : 64;
    p.tilesM = hpcs_cdiv(M, p.tile);
    p.tilesK = hpcs_cdiv(Kx, p.tile);
    // one output tile (EdgeConv weight gradients: M, K <= 64 over ~2 M columns): MFMA (102 us) and HBM (147 us) phases of a
    // workgroup do not overlap, so more, shorter workgroups hide more: 1024-column chunks -> 8 workgroups per CU
    // (measured 249 -> 224 us at K = 42, 216 -> 200 us at K = 21)
    p.chunk = p.tilesM * p.tilesK == 1 ? 1024 : 2048;
    {   // too few workgroups for the chip (row-oriented bias GEMMs, conv6 / head weight gradients with 1-2 units per
        // cloud): shrink the chunk until ~2048 workgroups share the columns, as long as the split partials
        // (nsplit * M * Kx floats) stay small
        const long tiles = (long)p.tilesM * p.tilesK;
        const long units0 = (long)n_batch * hpcs_cdiv(Nc, p.chunk);
        const long split_cap = (8L << 20) / ((long)M * Kx);              // partial buffer <= 32 MB
        if (units0 * tiles < WG * 3 / 4 && split_cap > units0) {
            long target_units = WG / tiles > 0 ? WG / tiles : 1;
            if (target_units > split_cap) target_units = split_cap;
            long c = hpcs_cdiv((int64_t)Nc * n_batch, target_units);
            c = (c + 31) / 32 * 32;
            if (c < 64) c = 64;
            if (c < p.chunk) p.chunk = (int)c;
        }
    }
    if (p.tile == 128) {
        // gemm_nt_big_kernel: one round of workgroups, the step list cut into equal ranges
        p.chunk = Nc;
        p.units_per_batch = 1;
        const long T = (long)n_batch * hpcs_cdiv(Nc, NB_BK);
        // two workgroups per CU (512): 10.01-10.06 ms per step against 10.08-10.13 with four (1024); 256: 10.06
        const long big_wg = hpcs_tuning().nt_big_wg;
        long ns = big_wg / ((long)p.tilesM * p.tilesK);
        if (ns < 1) ns = 1;
        if (ns > T) ns = T;
        p.nsplit = (int)ns;
        return p;
    }
    p.units_per_batch = hpcs_cdiv(Nc, p.chunk);
    const long units = (long)n_batch * p.units_per_batch;
    long want = WG / ((long)p.tilesM * p.tilesK);
    if (want < 1) want = 1;
    p.nsplit = (int)(units < want ? units : want);
    if (p.nsplit < 1) p.nsplit = 1;
    return p;
}

// One launch builds both GEMM operands of a (stacked) weight matrix W = [W1; W2] (rows M1 + M2, K columns):
//   At [K][lda]  = W^T, zero padded to lda = pad4(M)   (forward:  C = At^T X)
//   Wp [M][ldp]  = W,   zero padded to ldp = pad4(K)   (data-grad: gX = Wp^T G)
__global__ void prep_weights_kernel(const float* __restrict__ W1, int ld1, int M1, const float* __restrict__ W2, int ld2,
                                    int M2, int K, float* __restrict__ At, int lda, float* __restrict__ Wp, int ldp) {
    const int M = M1 + M2;
    const int64_t nA = (int64_t)K * lda, nP = (int64_t)M * ldp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nA + nP; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < nA) {
            const int k = (int)(e / lda), m = (int)(e % lda);
            float v = 0.f;
            if (m < M1) v = W1[(int64_t)m * ld1 + k];
            else if (m < M) v = W2[(int64_t)(m - M1) * ld2 + k];
            At[e] = v;
        } else {
            const int64_t q = e - nA;
            const int m = (int)(q / ldp), k = (int)(q % ldp);
            float v = 0.f;
            if (k < K) v = m < M1 ? W1[(int64_t)m * ld1 + k] : W2[(int64_t)(m - M1) * ld2 + k];
            Wp[q] = v;
        }
    }
}

__global__ void prep_weights_multi_kernel(const hpcs_prep_desc* __restrict__ desc) {
    const hpcs_prep_desc d = desc[blockIdx.y];
    const float* W1 = reinterpret_cast<const float*>(d.W1);
    const float* W2 = reinterpret_cast<const float*>(d.W2);
    float* At = reinterpret_cast<float*>(d.At);
    float* Wp = reinterpret_cast<float*>(d.Wp);
    const int M = d.M1 + d.M2;
    const int64_t nA = (int64_t)d.K * d.lda, nP = (int64_t)M * d.ldp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nA + nP; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < nA) {
            const int k = (int)(e / d.lda), m = (int)(e % d.lda);
            float v = 0.f;
            if (m < d.M1) v = W1[(int64_t)m * d.ld1 + k];
            else if (m < M) v = W2[(int64_t)(m - d.M1) * d.ld2 + k];
            At[e] = v;
        } else {
            const int64_t q = e - nA;
            const int m = (int)(q / d.ldp), k = (int)(q % d.ldp);
            float v = 0.f;
            if (k < d.K) v = m < d.M1 ? W1[(int64_t)m * d.ld1 + k] : W2[(int64_t)(m - d.M1) * d.ld2 + k];
            Wp[q] = v;
        }
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// Point-level form of an EdgeConv layer's weights (VNGraphLayerFn.backward): W = [Wa | Wb] (difference | centre columns) ->
// Wuv = [Wa ; Wb - Wa], and the way back for the gradient: dWa = gU-part - gV-part, dWb = gV-part, ADDED into the two
// parameters' gradient slots (rows < Cout: map_to_feat, the rest: map_to_dir).  One launch each (was cat + sub + copy and
// sub + cat + two adds of torch element-wise kernels per layer).
__global__ void graph_wuv_kernel(const float* __restrict__ W, int ldw, int Ctot, int C, float* __restrict__ Wuv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * Ctot * C) return;
    const int r = i / C, c = i % C;
    Wuv[i] = r < Ctot ? W[(int64_t)r * ldw + c] : W[(int64_t)(r - Ctot) * ldw + C + c] - W[(int64_t)(r - Ctot) * ldw + c];
}
__global__ void graph_wgrad_fold_kernel(const float* __restrict__ gWuv, int Ctot, int C, int Cout, float* __restrict__ gf,
                                        float* __restrict__ gd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ctot * 2 * C) return;
    const int r = i / (2 * C), c = i % (2 * C);
    const float v = c < C ? gWuv[(int64_t)r * C + c] - gWuv[(int64_t)(Ctot + r) * C + c] : gWuv[(int64_t)(Ctot + r) * C + (c - C)];
    float* dst = r < Cout ? gf + (int64_t)r * 2 * C + c : gd + (int64_t)(r - Cout) * 2 * C + c;
    if (r < Cout ? gf != nullptr : gd != nullptr) *dst += v;
}

}  // namespace

#ifdef HPCS_TN_STAMPS
extern "C" int hpcs_debug_tn_stamps(long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tn_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int hpcs_gemm_tn(const float* At, int lda, const float* Bm, int ldb, int64_t strideB_o, int64_t strideB_i,
                            float* C, int ldc, int64_t strideC_o, int64_t strideC_i, int M, int Nc, int K, int n_outer,
                            int n_inner, const float* rowbias, int accumulate, hpcs_stream_t stream) {
    if (!At || !Bm || !C) return HPCS_ERR_NULL;
    if (M <= 0 || Nc <= 0 || K <= 0 || n_outer < 0 || n_inner <= 0 || lda < M || ldb < Nc || ldc < Nc)
        return HPCS_ERR_SHAPE;
    if (n_outer == 0) return HPCS_OK;
    hipStream_t st = (hipStream_t)stream;
    const int vecA = (lda % 4 == 0) && aligned16(At);
    const int vecB = (ldb % 4 == 0) && (strideB_o % 4 == 0) && (strideB_i % 4 == 0) && aligned16(Bm);
    const int vecC_env = hpcs_tuning().gemm_vecstore;
    const int vecC = vecC_env && (ldc % 4 == 0) && (strideC_o % 4 == 0) && (strideC_i % 4 == 0) && aligned16(C);
    const int nb = n_outer * n_inner;
    {   // fewer than 32 tiles of 128x128 over all batches and a short enough reduction: the per-element kernel
        const long tiles = (long)hpcs_cdiv(M, 128) * hpcs_cdiv(Nc, 128) * nb;
        if (tiles < 32 && Nc >= 16 && (long)K <= 4096) {
            const int colBlocks = hpcs_cdiv(Nc, 32), rowBlocks = hpcs_cdiv(M, SMALL_RB);
            const long blocks = (long)colBlocks * rowBlocks * nb;
            hipLaunchKernelGGL(gemm_tn_small_kernel, dim3((unsigned)blocks), dim3(256), 0, st, At, lda, Bm, ldb, strideB_o,
                               strideB_i, C, ldc, strideC_o, strideC_i, M, Nc, K, n_inner, rowbias, accumulate, colBlocks,
                               rowBlocks);
            HPCS_CHECK_LAUNCH();
            return HPCS_OK;
        }
    }
    // straight-line loader: every 16-byte load of both operands is in bounds after clamping (see tn_tile)
    const bool fast = vecA && vecB && Nc % 4 == 0 && Nc >= 4 && lda >= 4;
    // A/B switch: HPCS_GEMM_TN_GUARDED=1 forces the guarded loader
    const int guarded = hpcs_tuning().gemm_tn_guarded;
#define HPCS_TN_LAUNCH(BM_, WM_, WN_, FAST_)                                                                                 \
    hipLaunchKernelGGL((gemm_tn_kernel<BM_, 128, 16, WM_, WN_, FAST_>), dim3((unsigned)total), dim3(256), 0, st, At, lda, Bm,     \
                       ldb, strideB_o, strideB_i, C, ldc, strideC_o, strideC_i, M, Nc, K, n_inner, rowbias, accumulate, tm, tn,  \
                       (unsigned)total, vecA, vecB, vecC)
    // The large dense layers (M > 64) take the bf16x3 kernel (csrc/gemm_x3.hip): fp32-accurate products on the bf16 matrix
    // cores, 1.4-1.6x the fma-chain kernel below; HPCS_GEMM_X3=0 keeps the fma chain (A/B, and the form whose k-ordered
    // rounding sequence the kNN and the graph-operand GEMM are pinned to -- those never come here).
    // (its loads address one operand matrix with 32-bit byte offsets)
    if (M > 64 && hpcs_tuning().gemm_x3 && (int64_t)(K + 16) * lda * 4 < ((int64_t)1 << 31) && (int64_t)(K + 16) * ldb * 4 < ((int64_t)1 << 31))
        return hpcs_gemm_tn_x3(At, lda, Bm, ldb, strideB_o, strideB_i, C, ldc, strideC_o, strideC_i, M, Nc, K, n_outer, n_inner,
                               rowbias, accumulate, stream);
    const int BMsel = M > 64 ? 128 : 64;
    const int tm = hpcs_cdiv(M, BMsel), tn = hpcs_cdiv(Nc, 128);
    const long total = (long)tm * tn * nb;
    if (total > 0x7fffffffL) return HPCS_ERR_UNSUPPORTED;
    if (M > 64) {
        if (fast && !guarded) HPCS_TN_LAUNCH(128, 2, 2, true);
        else HPCS_TN_LAUNCH(128, 2, 2, false);
    } else {
        if (fast && !guarded) HPCS_TN_LAUNCH(64, 1, 4, true);
        else HPCS_TN_LAUNCH(64, 1, 4, false);
    }
#undef HPCS_TN_LAUNCH
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" size_t hpcs_gemm_tn_graph_workspace(int B, int C, int N) {
    if (B <= 0 || C <= 0 || N <= 0) return 0;
    return (size_t)B * 3 * N * ((C + 3) & ~3) * sizeof(float);
}

extern "C" int hpcs_gemm_tn_graph(const float* At, int lda, const float* feat, const int64_t* idx, int B, int C, int N, int k,
                                  int M, float* PD, void* ws, size_t ws_bytes, hpcs_stream_t stream) {
    if (!At || !feat || !idx || !PD || !ws) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || k <= 0 || M <= 0 || lda < M) return HPCS_ERR_SHAPE;
    // the loader's assumptions: 16-byte pieces never straddle a point's neighbour list, straight-line (clamped) loads
    if (M > GM_BM || 2 * C > GM_KMAX || k % 4 != 0 || lda % 4 != 0 || lda < 4 || !aligned16(At) || !aligned16(PD) ||
        (int64_t)3 * N * k > 0x7fffffffL)
        return HPCS_ERR_UNSUPPORTED;
    if (B == 0) return HPCS_OK;
    if (B > 65535 || ws_bytes < hpcs_gemm_tn_graph_workspace(B, C, N) || !aligned16(ws)) return HPCS_ERR_WORKSPACE;
    const int Cp = (C + 3) & ~3;
    float* featT = (float*)ws;
    hipLaunchKernelGGL(feat_transpose_kernel, dim3((unsigned)hpcs_cdiv((long)3 * N * (Cp / 4), 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, feat, C, Cp, N, featT);
    HPCS_CHECK_LAUNCH();
    const int Nc = 3 * N * k;
    const int tn = hpcs_cdiv(Nc, 128);
    const long total = (long)tn * B;
    if (total > 0x7fffffffL) return HPCS_ERR_UNSUPPORTED;
    const int Kp = (2 * C + 1) & ~1;
    size_t lds = (size_t)Kp * (GM_BM + GM_BN) * sizeof(float);
    const size_t patches = (size_t)4 * 32 * 36 * sizeof(float);
    if (lds < patches) lds = patches;
    hipLaunchKernelGGL(gemm_tn_graph_kernel, dim3((unsigned)total), dim3(256), lds, (hipStream_t)stream, At, lda, featT, idx, C, N,
                       k, PD, M, tn, (unsigned)total);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" size_t hpcs_gemm_nt_workspace(int M, int Kx, int Nc, int n_outer, int n_inner) {
    if (M <= 0 || Kx <= 0 || Nc <= 0 || n_outer <= 0 || n_inner <= 0) return 0;
    const NtPlan p = nt_plan(M, Kx, Nc, n_outer * n_inner);
    const size_t stage = sr_two_level(p.nsplit, (int64_t)M * Kx) ? (size_t)SR_SLICES * M * Kx : 0;
    const size_t plain = ((size_t)p.nsplit * M * Kx + stage) * sizeof(float);
    // the in-launch reduction keeps its partials tile-major and dense (whole tiles, padding included)
    const size_t te = (p.tilesM * p.tilesK == 1 && M <= NE_T && Kx <= NE_T) ? (size_t)NE_T * NE_T : (size_t)p.tile * p.tile;
    const size_t fused = ((size_t)p.nsplit + SR_SLICES) * p.tilesM * p.tilesK * te * sizeof(float);
    return plain > fused ? plain : fused;
}

// number of 32-bit tickets hpcs_gemm_nt_fused needs (zeroed by the caller before the FIRST use; left zero by every call)
extern "C" size_t hpcs_gemm_nt_tickets(int M, int Kx, int Nc, int n_outer, int n_inner) {
    if (M <= 0 || Kx <= 0 || Nc <= 0 || n_outer <= 0 || n_inner <= 0) return 0;
    const NtPlan p = nt_plan(M, Kx, Nc, n_outer * n_inner);
    const bool two = sr_two_level(p.nsplit, (int64_t)M * Kx);
    const int slice_len = two ? hpcs_cdiv(p.nsplit, SR_SLICES) : p.nsplit;
    const int slices = hpcs_cdiv(p.nsplit, slice_len);
    return (size_t)p.tilesM * p.tilesK * (p.tile == 128 ? 4 : 1) * (slices + 1);
}

static int gemm_nt_impl(const float* G, int ldg, int64_t strideG_o, int64_t strideG_i, const float* X, int ldx,
                        int64_t strideX_o, int64_t strideX_i, int M, int Kx, int Nc, int n_outer, int n_inner,
                        float* dW, int lddw, int accumulate, void* ws, size_t ws_bytes, unsigned* tickets, size_t n_tickets,
                        hpcs_stream_t stream) {
    if (!G || !X || !dW || !ws) return HPCS_ERR_NULL;
    if (M <= 0 || Kx <= 0 || Nc <= 0 || n_outer <= 0 || n_inner <= 0 || ldg < Nc || ldx < Nc || lddw < Kx)
        return HPCS_ERR_SHAPE;
    const int nb = n_outer * n_inner;
    const NtPlan p = nt_plan(M, Kx, Nc, nb);
    const bool two_level = sr_two_level(p.nsplit, (int64_t)M * Kx);
    if (ws_bytes < hpcs_gemm_nt_workspace(M, Kx, Nc, n_outer, n_inner)) return HPCS_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int vecG = (ldg % 4 == 0) && (strideG_o % 4 == 0) && (strideG_i % 4 == 0) && aligned16(G);
    const int vecX = (ldx % 4 == 0) && (strideX_o % 4 == 0) && (strideX_i % 4 == 0) && aligned16(X);
    const long blocks = (long)p.tilesM * p.tilesK * p.nsplit;
    const int64_t ne = (int64_t)M * Kx;
    const int slice_len = two_level ? hpcs_cdiv(p.nsplit, SR_SLICES) : p.nsplit;
    const int slices = hpcs_cdiv(p.nsplit, slice_len);
    float* stage = (float*)ws + (size_t)p.nsplit * ne;
    const bool edge = p.tilesM * p.tilesK == 1 && M <= NE_T && Kx <= NE_T;
    NtFuse fuse{nullptr, stage, dW, lddw, accumulate, slice_len, slices, 0u, 0u};
    const size_t rtiles = (size_t)p.tilesM * p.tilesK * (p.tile == 128 ? 4 : 1);      // reduction units: 64 x 64 quarters of a big tile
    if (tickets && n_tickets < rtiles * (slices + 1)) return HPCS_ERR_WORKSPACE;
    // (gemm_nt_big_kernel's own in-launch reduction -- 64 x 64 quarter tiles through nt_fused_reduce -- cost it ~75 us per launch,
    // 392 / 687 us against 316 / 614 us + two split_reduce launches, with the step unchanged: measured in round 3, the
    // instantiation and its switch were removed in round 4; the 128-tile kernel writes plain partials)
    if (tickets && p.tile != 128) {
        const size_t te = edge ? (size_t)NE_T * NE_T : (p.tile == 128 ? (size_t)64 * 64 : (size_t)p.tile * p.tile);
        const size_t pb = (size_t)p.nsplit * rtiles * te * sizeof(float);
        const size_t sb = (size_t)slices * rtiles * te * sizeof(float);
        if (pb >= (1ull << 31) || sb >= (1ull << 31)) return HPCS_ERR_UNSUPPORTED;
        fuse.tickets = tickets;
        fuse.stage = (float*)((char*)ws + pb);
        fuse.part_bytes = (unsigned)pb;
        fuse.stage_bytes = (unsigned)(sb ? sb : 16);
    }
    if (p.tilesM * p.tilesK == 1 && M <= NE_T && Kx <= NE_T)
        hipLaunchKernelGGL(gemm_nt_edge_kernel, dim3((unsigned)blocks), dim3(256), 0, st, G, ldg, strideG_o, strideG_i, X, ldx,
                           strideX_o, strideX_i, M, Kx, Nc, n_inner, nb, p.chunk, p.units_per_batch, p.nsplit, (float*)ws, vecG,
                           vecX, fuse);
    else if (p.tile == 128 && hpcs_tuning().gemm_x3) {
        // the same tile and work split on the bf16 matrix cores at fp32 accuracy (csrc/gemm_x3.hip)
        if (const int rc = hpcs_x3_nt_big_launch(G, ldg, strideG_o, strideG_i, X, ldx, strideX_o, strideX_i, M, Kx, Nc, n_inner, nb,
                                                 p.nsplit, (float*)ws, p.tilesM, p.tilesK, vecG, vecX, blocks, st))
            return rc;
    } else if (p.tile == 128)
        hipLaunchKernelGGL(gemm_nt_big_kernel<false>, dim3((unsigned)blocks), dim3(256), nb_lds_bytes(), st, G, ldg, strideG_o, strideG_i, X,
                           ldx, strideX_o, strideX_i, M, Kx, Nc, n_inner, nb, p.chunk, p.units_per_batch, p.nsplit, (float*)ws,
                           p.tilesM, p.tilesK, vecG, vecX, fuse);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<64, 64, NT_BKC, 2, 2>), dim3((unsigned)blocks), dim3(256), 0, st, G, ldg,
                           strideG_o, strideG_i, X, ldx, strideX_o, strideX_i, M, Kx, Nc, n_inner, nb, p.chunk,
                           p.units_per_batch, p.nsplit, (float*)ws, p.tilesM, p.tilesK, vecG, vecX, fuse);
    HPCS_CHECK_LAUNCH();
    if (fuse.tickets) return HPCS_OK;
    if (two_level) {
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)hpcs_cdiv(ne, 256), (unsigned)slices), dim3(256), 0, st,
                           (const float*)ws, p.nsplit, slice_len, M, Kx, dW, lddw, 0, stage);
        HPCS_CHECK_LAUNCH();
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)hpcs_cdiv(ne, 256)), dim3(256), 0, st, (const float*)stage,
                           slices, slices, M, Kx, dW, lddw, accumulate, (float*)nullptr);
    } else {
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)hpcs_cdiv(ne, 256)), dim3(256), 0, st, (const float*)ws,
                           p.nsplit, p.nsplit, M, Kx, dW, lddw, accumulate, (float*)nullptr);
    }
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_gemm_nt(const float* G, int ldg, int64_t strideG_o, int64_t strideG_i, const float* X, int ldx,
                            int64_t strideX_o, int64_t strideX_i, int M, int Kx, int Nc, int n_outer, int n_inner,
                            float* dW, int lddw, int accumulate, void* ws, size_t ws_bytes, hpcs_stream_t stream) {
    return gemm_nt_impl(G, ldg, strideG_o, strideG_i, X, ldx, strideX_o, strideX_i, M, Kx, Nc, n_outer, n_inner, dW, lddw,
                        accumulate, ws, ws_bytes, nullptr, 0, stream);
}

extern "C" int hpcs_gemm_nt_fused(const float* G, int ldg, int64_t strideG_o, int64_t strideG_i, const float* X, int ldx,
                                  int64_t strideX_o, int64_t strideX_i, int M, int Kx, int Nc, int n_outer, int n_inner,
                                  float* dW, int lddw, int accumulate, void* ws, size_t ws_bytes, int32_t* tickets,
                                  size_t n_tickets, hpcs_stream_t stream) {
    if (!tickets) return HPCS_ERR_NULL;
    return gemm_nt_impl(G, ldg, strideG_o, strideG_i, X, ldx, strideX_o, strideX_i, M, Kx, Nc, n_outer, n_inner, dW, lddw,
                        accumulate, ws, ws_bytes, (unsigned*)tickets, n_tickets, stream);
}

extern "C" int hpcs_prep_weights(const float* W1, int ld1, int M1, const float* W2, int ld2, int M2, int K, float* At, int lda,
                                 float* Wp, int ldp, hpcs_stream_t stream) {
    if (!W1 || !At || !Wp || (M2 > 0 && !W2)) return HPCS_ERR_NULL;
    if (M1 <= 0 || M2 < 0 || K <= 0 || ld1 < K || (M2 > 0 && ld2 < K) || lda < M1 + M2 || ldp < K) return HPCS_ERR_SHAPE;
    const int64_t n = (int64_t)K * lda + (int64_t)(M1 + M2) * ldp;
    int blocks = hpcs_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(prep_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W1, ld1, M1, W2, ld2, M2, K, At, lda,
                       Wp, ldp);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_graph_wuv(const float* W, int ldw, int Ctot, int C, float* Wuv, hpcs_stream_t stream) {
    if (!W || !Wuv) return HPCS_ERR_NULL;
    if (Ctot <= 0 || C <= 0 || ldw < 2 * C) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(graph_wuv_kernel, dim3(hpcs_cdiv(2 * Ctot * C, 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, Ctot, C, Wuv);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_graph_wgrad_fold(const float* gWuv, int Ctot, int C, int Cout, float* gWf, float* gWd, hpcs_stream_t stream) {
    if (!gWuv || (!gWf && !gWd)) return HPCS_ERR_NULL;
    if (Ctot <= 0 || C <= 0 || Cout < 0 || Cout > Ctot) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(graph_wgrad_fold_kernel, dim3(hpcs_cdiv(Ctot * 2 * C, 256)), dim3(256), 0, (hipStream_t)stream, gWuv, Ctot, C,
                       Cout, gWf, gWd);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_prep_weights_multi(const hpcs_prep_desc* desc_dev, int n, hpcs_stream_t stream) {
    if (!desc_dev) return HPCS_ERR_NULL;
    if (n < 0 || n > 65535) return HPCS_ERR_SHAPE;
    if (n == 0) return HPCS_OK;
    hipLaunchKernelGGL(prep_weights_multi_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, desc_dev);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
