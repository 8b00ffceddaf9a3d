// EdgeConv edge-level Vector-Neuron layers WITHOUT their pre-activation tensor (round 4).
//
// A VNLinearLeakyReLU on an edge tensor (conv2 / conv4 of VN_DGCNN_partseg: hpcs/nn/dgcnn/vn_dgcnn_partseg.py:66-75,
// layer = hpcs/nn/dgcnn/utils/vn_layers.py:48-77 with VNBatchNorm :112-132, followed by mean_pool :150-153) maps
// X [B, Cin=21, 3, L=N*k] to PD = [W_feat; W_dir] X [B, 42, 3, L] and then works point-wise on PD.  The streaming design
// materialises PD (330 MB at B=32, N=1024, k=20), reads it in the statistics pass, the apply pass, the backward reduction
// and the backward apply, writes the gradient GPD (330 MB) and reads that back in the data-gradient GEMM: 3.1 GB per layer
// and step.  Here every pass re-forms its PD tile from X (165 MB) on the matrix cores instead:
//
//   hpcs_vn_rc_stats        X -> sum |p|, sum |p|^2 per channel (+ BatchNorm finalisation by the last workgroup)
//   hpcs_vn_rc_apply_fwd    X -> BN + VN-LeakyReLU (+ mean over k) -> out
//   hpcs_vn_rc_bwd_reduce   X, g -> the two BatchNorm backward sums per channel
//   hpcs_vn_rc_bwd_apply    X, g -> GPD (for the side stream's weight-gradient GEMM) and gX = W^T GPD in the same pass
//
// so PD is never written or read, GPD is written once and read once, and the edge-level data-gradient GEMM is gone.
//
// Tile = 16 consecutive positions of one cloud per wave step (v_mfma_f32_16x16x4_f32: A 16x4, B 4x16, D 16x16).
//   * B operand = X straight from global memory: lane (n = lane & 15, g = lane >> 4) loads X[cin = 4 s + g][xyz][l0 + n] for
//     k-step s -- four 64-byte row segments per load instruction, no LDS, no barrier.
//   * A operand = weights, held in registers for the whole kernel.  The 42 stacked rows are laid over three 16-row tiles so
//     that p_c and d_c of one channel land in the SAME lane (D layout: col = lane & 15, row = 4 (lane >> 4) + reg):
//       tile 0 row c      = p_c,  c < 16          tile 1 row c = d_c,  c < 16
//       tile 2 row 4g+2u  = p_c,  row 4g+2u+1 = d_c  for c = 16 + 2g + u   (c < 24)
//     Lane group g then owns channels 4g..4g+3 and 16+2g, 16+2g+1: six (p, d) slots per lane, all three xyz planes in
//     registers -- the point-wise code of vn_point.h runs on them unchanged.
//   * The f32 MFMA is the k-ordered fp32 fma chain, so the recomputed PD is the value the GEMM would have
//     written, and the results of these kernels are held equal to the materialised path's by the tests.
//   * Backward apply: the gradient tile (dL/dp, dL/dd) replaces PD in the accumulator registers and is the B operand of
//     gX = W^T GPD as it stands (an accumulator tile has its column on the lane and its rows in the registers: k-step
//     (tile, reg) sums over rows {reg, 4 + reg, 8 + reg, 12 + reg}; the A operand W^T is pre-arranged to match).
#include "common.h"

namespace {

#include "vn_point.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RC_TPB = 256;                 // four waves
constexpr int RC_WAVES = RC_TPB / 64;
constexpr int RC_KS = 6;                    // k-steps of 4 input channels: Cin <= 24
constexpr int RC_CMAX = 24;                 // Cout = Cd <= 24
constexpr int RC_SLOTS = 6;                 // (p, d) channel slots per lane
constexpr int RC_FLAGS = HPCS_VN_BN | HPCS_VN_LRELU;

// Wp row of MFMA row `m` (0..15) of tile T, or -1:  channel index in *c.
__device__ __forceinline__ int rc_row(int T, int m, int Cout, int* c_out) {
    int c, row;
    if (T == 0) { c = m; row = c; }
    else if (T == 1) { c = m; row = Cout + c; }
    else { c = 16 + 2 * (m >> 2) + ((m >> 1) & 1); row = (m & 1) ? Cout + c : c; }
    if (c_out) *c_out = c;
    return c < Cout ? row : -1;
}
// channel of slot sigma of lane group g
__device__ __forceinline__ int rc_slot_channel(int sigma, int g) { return sigma < 4 ? 4 * g + sigma : 16 + 2 * g + (sigma - 4); }

// forward A operands: Aw[T][s] = W[row(T, n)][cin = 4 s + g]
__device__ __forceinline__ void rc_weights_fwd(const float* __restrict__ Wp, int ldw, int Cout, int Cin, int n, int g,
                                               float (&Aw)[3][RC_KS]) {
#pragma unroll
    for (int T = 0; T < 3; ++T) {
        const int row = rc_row(T, n, Cout, nullptr);
#pragma unroll
        for (int s = 0; s < RC_KS; ++s) {
            const int cin = 4 * s + g;
            Aw[T][s] = (row >= 0 && cin < Cin) ? Wp[(int64_t)row * ldw + cin] : 0.f;
        }
    }
}

// Per-lane, loop-invariant element offsets (32-bit: a cloud's tensors stay below 2^31 elements, checked by the launcher).
// UNSIGNED, and always applied as (uniform pointer)[offset]: that is the shape the backend turns into the scalar-base +
// 32-bit-vector-offset addressing mode; with signed offsets folded into 64-bit pointer arithmetic it precomputed some forty
// per-lane 64-bit addresses outside the loop and spilled them.
struct RcOff {
    unsigned x;        // g * 3 L + n                  X rows 4 s + g  (+ uniform 12 s L)
    unsigned x_tail;   // the partial last k-step (Cin % 4 != 0): row clamped to Cin - 1
    unsigned a;        // g * 12 L + n                 rows of channel 4 g + r (tiles 0 / 1 of GPD, out, g), gX rows 16 mt + 4 g + r
    unsigned b;        // g * 6 L + n                  rows of channel 16 + 2 g + u (tile 2)
};
__device__ __forceinline__ RcOff rc_offsets(int n, int g, int Cin, int64_t L) {
    RcOff o;
    const int L3 = 3 * (int)L;
    o.x = (unsigned)(g * L3 + n);
    const int tail0 = (Cin / 4) * 4;                                    // first row of the partial k-step
    const int gt = tail0 + g < Cin ? g : (Cin - 1 - tail0 > 0 ? Cin - 1 - tail0 : 0);
    o.x_tail = (unsigned)(gt * L3 + n);
    o.a = (unsigned)(g * 4 * L3 + n);
    o.b = (unsigned)(g * 2 * L3 + n);
    return o;
}

// B operands of one 16-position sub-tile: x[a][s] = X[cin = 4 s + g][a][l0 + n].  Xs = X + cloud offset + l0 (uniform, scalar
// registers); rows beyond Cin are zero (the partial k-step reads a clamped row, whole k-steps beyond Cin are not loaded).
__device__ __forceinline__ void rc_load_x(const float* __restrict__ Xs, int Cin, int64_t L, int g, const RcOff& off,
                                          float (&x)[3][RC_KS]) {
#pragma unroll
    for (int s = 0; s < RC_KS; ++s) {
        const float* q = Xs + (int64_t)(4 * s) * 3 * L;                  // uniform
        if (4 * s + 3 < Cin) {
#pragma unroll
            for (int a = 0; a < 3; ++a) x[a][s] = (q + (int64_t)a * L)[off.x];
        } else if (4 * s < Cin) {
            const bool ok = 4 * s + g < Cin;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = (q + (int64_t)a * L)[off.x_tail];
                x[a][s] = ok ? v : 0.f;
            }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) x[a][s] = 0.f;
        }
    }
}

// The A operands live in LDS, one 64-lane row per (tile, k-step): wt[i * 64 + lane].  (Held in registers for the whole kernel
// they were spilled to scratch by the register allocator -- 42 values next to 36 accumulators and the point-wise state -- and
// re-read from there every sub-tile; a conflict-free ds_read_b32 each is the cheaper home.)  Rows 0..17 = forward operands
// [T][s], rows 18..41 = data-gradient operands [mt][T][r].
constexpr int RC_WT_FWD = 3 * RC_KS, RC_WT_BWD = 2 * 3 * 4;
__device__ __forceinline__ void rc_weights_to_lds(float* wt, const float* __restrict__ Wp, int ldw, int Cout, int Cin, bool with_bwd) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int T = 0; T < 3; ++T) {
            const int row = rc_row(T, n, Cout, nullptr);
#pragma unroll
            for (int s = 0; s < RC_KS; ++s) {
                const int cin = 4 * s + g;
                wt[(T * RC_KS + s) * 64 + lane] = (row >= 0 && cin < Cin) ? Wp[(int64_t)row * ldw + cin] : 0.f;
            }
        }
        if (with_bwd) {
            // data-gradient A operands: W[row(T, 4 g + r)][cin = 16 mt + n]
#pragma unroll
            for (int T = 0; T < 3; ++T)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rc_row(T, 4 * g + r, Cout, nullptr);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int cin = 16 * mt + n;
                        wt[(RC_WT_FWD + (mt * 3 + T) * 4 + r) * 64 + lane] = (row >= 0 && cin < Cin) ? Wp[(int64_t)row * ldw + cin] : 0.f;
                    }
                }
        }
    }
}

template <bool T0, bool T1, bool T2>
__device__ __forceinline__ void rc_recompute_lds(const float* wl /* wt + lane (opaque to LICM) */, const float (&x)[3][RC_KS],
                                                 f32x4 (&acc)[3][3]) {
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[T][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < RC_KS; ++s) {
        const float w0 = T0 ? wl[(0 * RC_KS + s) * 64] : 0.f, w1 = T1 ? wl[(1 * RC_KS + s) * 64] : 0.f,
                    w2 = T2 ? wl[(2 * RC_KS + s) * 64] : 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (T0) acc[0][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0, x[a][s], acc[0][a], 0, 0, 0);
            if (T1) acc[1][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, x[a][s], acc[1][a], 0, 0, 0);
            if (T2) acc[2][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2, x[a][s], acc[2][a], 0, 0, 0);
        }
    }
}

template <bool T0, bool T1, bool T2>
__device__ __forceinline__ void rc_recompute(const float (&Aw)[3][RC_KS], const float (&x)[3][RC_KS], f32x4 (&acc)[3][3]) {
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[T][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < RC_KS; ++s)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (T0) acc[0][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[0][s], x[a][s], acc[0][a], 0, 0, 0);
            if (T1) acc[1][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[1][s], x[a][s], acc[1][a], 0, 0, 0);
            if (T2) acc[2][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[2][s], x[a][s], acc[2][a], 0, 0, 0);
        }
}

// (p, d) of slot SIGMA out of / back into the accumulator tiles
template <int SIGMA>
__device__ __forceinline__ void rc_slot_get(const f32x4 (&acc)[3][3], float (&p)[3], float (&d)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (SIGMA < 4) { p[a] = acc[0][a][SIGMA]; d[a] = acc[1][a][SIGMA]; }
        else { p[a] = acc[2][a][2 * (SIGMA - 4)]; d[a] = acc[2][a][2 * (SIGMA - 4) + 1]; }
    }
}
template <int SIGMA>
__device__ __forceinline__ void rc_slot_put(f32x4 (&acc)[3][3], const float (&p)[3], const float (&d)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (SIGMA < 4) { acc[0][a][SIGMA] = p[a]; acc[1][a][SIGMA] = d[a]; }
        else { acc[2][a][2 * (SIGMA - 4)] = p[a]; acc[2][a][2 * (SIGMA - 4) + 1] = d[a]; }
    }
}

// per-channel constants in LDS: [c][8] = mean, rstd, gamma, beta, m1, m2, -, -   (zeros for c >= Cout)
struct RcPrm { float mean, rstd, gamma, beta, m1, m2; };
__device__ __forceinline__ void rc_params_to_lds(float* prm, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const double* __restrict__ red, int training, double count, int Cout) {
    for (int c = threadIdx.x; c < RC_CMAX; c += blockDim.x) {
        const bool ok = c < Cout;
        const float ga = ok ? gamma[c] : 0.f;
        prm[c * 8 + 0] = ok ? mean[c] : 0.f;
        prm[c * 8 + 1] = ok ? rstd[c] : 0.f;
        prm[c * 8 + 2] = ga;
        prm[c * 8 + 3] = ok ? beta[c] : 0.f;
        prm[c * 8 + 4] = (ok && red && training) ? (float)(red[c] / count) * ga : 0.f;
        prm[c * 8 + 5] = (ok && red && training) ? (float)(red[Cout + c] / count) * ga : 0.f;
    }
}
__device__ __forceinline__ RcPrm rc_prm(const float* prm, int c) {
    const float4 u = *reinterpret_cast<const float4*>(prm + c * 8);
    const float2 v = *reinterpret_cast<const float2*>(prm + c * 8 + 4);
    return RcPrm{u.x, u.y, u.z, u.w, v.x, v.y};
}

// Upstream gradient of slot SG (channel 4 g + SG, or 16 + 2 g + (SG - 4)) at position l0 + n.  gs = gout + cloud offset
// (+ l0 when not pooled): uniform.  Pooled layers (MeanBackward): grad / k of the position's point -- a quad of 4 positions
// lies inside one point; goff_a / goff_b = g * 12 Np + pt, g * 6 Np + pt.  Un-pooled: the lane offsets of RcOff.
template <int SG, bool POOL>
__device__ __forceinline__ void rc_load_g(const float* __restrict__ gs, int64_t L, int pool_k, unsigned goff_a, unsigned goff_b,
                                          unsigned goff_safe, bool ok, float (&g)[3]) {
    const int64_t plane = POOL ? L / pool_k : L;
    const float* q = gs + (int64_t)(SG < 4 ? SG : 16 + (SG - 4)) * 3 * plane;             // uniform
    const unsigned o = SG < 4 ? goff_a : (ok ? goff_b : goff_safe);      // an absent channel reads group 0's row, zeroed below
    float v0 = q[o], v1 = (q + plane)[o], v2 = (q + 2 * plane)[o];
    if (POOL) {
        const float kf = (float)pool_k;
        v0 = v0 / kf; v1 = v1 / kf; v2 = v2 / kf;
    }
    g[0] = ok ? v0 : 0.f;
    g[1] = ok ? v1 : 0.f;
    g[2] = ok ? v2 : 0.f;
}

// sum over the 16 lanes of a lane group (same value in all of them afterwards)
__device__ __forceinline__ float rc_row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// A wave's range of sub-tiles: equal chunks (multiples of `unit` sub-tiles) over all waves of the grid
__device__ __forceinline__ void rc_wave_range(int64_t total_st, int unit, int64_t* st0, int64_t* st1) {
    const int64_t waves = (int64_t)gridDim.x * RC_WAVES;
    const int64_t units = (total_st + unit - 1) / unit;
    const int64_t per = (units + waves - 1) / waves;
    // (readfirstlane: the wave index is uniform, so the whole sub-tile walk -- cloud, first position, every base address --
    // stays in scalar registers and the per-lane part of an address is one loop-invariant 32-bit offset)
    const int64_t w = (int64_t)blockIdx.x * RC_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    *st0 = w * per * unit;
    const int64_t e = *st0 + per * unit;
    *st1 = e < total_st ? e : total_st;
}

// Two per-channel sums of a workgroup -> fp64 atomics (one per channel and workgroup); s1 / s2: this lane's slot sums
__device__ __forceinline__ void rc_publish_sums(float (&s1)[RC_SLOTS], float (&s2)[RC_SLOTS], int Cout, int n, int g,
                                                float* part /* [RC_WAVES][2][RC_CMAX] */, double* __restrict__ out) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int sg = 0; sg < RC_SLOTS; ++sg) {
        const float a = rc_row16_sum(s1[sg]), b = rc_row16_sum(s2[sg]);
        const int c = rc_slot_channel(sg, g);
        if (n == 0 && c < RC_CMAX) {
            part[(wave * 2 + 0) * RC_CMAX + c] = a;
            part[(wave * 2 + 1) * RC_CMAX + c] = b;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * Cout; i += blockDim.x) {
        const int which = i / Cout, c = i % Cout;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < RC_WAVES; ++w) t += (double)part[(w * 2 + which) * RC_CMAX + c];
        atomicAdd(out + which * Cout + c, t);
    }
}

// ------------------------------------------------------------------------------------------------ forward statistics
template <int CIN, int COUT>
__global__ __launch_bounds__(RC_TPB, 2) void vn_rc_stats_kernel(const float* __restrict__ X, const float* __restrict__ Wp, int ldw,
                                                               int B, int Cin_rt, int Cout_rt, int64_t L, double* __restrict__ sums,
                                                               BnFin fin) {
    __shared__ float part[RC_WAVES * 2 * RC_CMAX];
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    constexpr int Cin = CIN, Cout = COUT;
    (void)Cin_rt; (void)Cout_rt;
    const RcOff off = rc_offsets(n, g, Cin, L);
    float Aw[3][RC_KS];
    rc_weights_fwd(Wp, ldw, Cout, Cin, n, g, Aw);
    for (int i = threadIdx.x; i < RC_WAVES * 2 * RC_CMAX; i += blockDim.x) part[i] = 0.f;
    float s1[RC_SLOTS], s2[RC_SLOTS];
#pragma unroll
    for (int sg = 0; sg < RC_SLOTS; ++sg) s1[sg] = s2[sg] = 0.f;
    const int64_t per_cloud = L / 16;
    int64_t st0, st1;
    rc_wave_range((int64_t)B * per_cloud, 1, &st0, &st1);
    float x[3][RC_KS], xn[3][RC_KS];
    if (st0 < st1) rc_load_x(X + (st0 / per_cloud) * Cin * 3 * L + (st0 % per_cloud) * 16, Cin, L, g, off, xn);
    for (int64_t st = st0; st < st1; ++st) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < RC_KS; ++s) x[a][s] = xn[a][s];
        if (st + 1 < st1) rc_load_x(X + ((st + 1) / per_cloud) * Cin * 3 * L + ((st + 1) % per_cloud) * 16, Cin, L, g, off, xn);
        f32x4 acc[3][3];
        rc_recompute<true, false, true>(Aw, x, acc);         // the norm statistics need the feature rows only
#pragma unroll
        for (int sg = 0; sg < RC_SLOTS; ++sg) {
            float px, py, pz;
            if (sg < 4) { px = acc[0][0][sg]; py = acc[0][1][sg]; pz = acc[0][2][sg]; }
            else { px = acc[2][0][2 * (sg - 4)]; py = acc[2][1][2 * (sg - 4)]; pz = acc[2][2][2 * (sg - 4)]; }
            const float nn = sqrtf(px * px + py * py + pz * pz) + VN_EPS;
            const bool ok = rc_slot_channel(sg, g) < Cout;
            s1[sg] += ok ? nn : 0.f;
            s2[sg] += ok ? nn * nn : 0.f;
        }
    }
    __syncthreads();
    rc_publish_sums(s1, s2, Cout, n, g, part, sums);
    bn_finalize_tail(fin, sums, Cout);
}

// ------------------------------------------------------------------------------------------------ forward apply (+ mean over k)
// POOL: out [B, Cout, 3, L / pool_k].  A wave walks whole macro-tiles of lcm(16, pool_k) positions; every quad of 4 positions
// belongs to one point (pool_k % 4 == 0), its partial sum -- ((o0 + o1) + o2) + o3, the order of vn_apply_pool_tile_kernel --
// goes to a wave-private LDS patch and the pool_k / 4 partials of a point are added in position order.
template <int CIN, int COUT, bool POOL>
__global__ __launch_bounds__(RC_TPB, 2) void vn_rc_apply_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Wp, int ldw,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   int B, int Cin_rt, int Cout_rt, int64_t L, int pool_k_rt, int macro_st,
                                                                   float slope, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* prm = smem;                                                   // [RC_CMAX][8]
    float* patch = smem + RC_CMAX * 8 + (threadIdx.x >> 6) * (macro_st * 4 * 72);     // POOL: [quads][4 groups][6 slots][3]
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    constexpr int Cin = CIN, Cout = COUT;
    (void)Cin_rt; (void)Cout_rt;
    const int pool_k = POOL ? pool_k_rt : 0;
    const RcOff off = rc_offsets(n, g, Cin, L);
    float Aw[3][RC_KS];
    rc_weights_fwd(Wp, ldw, Cout, Cin, n, g, Aw);
    rc_params_to_lds(prm, mean, rstd, gamma, beta, nullptr, 0, 1.0, Cout);
    __syncthreads();
    const int64_t per_cloud = L / 16;
    int64_t st0, st1;
    rc_wave_range((int64_t)B * per_cloud, POOL ? macro_st : 1, &st0, &st1);
    float x[3][RC_KS], xn[3][RC_KS];
    if (st0 < st1) rc_load_x(X + (st0 / per_cloud) * Cin * 3 * L + (st0 % per_cloud) * 16, Cin, L, g, off, xn);
    for (int64_t st = st0; st < st1; ++st) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < RC_KS; ++s) x[a][s] = xn[a][s];
        if (st + 1 < st1) rc_load_x(X + ((st + 1) / per_cloud) * Cin * 3 * L + ((st + 1) % per_cloud) * 16, Cin, L, g, off, xn);
        const int64_t b = st / per_cloud, l0 = (st % per_cloud) * 16;      // uniform
        f32x4 acc[3][3];
        rc_recompute<true, true, true>(Aw, x, acc);
        const int q_in_macro = (int)((st - st0) % macro_st) * 4 + (n >> 2);
        float* ob = out + (b * Cout * 3) * L + l0;                         // (un-pooled output) uniform base
#define RC_FWD_SLOT(SG)                                                                                                  \
        {                                                                                                                \
            const int c = rc_slot_channel(SG, g);                                                                        \
            const RcPrm pr = rc_prm(prm, c < RC_CMAX ? c : 0);                                                           \
            VnPoint sp;                                                                                                  \
            rc_slot_get<SG>(acc, sp.p, sp.d);                                                                            \
            vn_forward_point(sp, pr.mean, pr.rstd, pr.gamma, pr.beta, RC_FLAGS);                                         \
            float o[3];                                                                                                  \
            vn_output_point(sp, slope, RC_FLAGS, o);                                                                     \
            if (POOL) {                                                                                                  \
                _Pragma("unroll") for (int a = 0; a < 3; ++a) {                                                          \
                    const int v = __float_as_int(o[a]);                                                                  \
                    const float o1 = __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x55, 0xF, 0xF, false));           \
                    const float o2 = __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0xAA, 0xF, 0xF, false));           \
                    const float o3 = __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0xFF, 0xF, 0xF, false));           \
                    const float q4 = ((o[a] + o1) + o2) + o3;                                                            \
                    if ((n & 3) == 0) patch[((q_in_macro * 4 + g) * RC_SLOTS + SG) * 3 + a] = q4;                        \
                }                                                                                                        \
            } else if (c < Cout) {                                                                                       \
                float* op = ob + (int64_t)(SG < 4 ? SG : 16 + (SG - 4)) * 3 * L;          /* uniform */                  \
                const unsigned oo = SG < 4 ? off.a : off.b;                                                              \
                op[oo] = o[0];                                                                                           \
                (op + L)[oo] = o[1];                                                                                     \
                (op + 2 * L)[oo] = o[2];                                                                                 \
            }                                                                                                            \
        }
        RC_FWD_SLOT(0) RC_FWD_SLOT(1) RC_FWD_SLOT(2) RC_FWD_SLOT(3) RC_FWD_SLOT(4) RC_FWD_SLOT(5)
#undef RC_FWD_SLOT
        if (POOL && (st - st0) % macro_st == macro_st - 1) {
            // the macro-tile is complete: pool_k / 4 quad partials per point, added in position order, divided by k like
            // torch's mean
            __builtin_amdgcn_s_waitcnt(0xC07F);                           // lgkmcnt(0): the patch is wave-private
            const int qpp = pool_k / 4, pts = macro_st * 4 / qpp;
            const int64_t Np = L / pool_k;
            const int64_t pt0 = (((st - macro_st + 1) % per_cloud) * 16) / pool_k;
            const float kf = (float)pool_k;
            for (int o = lane; o < pts * 72; o += 64) {
                const int pt = o / 72, rem = o % 72, gg = rem / 18, sg = (rem % 18) / 3, a = rem % 3;
                const int c = rc_slot_channel(sg, gg);
                float t = 0.f;
                for (int q = 0; q < qpp; ++q) t += patch[(((pt * qpp + q) * 4 + gg) * RC_SLOTS + sg) * 3 + a];
                if (c < Cout) out[((b * Cout + c) * 3 + a) * Np + pt0 + pt] = t / kf;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward reduction
template <int CIN, int COUT, bool POOL>
__global__ __launch_bounds__(RC_TPB, 2) void vn_rc_bwd_reduce_kernel(const float* __restrict__ X, const float* __restrict__ Wp, int ldw,
                                                                    const float* __restrict__ gout, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int B, int Cin_rt, int Cout_rt, int64_t L,
                                                                    int pool_k_rt, float slope, double* __restrict__ red_out) {
    __shared__ __attribute__((aligned(16))) float prm[RC_CMAX * 8];
    __shared__ float part[RC_WAVES * 2 * RC_CMAX];
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    constexpr int Cin = CIN, Cout = COUT;
    (void)Cin_rt; (void)Cout_rt;
    const int pool_k = POOL ? pool_k_rt : 0;
    const RcOff off = rc_offsets(n, g, Cin, L);
    float Aw[3][RC_KS];
    rc_weights_fwd(Wp, ldw, Cout, Cin, n, g, Aw);
    rc_params_to_lds(prm, mean, rstd, gamma, beta, nullptr, 0, 1.0, Cout);
    for (int i = threadIdx.x; i < RC_WAVES * 2 * RC_CMAX; i += blockDim.x) part[i] = 0.f;
    __syncthreads();
    float s1[RC_SLOTS], s2[RC_SLOTS];
#pragma unroll
    for (int sg = 0; sg < RC_SLOTS; ++sg) s1[sg] = s2[sg] = 0.f;
    const int64_t per_cloud = L / 16;
    const int64_t g_cloud = pool_k > 0 ? (int64_t)Cout * 3 * (L / pool_k) : (int64_t)Cout * 3 * L;
    int64_t st0, st1;
    rc_wave_range((int64_t)B * per_cloud, 1, &st0, &st1);
    float x[3][RC_KS], xn[3][RC_KS];
    if (st0 < st1) rc_load_x(X + (st0 / per_cloud) * Cin * 3 * L + (st0 % per_cloud) * 16, Cin, L, g, off, xn);
    for (int64_t st = st0; st < st1; ++st) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < RC_KS; ++s) x[a][s] = xn[a][s];
        if (st + 1 < st1) rc_load_x(X + ((st + 1) / per_cloud) * Cin * 3 * L + ((st + 1) % per_cloud) * 16, Cin, L, g, off, xn);
        const int64_t b = st / per_cloud, l0 = (st % per_cloud) * 16;      // uniform
        // invalid channel slots (group 3 has no channel 22 / 23 ...) read a clamped, valid address and are zeroed
        const unsigned pt = POOL ? (unsigned)((l0 / 4 + (n >> 2)) / (pool_k / 4)) : 0u;
        const unsigned Np3 = POOL ? 3u * (unsigned)(L / pool_k) : 0u;
        const unsigned goff_a = POOL ? g * 4u * Np3 + pt : off.a;
        const unsigned goff_b = POOL ? g * 2u * Np3 + pt : off.b;
        const unsigned goff_safe = POOL ? pt : (unsigned)n;
        const float* gs = gout + b * g_cloud + (POOL ? 0 : l0);
        f32x4 acc[3][3];
        rc_recompute<true, true, true>(Aw, x, acc);
#define RC_RED_SLOT(SG)                                                                                                  \
        {                                                                                                                \
            const int c = rc_slot_channel(SG, g);                                                                        \
            const bool ok = c < Cout;                                                                                    \
            const RcPrm pr = rc_prm(prm, c < RC_CMAX ? c : 0);                                                           \
            VnPoint sp;                                                                                                  \
            rc_slot_get<SG>(acc, sp.p, sp.d);                                                                            \
            vn_forward_point(sp, pr.mean, pr.rstd, pr.gamma, pr.beta, RC_FLAGS);                                         \
            float gv[3], gq[3], gd[3];                                                                                   \
            rc_load_g<SG, POOL>(gs, L, pool_k, goff_a, goff_b, goff_safe, ok, gv);                                       \
            vn_lrelu_bwd_point(sp, gv, slope, RC_FLAGS, gq, gd);                                                         \
            const float gnb = (gq[0] * sp.p[0] + gq[1] * sp.p[1] + gq[2] * sp.p[2]) / sp.n;                              \
            s1[SG] += ok ? gnb : 0.f;                                                                                    \
            s2[SG] += ok ? gnb * sp.xhat : 0.f;                                                                          \
        }
        RC_RED_SLOT(0) RC_RED_SLOT(1) RC_RED_SLOT(2) RC_RED_SLOT(3) RC_RED_SLOT(4) RC_RED_SLOT(5)
#undef RC_RED_SLOT
    }
    __syncthreads();
    rc_publish_sums(s1, s2, Cout, n, g, part, red_out);
}

// ------------------------------------------------------------------------------------------------ backward apply + data gradient
template <int CIN, int COUT, bool POOL>
__global__ __launch_bounds__(RC_TPB, 2) void vn_rc_bwd_apply_kernel(const float* __restrict__ X, const float* __restrict__ Wp, int ldw,
                                                                   const float* __restrict__ gout, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const double* __restrict__ red,
                                                                   int training, int B, int Cin_rt, int Cout_rt, int64_t L, int pool_k_rt,
                                                                   float slope, float* __restrict__ GPD, float* __restrict__ gX,
                                                                   float* __restrict__ ggamma, float* __restrict__ gbeta) {
    __shared__ __attribute__((aligned(16))) float prm[RC_CMAX * 8];
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    constexpr int Cin = CIN, Cout = COUT;
    (void)Cin_rt; (void)Cout_rt;
    const int pool_k = POOL ? pool_k_rt : 0;
    const RcOff off = rc_offsets(n, g, Cin, L);
    const int Ctot = 2 * Cout;
    if (ggamma && blockIdx.x == 0)
        for (int c = threadIdx.x; c < Cout; c += blockDim.x) {          // BatchNorm parameter gradients = the finished sums
            ggamma[c] += (float)red[Cout + c];
            gbeta[c] += (float)red[c];
        }
    __shared__ float wt[(RC_WT_FWD + RC_WT_BWD) * 64];
    rc_weights_to_lds(wt, Wp, ldw, Cout, Cin, true);
    rc_params_to_lds(prm, mean, rstd, gamma, beta, red, training, (double)B * (double)L, Cout);
    __syncthreads();
    const int64_t per_cloud = L / 16;
    const int64_t g_cloud = pool_k > 0 ? (int64_t)Cout * 3 * (L / pool_k) : (int64_t)Cout * 3 * L;
    int64_t st0, st1;
    rc_wave_range((int64_t)B * per_cloud, 1, &st0, &st1);
    float x[3][RC_KS], xn[3][RC_KS];
    if (st0 < st1) rc_load_x(X + (st0 / per_cloud) * Cin * 3 * L + (st0 % per_cloud) * 16, Cin, L, g, off, xn);
    for (int64_t st = st0; st < st1; ++st) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < RC_KS; ++s) x[a][s] = xn[a][s];
        const int64_t b = st / per_cloud, l0 = (st % per_cloud) * 16;      // uniform
        // invalid channel slots (group 3 has no channel 22 / 23 ...) read a clamped, valid address and are zeroed
        const unsigned pt = POOL ? (unsigned)((l0 / 4 + (n >> 2)) / (pool_k / 4)) : 0u;
        const unsigned Np3 = POOL ? 3u * (unsigned)(L / pool_k) : 0u;
        const unsigned goff_a = POOL ? g * 4u * Np3 + pt : off.a;
        const unsigned goff_b = POOL ? g * 2u * Np3 + pt : off.b;
        const unsigned goff_safe = POOL ? pt : (unsigned)n;
        const float* gs = gout + b * g_cloud + (POOL ? 0 : l0);
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                                  // opaque: the weight reads stay inside the loop
        const float* wl = wt + lane_o;
        f32x4 acc[3][3];
        rc_recompute_lds<true, true, true>(wl, x, acc);
#define RC_BWD_SLOT(SG)                                                                                                  \
        {                                                                                                                \
            const int c = rc_slot_channel(SG, g);                                                                        \
            const bool ok = c < Cout;                                                                                    \
            const RcPrm pr = rc_prm(prm, c < RC_CMAX ? c : 0);                                                           \
            VnPoint sp;                                                                                                  \
            rc_slot_get<SG>(acc, sp.p, sp.d);                                                                            \
            vn_forward_point(sp, pr.mean, pr.rstd, pr.gamma, pr.beta, RC_FLAGS);                                         \
            float gv[3], gq[3], gd[3], gp[3];                                                                            \
            rc_load_g<SG, POOL>(gs, L, pool_k, goff_a, goff_b, goff_safe, ok, gv);                                       \
            vn_lrelu_bwd_point(sp, gv, slope, RC_FLAGS, gq, gd);                                                         \
            vn_bn_bwd_point(sp, gq, true, training != 0, pr.m1, pr.m2, pr.gamma, pr.rstd, gp);                           \
            if (!ok) { gp[0] = gp[1] = gp[2] = 0.f; gd[0] = gd[1] = gd[2] = 0.f; }                                        \
            rc_slot_put<SG>(acc, gp, gd);                                                                                \
        }
        RC_BWD_SLOT(0) RC_BWD_SLOT(1) RC_BWD_SLOT(2) RC_BWD_SLOT(3) RC_BWD_SLOT(4) RC_BWD_SLOT(5)
#undef RC_BWD_SLOT
        // the next sub-tile's operands are requested HERE, after the point-wise phase (its ~35 live values per slot plus the 36
        // accumulators leave no room for 18 more: requested at the loop head they were spilled to scratch and reloaded), and
        // fly under the 72 data-gradient MFMAs and the stores below
        if (st + 1 < st1) rc_load_x(X + ((st + 1) / per_cloud) * Cin * 3 * L + ((st + 1) % per_cloud) * 16, Cin, L, g, off, xn);
        // the gradient tile: stored once for the weight-gradient GEMM of the side stream ...  (row of (tile, reg): tile 0
        // channel 4 g + r, tile 1 the same channel's direction row, tile 2 channel 16 + 2 g + (r >> 1), feature / direction
        // row by r & 1: a uniform row block + this lane's offset)
        float* Gb = GPD + (b * Ctot * 3) * L + l0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (4 * g + r < Cout) {
                float* o0 = Gb + (int64_t)r * 3 * L;                          // uniform row blocks
                float* o1 = Gb + (int64_t)(Cout + r) * 3 * L;
                o0[off.a] = acc[0][0][r]; (o0 + L)[off.a] = acc[0][1][r]; (o0 + 2 * L)[off.a] = acc[0][2][r];
                o1[off.a] = acc[1][0][r]; (o1 + L)[off.a] = acc[1][1][r]; (o1 + 2 * L)[off.a] = acc[1][2][r];
            }
            if (16 + 2 * g + (r >> 1) < Cout) {
                float* o2 = Gb + (int64_t)(((r & 1) ? Cout : 0) + 16 + (r >> 1)) * 3 * L;
                o2[off.b] = acc[2][0][r]; (o2 + L)[off.b] = acc[2][1][r]; (o2 + 2 * L)[off.b] = acc[2][2][r];
            }
        }
        // ... and consumed from the registers as the B operand of gX = W^T GPD
        f32x4 gx[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int a = 0; a < 3; ++a) gx[mt][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int T = 0; T < 3; ++T)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d0 = wl[(RC_WT_FWD + (0 * 3 + T) * 4 + r) * 64], d1 = wl[(RC_WT_FWD + (1 * 3 + T) * 4 + r) * 64];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    gx[0][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(d0, acc[T][a][r], gx[0][a], 0, 0, 0);
                    gx[1][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1, acc[T][a][r], gx[1][a], 0, 0, 0);
                }
            }
        float* Xg = gX + (b * Cin * 3) * L + l0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * mt + 4 * g + r < Cin) {
                    float* o = Xg + (int64_t)(16 * mt + r) * 3 * L;           // uniform row block
                    o[off.a] = gx[mt][0][r];
                    (o + L)[off.a] = gx[mt][1][r];
                    (o + 2 * L)[off.a] = gx[mt][2][r];
                }
    }
}

int rc_grid(int B, int64_t L, int unit) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n_cu = v;
    }
    const int64_t units = ((int64_t)B * (L / 16) + unit - 1) / unit;
    int64_t blocks = (int64_t)n_cu * 2;                                  // two workgroups per CU (launch bounds)
    const int64_t need = (units + RC_WAVES - 1) / RC_WAVES;
    if (blocks > need) blocks = need;
    return (int)(blocks < 1 ? 1 : blocks);
}

int rc_gcd(int a, int b) { return b == 0 ? a : rc_gcd(b, a % b); }

int rc_dims_ok(int B, int Cin, int Cout, int64_t L, int pool_k) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || L <= 0) return HPCS_ERR_SHAPE;
    // instantiated for the channel counts of every HPCS EdgeConv block (64 // 3 = 21 in, 21 out: vn_dgcnn_partseg.py:28-36);
    // other widths take the streaming kernels
    if (Cin != 21 || Cout != 21 || L % 16 != 0) return HPCS_ERR_UNSUPPORTED;
    if ((int64_t)(2 * Cout > Cin ? 2 * Cout : Cin) * 3 * L >= ((int64_t)1 << 30)) return HPCS_ERR_UNSUPPORTED;    // 32-bit lane offsets
    if (pool_k < 0 || (pool_k > 0 && (pool_k % 4 != 0 || pool_k > 64 || L % pool_k != 0))) return HPCS_ERR_UNSUPPORTED;
    return HPCS_OK;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hpcs_vn_rc_supported(int Cin, int Cout, int Cd, int64_t L, int pool_k, int flags) {
    return flags == RC_FLAGS && Cd == Cout && rc_dims_ok(1, Cin, Cout, L, pool_k) == HPCS_OK ? 1 : 0;
}

extern "C" int hpcs_vn_rc_stats(const float* X, const float* Wp, int ldw, int B, int Cin, int Cout, int64_t L, double* sums,
                                uint32_t* counter, float eps, float momentum, float* running_mean, float* running_var,
                                float* mean, float* rstd, hpcs_stream_t stream) {
    if (!X || !Wp || !sums) return HPCS_ERR_NULL;
    if (counter && (!mean || !rstd)) return HPCS_ERR_NULL;
    if (const int rc = rc_dims_ok(B, Cin, Cout, L, 0)) return rc;
    if (ldw < Cin) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    BnFin f;
    f.counter = counter;
    f.count = (double)B * (double)L;
    f.eps = eps;
    f.momentum = momentum;
    f.rmean = running_mean;
    f.rvar = running_var;
    f.mean = mean;
    f.rstd = rstd;
    hipLaunchKernelGGL((vn_rc_stats_kernel<21, 21>), dim3(rc_grid(B, L, 1)), dim3(RC_TPB), 0, ST, X, Wp, ldw, B, Cin, Cout, L, sums, f);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_rc_apply_fwd(const float* X, const float* Wp, int ldw, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, int B, int Cin, int Cout, int64_t L, int pool_k,
                                    float slope, float* out, hpcs_stream_t stream) {
    if (!X || !Wp || !mean || !rstd || !gamma || !beta || !out) return HPCS_ERR_NULL;
    if (const int rc = rc_dims_ok(B, Cin, Cout, L, pool_k)) return rc;
    if (ldw < Cin) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (pool_k > 0) {
        const int macro_st = pool_k / rc_gcd(16, pool_k);                // lcm(16, k) / 16 sub-tiles
        const size_t lds = sizeof(float) * (RC_CMAX * 8 + (size_t)RC_WAVES * macro_st * 4 * 72);
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)vn_rc_apply_fwd_kernel<21, 21, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return HPCS_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL((vn_rc_apply_fwd_kernel<21, 21, true>), dim3(rc_grid(B, L, macro_st)), dim3(RC_TPB), lds, ST, X, Wp, ldw, mean, rstd,
                           gamma, beta, B, Cin, Cout, L, pool_k, macro_st, slope, out);
    } else {
        const size_t lds = sizeof(float) * (RC_CMAX * 8);
        hipLaunchKernelGGL((vn_rc_apply_fwd_kernel<21, 21, false>), dim3(rc_grid(B, L, 1)), dim3(RC_TPB), lds, ST, X, Wp, ldw, mean, rstd,
                           gamma, beta, B, Cin, Cout, L, 0, 1, slope, out);
    }
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_rc_bwd_reduce(const float* X, const float* Wp, int ldw, const float* gout, const float* mean,
                                     const float* rstd, const float* gamma, const float* beta, int B, int Cin, int Cout,
                                     int64_t L, int pool_k, float slope, double* red, hpcs_stream_t stream) {
    if (!X || !Wp || !gout || !mean || !rstd || !gamma || !beta || !red) return HPCS_ERR_NULL;
    if (const int rc = rc_dims_ok(B, Cin, Cout, L, pool_k)) return rc;
    if (ldw < Cin) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (pool_k > 0)
        hipLaunchKernelGGL((vn_rc_bwd_reduce_kernel<21, 21, true>), dim3(rc_grid(B, L, 1)), dim3(RC_TPB), 0, ST, X, Wp, ldw, gout, mean,
                           rstd, gamma, beta, B, Cin, Cout, L, pool_k, slope, red);
    else
        hipLaunchKernelGGL((vn_rc_bwd_reduce_kernel<21, 21, false>), dim3(rc_grid(B, L, 1)), dim3(RC_TPB), 0, ST, X, Wp, ldw, gout, mean,
                           rstd, gamma, beta, B, Cin, Cout, L, pool_k, slope, red);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_rc_bwd_apply(const float* X, const float* Wp, int ldw, const float* gout, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, const double* red, int training,
                                    int B, int Cin, int Cout, int64_t L, int pool_k, float slope, float* GPD, float* gX,
                                    float* ggamma, float* gbeta, hpcs_stream_t stream) {
    if (!X || !Wp || !gout || !mean || !rstd || !gamma || !beta || !red || !GPD || !gX) return HPCS_ERR_NULL;
    if ((ggamma == nullptr) != (gbeta == nullptr)) return HPCS_ERR_NULL;
    if (const int rc = rc_dims_ok(B, Cin, Cout, L, pool_k)) return rc;
    if (ldw < Cin) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (pool_k > 0)
        hipLaunchKernelGGL((vn_rc_bwd_apply_kernel<21, 21, true>), dim3(rc_grid(B, L, 1)), dim3(RC_TPB), 0, ST, X, Wp, ldw, gout, mean,
                           rstd, gamma, beta, red, training, B, Cin, Cout, L, pool_k, slope, GPD, gX, ggamma, gbeta);
    else
        hipLaunchKernelGGL((vn_rc_bwd_apply_kernel<21, 21, false>), dim3(rc_grid(B, L, 1)), dim3(RC_TPB), 0, ST, X, Wp, ldw, gout, mean,
                           rstd, gamma, beta, red, training, B, Cin, Cout, L, pool_k, slope, GPD, gX, ggamma, gbeta);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
