// a2-a6: the element-wise / reduction side of the Vector-Neuron layers (everything around the channel GEMMs).
//
// Replaces get_graph_feature (vn_dgcnn_util.py:13-41), VNBatchNorm (vn_layers.py:112-132), the VN-LeakyReLU
// of VNLinearLeakyReLU (vn_layers.py:66-77), mean_pool / VNMaxPool (vn_layers.py:135-153), the frame einsum of
// VNStdFeature (vn_layers.py:197) and the mean/max/concat glue of VN_DGCNN_partseg.forward
// (vn_dgcnn_partseg.py:79-95) plus the Conv1d heads' BatchNorm1d+LeakyReLU+Dropout (:40-57).
//
// All kernels are HBM-bound streaming passes over [B,C,3,L] tensors with L contiguous: threads run along L
// (coalesced 4-byte lanes, three xyz planes L apart), one grid.y slice per channel, per-channel BatchNorm
// reductions accumulate in double through one atomic per workgroup.
#include <stdlib.h>

#include "common.h"

namespace {

#include "vn_point.h"

constexpr int TPB = 256;
constexpr int ITEMS = 8;   // elements of the flattened (b,l) axis per thread in reduction kernels

// ------------------------------------------------------------------------------------------------ a2
__global__ void edge_gather_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int C, int N, int k,
                                       float* __restrict__ out) {
    const int64_t L = (int64_t)N * k;
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int n = (int)(l / k);
    const int64_t j = idx[(int64_t)b * L + l];
    const float* xb = x + ((int64_t)b * C + c) * 3 * N;
    float* o_diff = out + (((int64_t)b * 2 * C + c) * 3) * L + l;
    float* o_ctr = out + (((int64_t)b * 2 * C + C + c) * 3) * L + l;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ctr = xb[(int64_t)a * N + n];
        const float nb = xb[(int64_t)a * N + j];
        o_diff[(int64_t)a * L] = nb - ctr;
        o_ctr[(int64_t)a * L] = ctr;
    }
}

// centre part: gx[b,c,a,n] = sum_j (g_ctr - g_diff).  Each thread owns k contiguous floats per plane, so a
// wave reads one contiguous 64*k*4-byte span (float4 when k%4==0).
template <bool VEC>
__global__ void edge_gather_bwd_center_kernel(const float* __restrict__ gout, int C, int N, int k, float* __restrict__ gx) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t L = (int64_t)N * k;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* gd = gout + (((int64_t)b * 2 * C + c) * 3 + a) * L + (int64_t)n * k;
        const float* gc = gout + (((int64_t)b * 2 * C + C + c) * 3 + a) * L + (int64_t)n * k;
        float s = 0.f;
        if (VEC) {
            for (int j = 0; j < k; j += 4) {
                const float4 u = *reinterpret_cast<const float4*>(gc + j);
                const float4 v = *reinterpret_cast<const float4*>(gd + j);
                s += (u.x - v.x) + (u.y - v.y) + (u.z - v.z) + (u.w - v.w);
            }
        } else {
            for (int j = 0; j < k; ++j) s += gc[j] - gd[j];
        }
        gx[(((int64_t)b * C + c) * 3 + a) * N + n] = s;
    }
}

// Tiled centre part for k % 4 == 0, k <= 64: a workgroup owns 64 points of one (cloud, channel); thread t takes the 4
// consecutive edges 4t..4t+3 of that span (fully coalesced 16-byte loads of both halves), the k/4 partial sums of a point
// meet in LDS.  (The one-thread-per-point kernel above strides its loads by k floats: ~2 TB/s.)
__global__ __launch_bounds__(1024) void edge_gather_bwd_center_tile_kernel(const float* __restrict__ gout, int C, int N, int k,
                                                                           float* __restrict__ gx) {
    __shared__ float part[3][1024];
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t L = (int64_t)N * k;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int64_t l = n0 * k + 4 * (int64_t)t;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float sacc = 0.f;
        if (l < L) {
            const float4 v = *reinterpret_cast<const float4*>(gout + (((int64_t)b * 2 * C + c) * 3 + a) * L + l);
            const float4 u = *reinterpret_cast<const float4*>(gout + (((int64_t)b * 2 * C + C + c) * 3 + a) * L + l);
            sacc = (u.x - v.x) + (u.y - v.y) + (u.z - v.z) + (u.w - v.w);
        }
        part[a][t] = sacc;
    }
    __syncthreads();
    if (t < 64 && n0 + t < N) {
        const int q4 = k / 4;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float sacc = 0.f;
            for (int i = 0; i < q4; ++i) sacc += part[a][t * q4 + i];
            gx[(((int64_t)b * C + c) * 3 + a) * N + n0 + t] = sacc;
        }
    }
}

// neighbour part.  A workgroup takes 64 consecutive edges of one cloud and all C*3 difference channels: the tile is
// read coalesced along l, transposed through LDS, and each edge's C*3 values are added with ONE wave instruction
// to a contiguous row of the point-major scratch tmp[b][point][CP] (full-rate atomic shape: MI355X_MICROARCH
// "Global float atomics").  A second pass folds tmp into gx.
constexpr int ET = 64;
__global__ __launch_bounds__(256) void edge_scatter_tile_kernel(const float* __restrict__ gout, const int64_t* __restrict__ idx,
                                                                int C, int Csrc, int N, int k, int CP, float* __restrict__ tmp) {
    extern __shared__ float tile[];            // [ET][C3+1]
    const int C3 = C * 3;
    const int ld = C3 | 1;                     // odd row pitch: the transposing LDS writes hit 32 distinct banks
    const int64_t L = (int64_t)N * k;
    const int b = blockIdx.y;
    const int64_t l0 = (int64_t)blockIdx.x * ET;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool lok = l0 + lane < L;
    for (int ca = w; ca < C3; ca += 4) {
        const int c = ca / 3, a = ca % 3;
        tile[lane * ld + ca] = lok ? gout[(((int64_t)b * Csrc + c) * 3 + a) * L + l0 + lane] : 0.f;
    }
    __syncthreads();
    for (int e = w; e < ET; e += 4) {
        if (l0 + e >= L) break;
        const int64_t j = idx[(int64_t)b * L + l0 + e];
        float* row = tmp + ((int64_t)b * N + j) * CP;
        for (int ca = lane; ca < C3; ca += 64) atomicAdd(row + ca, tile[e * ld + ca]);
    }
}

// accumulate != 0: gx += tmp (gx already holds the centre part); 0: gx = tmp.  Cdst = channels per cloud of gx.
__global__ void edge_scatter_fold_kernel(const float* __restrict__ tmp, int C, int Cdst, int N, int CP, int accumulate,
                                         float* __restrict__ gx) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int b = blockIdx.y;
    const float* row = tmp + ((int64_t)b * N + n) * CP;
    for (int ca = 0; ca < C * 3; ++ca) {
        float* d = gx + ((int64_t)b * Cdst * 3 + ca) * N + n;
        *d = accumulate ? *d + row[ca] : row[ca];
    }
}

// gV[b,c,a,n] = sum_j GPD[b,c,a,n*k+j]  (segment sum over the k neighbours; written into channels [Ctot,2Ctot) of gUV)
template <bool VEC>
__global__ void edge_segsum_kernel(const float* __restrict__ GPD, int Ctot, int N, int k, float* __restrict__ gUV) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t L = (int64_t)N * k;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* g = GPD + (((int64_t)b * Ctot + c) * 3 + a) * L + (int64_t)n * k;
        float s = 0.f;
        if (VEC) {
            for (int j = 0; j < k; j += 4) {
                const float4 u = *reinterpret_cast<const float4*>(g + j);
                s += (u.x + u.y) + (u.z + u.w);
            }
        } else {
            for (int j = 0; j < k; ++j) s += g[j];
        }
        gUV[(((int64_t)b * 2 * Ctot + Ctot + c) * 3 + a) * N + n] = s;
    }
}

// ------------------------------------------------------------------------------------------------ a3
// (VnPoint, vn_forward_point, vn_output_point, vn_lrelu_bwd_point, vn_bn_bwd_point, BnFin and the finalisation tails: vn_point.h)
__device__ __forceinline__ void load3(const float* __restrict__ base, int64_t L, float (&v)[3]) {
    v[0] = base[0];
    v[1] = base[L];
    v[2] = base[2 * L];
}

// Where a VN layer's pre-activations (p, d) come from.
//   idx == nullptr : base = PD [B, Ctot, 3, L], materialised by the channel GEMM.
//   idx != nullptr : EdgeConv "gather" layer.  Its input is the edge feature (x[nbr]-x[ctr], x[ctr]), so
//                    W [x_nbr - x_ctr ; x_ctr] = W_a x_nbr + (W_b - W_a) x_ctr: base = UV [B, 2*Ctot, 3, N] holds the
//                    POINT-level products U = W_a F, V = (W_b - W_a) F (a GEMM k times smaller than the edge GEMM) and
//                    PD[b,ch,:,n*k+j] = U[b,ch,:,idx[b,n,j]] + V[b,ch,:,n] is formed on the fly; neither the
//                    [B,2C,3,N,k] edge tensor (vn_dgcnn_util.py:33-39) nor PD ever reaches HBM.
//   w2 != nullptr  : TWO-input-channel layer (conv1: its input is the coordinate edge feature [B,2,3,L]).  base = that
//                    edge tensor, w2 = the stacked weight [Ctot] rows of pitch ldw, and PD[b,ch,a,l] = fma(w[ch][1], E1, w[ch][0] * E0)
//                    -- the k-ordered fma chain the MFMA GEMM evaluates, bit for bit -- is formed on the fly: the 330 MB PD of
//                    conv1 (42 channels out of 2) is never written nor read (5 passes over it per step).
struct PdRef {
    const float* base;
    const int64_t* idx;
    int Ctot;
    int64_t L;
    int N, k;
    const float* w2 = nullptr;
    int ldw = 0;
};
__host__ __device__ __forceinline__ bool pd_materialised(const PdRef& r) { return r.idx == nullptr && r.w2 == nullptr; }

__device__ __forceinline__ void pd_load3(const PdRef& r, int64_t b, int ch, int64_t l, float (&v)[3]) {
    if (r.w2 != nullptr) {
        const float w0 = r.w2[(int64_t)ch * r.ldw], w1 = r.w2[(int64_t)ch * r.ldw + 1];
        const float* e0 = r.base + ((b * 2 + 0) * 3) * r.L + l;
        const float* e1 = r.base + ((b * 2 + 1) * 3) * r.L + l;
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = __fmaf_rn(w1, e1[(int64_t)a * r.L], __fmul_rn(w0, e0[(int64_t)a * r.L]));
        return;
    }
    if (r.idx == nullptr) {
        load3(r.base + ((b * r.Ctot + ch) * 3) * r.L + l, r.L, v);
        return;
    }
    const int64_t j = r.idx[b * r.L + l];
    const int64_t n = l / r.k;
    const float* U = r.base + ((b * 2 * r.Ctot + ch) * 3) * r.N;
    const float* V = r.base + ((b * 2 * r.Ctot + r.Ctot + ch) * 3) * r.N;
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = U[(int64_t)a * r.N + j] + V[(int64_t)a * r.N + n];
}

__global__ void vn_norm_stats_kernel(PdRef pd, int B, int Ctot, int64_t L, int Cout,
                                     double* __restrict__ sums, BnFin fin) {
    __shared__ double red[16];
    const int c = blockIdx.y;
    const int64_t total = (int64_t)B * L;
    const int64_t e0 = (int64_t)blockIdx.x * (TPB * ITEMS) + threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t e = e0 + (int64_t)it * TPB;
        if (e < total) {
            const int64_t b = e / L, l = e % L;
            float p[3];
            pd_load3(pd, b, c, l, p);
            const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) + VN_EPS;
            s1 += n;
            s2 += n * n;
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(sums + c, r1);
        atomicAdd(sums + Cout + c, r2);
        bn_finalize_tail_channel(fin, sums, Cout, c, gridDim.x);
    }
}

// Vector form for the materialised layout with L % 4 == 0: a workgroup walks VEC_ITEMS * 1024 consecutive positions of one
// (cloud, channel) row with 16-byte loads and no index arithmetic per element (the scalar kernel above pays a 64-bit
// division per element and ran at ~2 TB/s).
constexpr int VEC_ITEMS = 2;
// K2: PD = w2[c] . E (two input channels, PdRef::w2) formed on the fly: `PD` then points at E [B,2,3,L].
__device__ __forceinline__ void k2_vec(const float* __restrict__ E, int64_t b, int a, int64_t L, int64_t l, float w0, float w1,
                                       float (&o)[4]) {
    const float4 e0 = *reinterpret_cast<const float4*>(E + ((b * 2 + 0) * 3 + a) * L + l);
    const float4 e1 = *reinterpret_cast<const float4*>(E + ((b * 2 + 1) * 3 + a) * L + l);
    o[0] = __fmaf_rn(w1, e1.x, __fmul_rn(w0, e0.x));
    o[1] = __fmaf_rn(w1, e1.y, __fmul_rn(w0, e0.y));
    o[2] = __fmaf_rn(w1, e1.z, __fmul_rn(w0, e0.z));
    o[3] = __fmaf_rn(w1, e1.w, __fmul_rn(w0, e0.w));
}
template <bool K2>
__global__ __launch_bounds__(TPB) void vn_norm_stats_vec_kernel(const float* __restrict__ PD, int Ctot, int64_t L, int Cout,
                                                               double* __restrict__ sums, const float* __restrict__ w2, int ldw,
                                                               BnFin fin, int items) {
    __shared__ double red[16];
    const int c = blockIdx.y, b = blockIdx.z;
    const float* base = PD + (((int64_t)b * Ctot + c) * 3) * L;
    const float w0 = K2 ? w2[(int64_t)c * ldw] : 0.f, w1 = K2 ? w2[(int64_t)c * ldw + 1] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    // `items` x 1024 positions per workgroup, chosen by the launcher: the two block reductions, the two fp64 atomics and the
    // fold's ticket are a fixed ~3 us tail per workgroup -- at 2 x 1024 positions (24 KB of loads) it halved the streaming
    // rate of the EdgeConv layers' pass (165 MB in 64-75 us = 2.2-2.6 TB/s where its siblings run at 5)
#pragma unroll 2
    for (int it = 0; it < items; ++it) {
        const int64_t l = ((int64_t)blockIdx.x * items + it) * (TPB * 4) + threadIdx.x * 4;
        if (l < L) {
            float px[4], py[4], pz[4];
            if (K2) {
                k2_vec(PD, b, 0, L, l, w0, w1, px);
                k2_vec(PD, b, 1, L, l, w0, w1, py);
                k2_vec(PD, b, 2, L, l, w0, w1, pz);
            } else {
                const float4 x = *reinterpret_cast<const float4*>(base + l);
                const float4 y = *reinterpret_cast<const float4*>(base + L + l);
                const float4 z = *reinterpret_cast<const float4*>(base + 2 * L + l);
                px[0] = x.x; px[1] = x.y; px[2] = x.z; px[3] = x.w;
                py[0] = y.x; py[1] = y.y; py[2] = y.z; py[3] = y.w;
                pz[0] = z.x; pz[1] = z.y; pz[2] = z.z; pz[3] = z.w;
            }
            const float n0 = sqrtf(px[0] * px[0] + py[0] * py[0] + pz[0] * pz[0]) + VN_EPS;
            const float n1 = sqrtf(px[1] * px[1] + py[1] * py[1] + pz[1] * pz[1]) + VN_EPS;
            const float n2 = sqrtf(px[2] * px[2] + py[2] * py[2] + pz[2] * pz[2]) + VN_EPS;
            const float n3 = sqrtf(px[3] * px[3] + py[3] * py[3] + pz[3] * pz[3]) + VN_EPS;
            s1 += (n0 + n1) + (n2 + n3);
            s2 += (n0 * n0 + n1 * n1) + (n2 * n2 + n3 * n3);
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(sums + c, r1);
        atomicAdd(sums + Cout + c, r2);
        bn_finalize_tail_channel(fin, sums, Cout, c, gridDim.x * gridDim.z);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, int C, double count, float eps, float momentum,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ mean,
                                   float* __restrict__ rstd, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (training) {
        bn_finalize_channel(sums[c], sums[C + c], c, count, eps, momentum, rmean, rvar, mean, rstd);
    } else {
        mean[c] = rmean[c];
        rstd[c] = 1.0f / sqrtf(rvar[c] + eps);
    }
}

__global__ void vn_apply_fwd_kernel(PdRef pd, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, int B, int Cout, int Cd, int64_t L, float slope,
                                    int flags, float* __restrict__ out) {
    const int c = blockIdx.y;
    const int Ctot = Cout + Cd;
    const int cd = Cd == 1 ? Cout : Cout + c;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)B * L) return;
    const int64_t b = e / L, l = e % L;
    VnPoint s;
    pd_load3(pd, b, c, l, s.p);
    if (flags & HPCS_VN_LRELU) pd_load3(pd, b, cd, l, s.d);
    else s.d[0] = s.d[1] = s.d[2] = 0.f;
    const bool bn = flags & HPCS_VN_BN;
    vn_forward_point(s, bn ? mean[c] : 0.f, bn ? rstd[c] : 1.f, bn ? gamma[c] : 1.f, bn ? beta[c] : 0.f, flags);
    float o[3];
    vn_output_point(s, slope, flags, o);
    float* op = out + ((b * Cout + c) * 3) * L + l;
    op[0] = o[0];
    op[L] = o[1];
    op[2 * L] = o[2];
}

// Vector form of vn_apply_fwd_kernel (materialised PD or conv1's two-channel edge feature, L % 4 == 0): four positions per
// thread, 16-byte loads and stores, (b, c) from the grid -- the scalar form divides a 64-bit element index by L per thread.
template <bool K2>
__global__ __launch_bounds__(TPB) void vn_apply_fwd_vec_kernel(const float* __restrict__ PD, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int Cout, int Cd, int L, float slope,
                                                               int flags, float* __restrict__ out, const float* __restrict__ w2,
                                                               int ldw) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int Ctot = Cout + Cd;
    const int cd = Cd == 1 ? Cout : Cout + c;
    const int l = ((int)blockIdx.x * TPB + (int)threadIdx.x) * 4;
    if (l >= L) return;
    const bool bn = flags & HPCS_VN_BN, act = flags & HPCS_VN_LRELU;
    const float mu = bn ? mean[c] : 0.f, rs = bn ? rstd[c] : 1.f, ga = bn ? gamma[c] : 1.f, be = bn ? beta[c] : 0.f;
    const int64_t plane = L;
    float pv[3][4], dv[3][4];
    if (K2) {
        const float wp0 = w2[(int64_t)c * ldw], wp1 = w2[(int64_t)c * ldw + 1];
        const float wd0 = w2[(int64_t)cd * ldw], wd1 = w2[(int64_t)cd * ldw + 1];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            k2_vec(PD, b, a, plane, l, wp0, wp1, pv[a]);
            if (act) k2_vec(PD, b, a, plane, l, wd0, wd1, dv[a]);
        }
    } else {
        const float* pp = PD + (((int64_t)b * Ctot + c) * 3) * plane + l;
        const float* dp = PD + (((int64_t)b * Ctot + cd) * 3) * plane + l;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float4 u = *reinterpret_cast<const float4*>(pp + a * plane);
            pv[a][0] = u.x; pv[a][1] = u.y; pv[a][2] = u.z; pv[a][3] = u.w;
            if (act) {
                const float4 v = *reinterpret_cast<const float4*>(dp + a * plane);
                dv[a][0] = v.x; dv[a][1] = v.y; dv[a][2] = v.z; dv[a][3] = v.w;
            }
        }
    }
    float ov[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        VnPoint s;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s.p[a] = pv[a][q];
            s.d[a] = act ? dv[a][q] : 0.f;
        }
        vn_forward_point(s, mu, rs, ga, be, flags);
        float o[3];
        vn_output_point(s, slope, flags, o);
#pragma unroll
        for (int a = 0; a < 3; ++a) ov[a][q] = o[a];
    }
    float* op = out + (((int64_t)b * Cout + c) * 3) * plane + l;
#pragma unroll
    for (int a = 0; a < 3; ++a) *reinterpret_cast<float4*>(op + a * plane) = make_float4(ov[a][0], ov[a][1], ov[a][2], ov[a][3]);
}

// fused mean_pool over groups of pool_k consecutive l: one thread per (b, n).  Each thread owns pool_k contiguous
// floats per plane, so a wave covers one contiguous span; float4 loads when pool_k % 4 == 0.
template <bool VEC>
__global__ void vn_apply_pool_fwd_kernel(PdRef pd, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int B, int Cout, int Cd, int64_t L, int pool_k,
                                         float slope, int flags, float* __restrict__ out) {
    const int c = blockIdx.y;
    const int Ctot = Cout + Cd;
    const int cd = Cd == 1 ? Cout : Cout + c;
    const int64_t Np = L / pool_k;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)B * Np) return;
    const int64_t b = e / Np, n = e % Np;
    const bool bn = flags & HPCS_VN_BN;
    const bool act = flags & HPCS_VN_LRELU;
    const float mu = bn ? mean[c] : 0.f, rs = bn ? rstd[c] : 1.f, ga = bn ? gamma[c] : 1.f, be = bn ? beta[c] : 0.f;
    const float* PD = pd.base;
    const float* pp = PD + ((b * Ctot + c) * 3) * L + n * pool_k;       // only dereferenced in the materialised mode
    const float* dp = PD + ((b * Ctot + cd) * 3) * L + n * pool_k;
    float acc[3] = {0.f, 0.f, 0.f};
    constexpr int W = VEC ? 4 : 1;
    for (int j = 0; j < pool_k; j += W) {
        float pv[3][4], dv[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (VEC) {
                const float4 u = *reinterpret_cast<const float4*>(pp + a * L + j);
                pv[a][0] = u.x; pv[a][1] = u.y; pv[a][2] = u.z; pv[a][3] = u.w;
                if (act) {
                    const float4 v = *reinterpret_cast<const float4*>(dp + a * L + j);
                    dv[a][0] = v.x; dv[a][1] = v.y; dv[a][2] = v.z; dv[a][3] = v.w;
                }
            } else if (pd_materialised(pd)) {
                pv[a][0] = pp[a * L + j];
                if (act) dv[a][0] = dp[a * L + j];
            }
        }
        if (!VEC && !pd_materialised(pd)) {
            float t3[3];
            pd_load3(pd, b, c, n * pool_k + j, t3);
            pv[0][0] = t3[0]; pv[1][0] = t3[1]; pv[2][0] = t3[2];
            if (act) {
                pd_load3(pd, b, cd, n * pool_k + j, t3);
                dv[0][0] = t3[0]; dv[1][0] = t3[1]; dv[2][0] = t3[2];
            }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) {
            VnPoint s;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s.p[a] = pv[a][q];
                s.d[a] = act ? dv[a][q] : 0.f;
            }
            vn_forward_point(s, mu, rs, ga, be, flags);
            float o[3];
            vn_output_point(s, slope, flags, o);
            acc[0] += o[0];
            acc[1] += o[1];
            acc[2] += o[2];
        }
    }
    float* op = out + ((b * Cout + c) * 3) * Np + n;
    const float kf = (float)pool_k;              // sum / k like torch's mean (a true division: x * (1/k) rounds twice)
    op[0] = acc[0] / kf;
    op[Np] = acc[1] / kf;
    op[2 * Np] = acc[2] / kf;
}

// Tiled form of the fused mean_pool for the materialised layout with pool_k % 4 == 0, pool_k <= 64: a workgroup owns 64
// points of one (cloud, channel); thread t takes the 4 consecutive positions 4t..4t+3 of that span (six fully coalesced
// 16-byte loads), the pool_k/4 partial sums of a point meet in LDS.  (The one-thread-per-point kernel above strides
// its 16-byte loads by pool_k floats and ran at ~1.5 TB/s.)
__global__ __launch_bounds__(1024) void vn_apply_pool_tile_kernel(const float* __restrict__ PD, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int Cout, int Cd, int64_t L,
                                                                  int pool_k, float slope, int flags, float* __restrict__ out) {
    __shared__ float part[3][1024];
    const int c = blockIdx.y, b = blockIdx.z;
    const int Ctot = Cout + Cd;
    const int cd = Cd == 1 ? Cout : Cout + c;
    const int64_t Np = L / pool_k;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int t = threadIdx.x;
    const int64_t l = n0 * pool_k + 4 * (int64_t)t;
    const bool bn = flags & HPCS_VN_BN;
    const bool act = flags & HPCS_VN_LRELU;
    const float mu = bn ? mean[c] : 0.f, rs = bn ? rstd[c] : 1.f, ga = bn ? gamma[c] : 1.f, be = bn ? beta[c] : 0.f;
    float acc[3] = {0.f, 0.f, 0.f};
    if (l < L) {
        const float* pp = PD + (((int64_t)b * Ctot + c) * 3) * L + l;
        const float* dp = PD + (((int64_t)b * Ctot + cd) * 3) * L + l;
        float pv[3][4], dv[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float4 u = *reinterpret_cast<const float4*>(pp + a * L);
            pv[a][0] = u.x; pv[a][1] = u.y; pv[a][2] = u.z; pv[a][3] = u.w;
            if (act) {
                const float4 v = *reinterpret_cast<const float4*>(dp + a * L);
                dv[a][0] = v.x; dv[a][1] = v.y; dv[a][2] = v.z; dv[a][3] = v.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            VnPoint sp;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                sp.p[a] = pv[a][q];
                sp.d[a] = act ? dv[a][q] : 0.f;
            }
            vn_forward_point(sp, mu, rs, ga, be, flags);
            float o[3];
            vn_output_point(sp, slope, flags, o);
            acc[0] += o[0];
            acc[1] += o[1];
            acc[2] += o[2];
        }
    }
    part[0][t] = acc[0];
    part[1][t] = acc[1];
    part[2][t] = acc[2];
    __syncthreads();
    if (t < 64 && n0 + t < Np) {
        const int q4 = pool_k / 4;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < q4; ++i) {
            s0 += part[0][t * q4 + i];
            s1 += part[1][t * q4 + i];
            s2 += part[2][t * q4 + i];
        }
        float* op = out + (((int64_t)b * Cout + c) * 3) * Np + n0 + t;
        const float kf = (float)pool_k;          // sum / k like torch's mean
        op[0] = s0 / kf;
        op[Np] = s1 / kf;
        op[2 * Np] = s2 / kf;
    }
}

__device__ __forceinline__ void load_gout(const float* __restrict__ gout, int64_t b, int Cout, int c, int64_t L, int64_t l,
                                          int pool_k, float (&g)[3]) {
    if (pool_k > 0) {
        const int64_t Np = L / pool_k;
        const float kf = (float)pool_k;          // MeanBackward: grad / k
        const float* gp = gout + ((b * Cout + c) * 3) * Np + l / pool_k;
        g[0] = gp[0] / kf;
        g[1] = gp[Np] / kf;
        g[2] = gp[2 * Np] / kf;
    } else {
        load3(gout + ((b * Cout + c) * 3) * L + l, L, g);
    }
}

__global__ void vn_bwd_reduce_kernel(PdRef pd, const float* __restrict__ gout,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, int B, int Cout,
                                     int Cd, int64_t L, int pool_k, float slope, int flags, double* __restrict__ red_out) {
    __shared__ double red[16];
    const int c = blockIdx.y;
    const int Ctot = Cout + Cd;
    const int cd = Cd == 1 ? Cout : Cout + c;
    const int64_t total = (int64_t)B * L;
    const int64_t e0 = (int64_t)blockIdx.x * (TPB * ITEMS) + threadIdx.x;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t e = e0 + (int64_t)it * TPB;
        if (e < total) {
            const int64_t b = e / L, l = e % L;
            VnPoint s;
            pd_load3(pd, b, c, l, s.p);
            if (flags & HPCS_VN_LRELU) pd_load3(pd, b, cd, l, s.d);
            else s.d[0] = s.d[1] = s.d[2] = 0.f;
            vn_forward_point<true>(s, mu, rs, ga, be, flags);
            float g[3], gq[3], gd[3];
            load_gout(gout, b, Cout, c, L, l, pool_k, g);
            vn_lrelu_bwd_point(s, g, slope, flags, gq, gd);
            const float gnb = vn_div(gq[0] * s.p[0] + gq[1] * s.p[1] + gq[2] * s.p[2], s.n);
            s1 += gnb;
            s2 += gnb * s.xhat;
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(red_out + c, r1);
        atomicAdd(red_out + Cout + c, r2);
    }
}

// Vector form of the backward reduction (materialised layout, L % 4 == 0; pooled layers need pool_k % 4 == 0 so the four
// positions of a lane belong to one point).
template <bool K2>
__global__ __launch_bounds__(TPB) void vn_bwd_reduce_vec_kernel(const float* __restrict__ PD, const float* __restrict__ gout,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int Cout, int Cd, int64_t L, int pool_k, float slope, int flags,
                                                               double* __restrict__ red_out, const float* __restrict__ w2, int ldw) {
    __shared__ double red[16];
    const int c = blockIdx.y, b = blockIdx.z;
    const int Ctot = Cout + Cd;
    const int cd = Cd == 1 ? Cout : Cout + c;
    const bool act = flags & HPCS_VN_LRELU;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    const float wp0 = K2 ? w2[(int64_t)c * ldw] : 0.f, wp1 = K2 ? w2[(int64_t)c * ldw + 1] : 0.f;
    const float wd0 = K2 ? w2[(int64_t)cd * ldw] : 0.f, wd1 = K2 ? w2[(int64_t)cd * ldw + 1] : 0.f;
    const float* pp = PD + (((int64_t)b * Ctot + c) * 3) * L;
    const float* dp = PD + (((int64_t)b * Ctot + cd) * 3) * L;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < VEC_ITEMS; ++it) {
        const int64_t l = ((int64_t)blockIdx.x * VEC_ITEMS + it) * (TPB * 4) + threadIdx.x * 4;
        if (l < L) {
            float pv[3][4], dv[3][4], gv[3][4];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (K2) {
                    k2_vec(PD, b, a, L, l, wp0, wp1, pv[a]);
                    if (act) k2_vec(PD, b, a, L, l, wd0, wd1, dv[a]);
                } else {
                    const float4 u = *reinterpret_cast<const float4*>(pp + a * L + l);
                    pv[a][0] = u.x; pv[a][1] = u.y; pv[a][2] = u.z; pv[a][3] = u.w;
                    if (act) {
                        const float4 v = *reinterpret_cast<const float4*>(dp + a * L + l);
                        dv[a][0] = v.x; dv[a][1] = v.y; dv[a][2] = v.z; dv[a][3] = v.w;
                    }
                }
            }
            if (pool_k > 0) {
                const int64_t Np = L / pool_k;
                const float kf = (float)pool_k;
                const float* gp = gout + (((int64_t)b * Cout + c) * 3) * Np + (int)((unsigned)l / (unsigned)pool_k);   // l < L < 2^31 (host)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float g = gp[a * Np] / kf;
                    gv[a][0] = g; gv[a][1] = g; gv[a][2] = g; gv[a][3] = g;
                }
            } else {
                const float* gp = gout + (((int64_t)b * Cout + c) * 3) * L + l;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float4 g = *reinterpret_cast<const float4*>(gp + a * L);
                    gv[a][0] = g.x; gv[a][1] = g.y; gv[a][2] = g.z; gv[a][3] = g.w;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                VnPoint sp;
                float g[3], gq[3], gd[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    sp.p[a] = pv[a][q];
                    sp.d[a] = act ? dv[a][q] : 0.f;
                    g[a] = gv[a][q];
                }
                vn_forward_point<true>(sp, mu, rs, ga, be, flags);
                vn_lrelu_bwd_point(sp, g, slope, flags, gq, gd);
                const float gnb = vn_div(gq[0] * sp.p[0] + gq[1] * sp.p[1] + gq[2] * sp.p[2], sp.n);
                s1 += gnb;
                s2 += gnb * sp.xhat;
            }
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(red_out + c, r1);
        atomicAdd(red_out + Cout + c, r2);
    }
}

// ---- two-input-channel layers, all output channels per thread (conv1: Cout = Cd = COUT = 21) ---------------------------
// A thread owns 4 consecutive positions: it loads the edge feature once (six 16-byte loads) and walks the COUT channels
// with the weights in scalar registers, so E is read ONCE per pass instead of once per channel (the per-channel K2 kernels
// re-read it 21 times through L2: 61 us for the statistics against 42 us for reading a materialised PD).
template <int COUT>
__device__ __forceinline__ void k2_block_reduce(float (&s1)[COUT], float (&s2)[COUT], double* __restrict__ out, int Cout) {
    __shared__ float part[2 * COUT][TPB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        const float a = hpcs_wave_sum(s1[c]), b = hpcs_wave_sum(s2[c]);
        if (lane == 0) { part[c][wv] = a; part[COUT + c][wv] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * COUT) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) t += (double)part[threadIdx.x][w];
        const int c = threadIdx.x < COUT ? threadIdx.x : threadIdx.x - COUT;
        atomicAdd(out + (threadIdx.x < COUT ? c : Cout + c), t);
    }
}

template <int COUT>
__global__ __launch_bounds__(TPB) void vn_norm_stats_k2all_kernel(const float* __restrict__ E, const float* __restrict__ w2, int ldw,
                                                                 int64_t L, double* __restrict__ sums, BnFin fin) {
    const int b = blockIdx.y;
    const int64_t l = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 4;
    float s1[COUT], s2[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) s1[c] = s2[c] = 0.f;
    if (l < L) {
        float4 e0[3], e1[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            e0[a] = *reinterpret_cast<const float4*>(E + (((int64_t)b * 2 + 0) * 3 + a) * L + l);
            e1[a] = *reinterpret_cast<const float4*>(E + (((int64_t)b * 2 + 1) * 3 + a) * L + l);
        }
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            const float w0 = w2[(int64_t)c * ldw], w1 = w2[(int64_t)c * ldw + 1];
            float p[3][4];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                p[a][0] = __fmaf_rn(w1, e1[a].x, __fmul_rn(w0, e0[a].x));
                p[a][1] = __fmaf_rn(w1, e1[a].y, __fmul_rn(w0, e0[a].y));
                p[a][2] = __fmaf_rn(w1, e1[a].z, __fmul_rn(w0, e0[a].z));
                p[a][3] = __fmaf_rn(w1, e1[a].w, __fmul_rn(w0, e0[a].w));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float n = sqrtf(p[0][q] * p[0][q] + p[1][q] * p[1][q] + p[2][q] * p[2][q]) + VN_EPS;
                s1[c] += n;
                s2[c] += n * n;
            }
        }
    }
    k2_block_reduce<COUT>(s1, s2, sums, COUT);
    bn_finalize_tail(fin, sums, COUT);
}

// The backward reduction walks the channels in a ROLLED loop (its body -- forward point, leaky-ReLU and norm backward of four
// positions -- unrolled 21 times spilled: 370 us) with the next channel's gradient plane prefetched under the arithmetic of the
// current one (without the prefetch the loop is 21 serial HBM round trips: 290 us at N=2048); a channel's two sums are reduced
// across the wave at once, so the block keeps 2*COUT*4 floats of LDS.
template <int COUT>
__global__ __launch_bounds__(TPB) void vn_bwd_reduce_k2all_kernel(const float* __restrict__ E, const float* __restrict__ w2, int ldw,
                                                                 const float* __restrict__ gout, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int64_t L, float slope, int flags,
                                                                 double* __restrict__ red_out) {
    __shared__ float part[2 * COUT][TPB / 64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t l0 = ((int64_t)blockIdx.x * TPB + tid) * 4;
    const bool live = l0 < L;
    const int64_t l = live ? l0 : 0;                       // dead threads read position 0 and contribute nothing
    float e0v[3][4], e1v[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 u = *reinterpret_cast<const float4*>(E + (((int64_t)b * 2 + 0) * 3 + a) * L + l);
        const float4 v = *reinterpret_cast<const float4*>(E + (((int64_t)b * 2 + 1) * 3 + a) * L + l);
        e0v[a][0] = u.x; e0v[a][1] = u.y; e0v[a][2] = u.z; e0v[a][3] = u.w;
        e1v[a][0] = v.x; e1v[a][1] = v.y; e1v[a][2] = v.z; e1v[a][3] = v.w;
    }
    const float* gp = gout + ((int64_t)b * COUT * 3) * L + l;
    float4 gn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) gn[a] = *reinterpret_cast<const float4*>(gp + a * L);
#pragma unroll 1
    for (int c = 0; c < COUT; ++c) {
        float gv[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) { gv[a][0] = gn[a].x; gv[a][1] = gn[a].y; gv[a][2] = gn[a].z; gv[a][3] = gn[a].w; }
        const int cn = c + 1 < COUT ? c + 1 : c;
#pragma unroll
        for (int a = 0; a < 3; ++a) gn[a] = *reinterpret_cast<const float4*>(gp + ((int64_t)cn * 3 + a) * L);
        const float wp0 = w2[(int64_t)c * ldw], wp1 = w2[(int64_t)c * ldw + 1];
        const float wd0 = w2[(int64_t)(COUT + c) * ldw], wd1 = w2[(int64_t)(COUT + c) * ldw + 1];
        const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            VnPoint sp;
            float g[3], gq[3], gd[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                sp.p[a] = __fmaf_rn(wp1, e1v[a][q], __fmul_rn(wp0, e0v[a][q]));
                sp.d[a] = __fmaf_rn(wd1, e1v[a][q], __fmul_rn(wd0, e0v[a][q]));
                g[a] = gv[a][q];
            }
            vn_forward_point<true>(sp, mu, rs, ga, be, flags);
            vn_lrelu_bwd_point(sp, g, slope, flags, gq, gd);
            const float gnb = vn_div(gq[0] * sp.p[0] + gq[1] * sp.p[1] + gq[2] * sp.p[2], sp.n);
            s1 += gnb;
            s2 += gnb * sp.xhat;
        }
        s1 = hpcs_wave_sum(live ? s1 : 0.f);
        s2 = hpcs_wave_sum(live ? s2 : 0.f);
        if (lane == 0) { part[c][wv] = s1; part[COUT + c][wv] = s2; }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) t += (double)part[tid][w];
        atomicAdd(red_out + tid, t);                       // [0,COUT): sum gnb, [COUT,2COUT): sum gnb*xhat
    }
}

template <bool SHARED>
__global__ void vn_bwd_apply_kernel(PdRef pd, const float* __restrict__ gout,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const double* __restrict__ red, int training, int B, int Cout, int Cd, int64_t L,
                                    int pool_k, float slope, int flags, float* __restrict__ GPD, int c_chunk,
                                    float* __restrict__ ggamma, float* __restrict__ gbeta) {
    const int Ctot = Cout + Cd;
    if (ggamma && blockIdx.x == 0 && threadIdx.x == 0) {
        // BatchNorm parameter gradients are the finished reductions themselves: red = [sum_beta(C) | sum_gamma(C)].  One
        // writer per channel adds them into the parameters' gradient slots (no separate launches for them).
        const int c0_ = SHARED ? blockIdx.y * c_chunk : blockIdx.y;
        const int c1_ = SHARED ? min(Cout, c0_ + c_chunk) : blockIdx.y + 1;
        for (int c = c0_; c < c1_; ++c) {
            ggamma[c] += (float)red[Cout + c];
            gbeta[c] += (float)red[c];
        }
    }
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)B * L) return;
    const int64_t b = e / L, l = e % L;
    const double count = (double)B * (double)L;
    // SHARED (one direction channel for all features): blockIdx.y owns a chunk of channels and adds its part of the
    // direction gradient atomically into the (pre-zeroed) direction rows
    const int c_begin = SHARED ? blockIdx.y * c_chunk : blockIdx.y;
    const int c_end = SHARED ? min(Cout, c_begin + c_chunk) : blockIdx.y + 1;
    const bool bn = flags & HPCS_VN_BN;
    float gdsum[3] = {0.f, 0.f, 0.f};
    float dsh[3] = {0.f, 0.f, 0.f};
    if (SHARED && (flags & HPCS_VN_LRELU)) pd_load3(pd, b, Cout, l, dsh);
    for (int c = c_begin; c < c_end; ++c) {
        VnPoint s;
        pd_load3(pd, b, c, l, s.p);
        if (SHARED) {
            s.d[0] = dsh[0]; s.d[1] = dsh[1]; s.d[2] = dsh[2];
        } else if (flags & HPCS_VN_LRELU) {
            pd_load3(pd, b, Cout + c, l, s.d);
        } else {
            s.d[0] = s.d[1] = s.d[2] = 0.f;
        }
        const float mu = bn ? mean[c] : 0.f, rs = bn ? rstd[c] : 1.f, ga = bn ? gamma[c] : 1.f, be = bn ? beta[c] : 0.f;
        vn_forward_point<true>(s, mu, rs, ga, be, flags);
        float g[3], gq[3], gd[3];
        load_gout(gout, b, Cout, c, L, l, pool_k, g);
        vn_lrelu_bwd_point(s, g, slope, flags, gq, gd);
        float gp[3];
        {
            const float m1 = (bn && training) ? (float)(red[c] / count) * ga : 0.f;
            const float m2 = (bn && training) ? (float)(red[Cout + c] / count) * ga : 0.f;
            vn_bn_bwd_point(s, gq, bn, training != 0, m1, m2, ga, rs, gp);
        }
        float* o = GPD + ((b * Ctot + c) * 3) * L + l;
        o[0] = gp[0];
        o[L] = gp[1];
        o[2 * L] = gp[2];
        if (SHARED) {
            gdsum[0] += gd[0]; gdsum[1] += gd[1]; gdsum[2] += gd[2];
        } else if (Cd > 0) {
            float* od = GPD + ((b * Ctot + Cout + c) * 3) * L + l;
            od[0] = gd[0];
            od[L] = gd[1];
            od[2 * L] = gd[2];
        }
    }
    if (SHARED) {
        float* od = GPD + ((b * Ctot + Cout) * 3) * L + l;
        atomicAdd(od, gdsum[0]);
        atomicAdd(od + L, gdsum[1]);
        atomicAdd(od + 2 * L, gdsum[2]);
    }
}

// Per-channel-direction apply (EdgeConv layers, vn1, vn2) on a materialised PD -- or conv1's two-channel edge feature with
// the products formed on the fly (K2) -- with L % 4 == 0: four positions per thread, 16-byte loads and stores, (b, c) from the
// grid.  The scalar kernel above spent most of its ~690 VALU instructions per position on index arithmetic (five integer
// divisions, two of them 64-bit) and a double-precision division, and moved 4 bytes per lane per access: 184 us per
// EdgeConv layer at B=32, N=1024, k=20 (4.5 TB/s of the 825 MB it has to move).
template <bool K2>
__global__ __launch_bounds__(TPB) void vn_bwd_apply_vec_kernel(
    const float* __restrict__ PD, const float* __restrict__ gout, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ red, int training, int B, int Cout,
    int Cd, int L, int pool_k, float slope, int flags, float* __restrict__ GPD, float* __restrict__ ggamma,
    float* __restrict__ gbeta, const float* __restrict__ w2, int ldw) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int Ctot = Cout + Cd;
    if (ggamma && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
        ggamma[c] += (float)red[Cout + c];
        gbeta[c] += (float)red[c];
    }
    const int l = ((int)blockIdx.x * TPB + (int)threadIdx.x) * 4;
    if (l >= L) return;
    const bool bn = flags & HPCS_VN_BN, act = flags & HPCS_VN_LRELU;
    const float mu = bn ? mean[c] : 0.f, rs = bn ? rstd[c] : 1.f, ga = bn ? gamma[c] : 1.f, be = bn ? beta[c] : 0.f;
    const double count = (double)B * (double)L;
    const float m1 = (bn && training) ? (float)(red[c] / count) * ga : 0.f;
    const float m2 = (bn && training) ? (float)(red[Cout + c] / count) * ga : 0.f;
    const int64_t plane = L;
    float pv[3][4], dv[3][4], gv[3][4];
    if (K2) {
        const float wp0 = w2[(int64_t)c * ldw], wp1 = w2[(int64_t)c * ldw + 1];
        const float wd0 = w2[(int64_t)(Cout + c) * ldw], wd1 = w2[(int64_t)(Cout + c) * ldw + 1];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            k2_vec(PD, b, a, plane, l, wp0, wp1, pv[a]);
            if (act) k2_vec(PD, b, a, plane, l, wd0, wd1, dv[a]);
        }
    } else {
        const float* pp = PD + (((int64_t)b * Ctot + c) * 3) * plane + l;
        const float* dp = PD + (((int64_t)b * Ctot + Cout + c) * 3) * plane + l;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float4 u = *reinterpret_cast<const float4*>(pp + a * plane);
            pv[a][0] = u.x; pv[a][1] = u.y; pv[a][2] = u.z; pv[a][3] = u.w;
            if (act) {
                const float4 v = *reinterpret_cast<const float4*>(dp + a * plane);
                dv[a][0] = v.x; dv[a][1] = v.y; dv[a][2] = v.z; dv[a][3] = v.w;
            }
        }
    }
    if (pool_k > 0) {                              // MeanBackward: the four positions belong to one point (pool_k % 4 == 0)
        const int Np = L / pool_k;
        const float kf = (float)pool_k;
        const float* gp = gout + (((int64_t)b * Cout + c) * 3) * Np + l / pool_k;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g = gp[(int64_t)a * Np] / kf;
            gv[a][0] = g; gv[a][1] = g; gv[a][2] = g; gv[a][3] = g;
        }
    } else {
        const float* gp = gout + (((int64_t)b * Cout + c) * 3) * plane + l;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float4 g = *reinterpret_cast<const float4*>(gp + a * plane);
            gv[a][0] = g.x; gv[a][1] = g.y; gv[a][2] = g.z; gv[a][3] = g.w;
        }
    }
    float op[3][4], od[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        VnPoint s;
        float g[3], gq[3], gd[3], gp[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s.p[a] = pv[a][q];
            s.d[a] = act ? dv[a][q] : 0.f;
            g[a] = gv[a][q];
        }
        vn_forward_point<true>(s, mu, rs, ga, be, flags);
        vn_lrelu_bwd_point(s, g, slope, flags, gq, gd);
        vn_bn_bwd_point(s, gq, bn, training != 0, m1, m2, ga, rs, gp);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            op[a][q] = gp[a];
            od[a][q] = gd[a];
        }
    }
    float* o = GPD + (((int64_t)b * Ctot + c) * 3) * plane + l;
#pragma unroll
    for (int a = 0; a < 3; ++a) *reinterpret_cast<float4*>(o + a * plane) = make_float4(op[a][0], op[a][1], op[a][2], op[a][3]);
    if (Cd > 0) {
        float* o2 = GPD + (((int64_t)b * Ctot + Cout + c) * 3) * plane + l;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            *reinterpret_cast<float4*>(o2 + a * plane) = make_float4(od[a][0], od[a][1], od[a][2], od[a][3]);
    }
}

// Shared-direction apply (conv6: 341 feature channels, one direction channel) on a materialised PD with L % 4 == 0: four
// positions per thread (16-byte loads and stores), the next channel's p and g planes in flight under the arithmetic of the
// current one.  Same point-wise code as vn_bwd_apply_kernel<true> (which took 689 us at B=32, N=1024: one 4-byte load per
// lane and 22 dependent round trips per thread).
__global__ __launch_bounds__(TPB) void vn_bwd_apply_shared_vec_kernel(
    const float* __restrict__ PD, const float* __restrict__ gout, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ red, int training, int B, int Cout,
    int64_t L, float slope, int flags, float* __restrict__ GPD, int c_chunk, float* __restrict__ ggamma, float* __restrict__ gbeta) {
    const int Ctot = Cout + 1;
    const int b = blockIdx.y;
    const int c_begin = blockIdx.z * c_chunk, c_end = min(Cout, c_begin + c_chunk);
    if (ggamma && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
        for (int c = c_begin; c < c_end; ++c) {
            ggamma[c] += (float)red[Cout + c];
            gbeta[c] += (float)red[c];
        }
    }
    const int64_t l = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 4;
    const bool dead = l >= L;                                  // (stays for the wave-wide transposed atomics at the end)
    float gdsum[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) gdsum[a][q] = 0.f;
    if (!dead) {
    const double count = (double)B * (double)L;
    const bool bn = flags & HPCS_VN_BN;
    const float* Pb = PD + ((int64_t)b * Ctot * 3) * L + l;
    const float* Gb = gout + ((int64_t)b * Cout * 3) * L + l;
    float* Ob = GPD + ((int64_t)b * Ctot * 3) * L + l;
    float dsh[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 v = (flags & HPCS_VN_LRELU) ? *reinterpret_cast<const float4*>(Pb + ((int64_t)Cout * 3 + a) * L) : make_float4(0.f, 0.f, 0.f, 0.f);
        dsh[a][0] = v.x; dsh[a][1] = v.y; dsh[a][2] = v.z; dsh[a][3] = v.w;
    }
    float4 pn[3], gn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        pn[a] = *reinterpret_cast<const float4*>(Pb + ((int64_t)c_begin * 3 + a) * L);
        gn[a] = *reinterpret_cast<const float4*>(Gb + ((int64_t)c_begin * 3 + a) * L);
    }
#pragma unroll 1
    for (int c = c_begin; c < c_end; ++c) {
        float pv[3][4], gv[3][4];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            pv[a][0] = pn[a].x; pv[a][1] = pn[a].y; pv[a][2] = pn[a].z; pv[a][3] = pn[a].w;
            gv[a][0] = gn[a].x; gv[a][1] = gn[a].y; gv[a][2] = gn[a].z; gv[a][3] = gn[a].w;
        }
        const int cn = c + 1 < c_end ? c + 1 : c;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            pn[a] = *reinterpret_cast<const float4*>(Pb + ((int64_t)cn * 3 + a) * L);
            gn[a] = *reinterpret_cast<const float4*>(Gb + ((int64_t)cn * 3 + a) * L);
        }
        const float mu = bn ? mean[c] : 0.f, rs = bn ? rstd[c] : 1.f, ga = bn ? gamma[c] : 1.f, be = bn ? beta[c] : 0.f;
        const float m1 = (bn && training) ? (float)(red[c] / count) * ga : 0.f;
        const float m2 = (bn && training) ? (float)(red[Cout + c] / count) * ga : 0.f;
        float o[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            VnPoint sp;
            float g[3], gq[3], gd[3], gp[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { sp.p[a] = pv[a][q]; sp.d[a] = dsh[a][q]; g[a] = gv[a][q]; }
            vn_forward_point<true>(sp, mu, rs, ga, be, flags);
            vn_lrelu_bwd_point(sp, g, slope, flags, gq, gd);
            vn_bn_bwd_point(sp, gq, bn, training != 0, m1, m2, ga, rs, gp);
#pragma unroll
            for (int a = 0; a < 3; ++a) { o[a][q] = gp[a]; gdsum[a][q] += gd[a]; }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            *reinterpret_cast<float4*>(Ob + ((int64_t)c * 3 + a) * L) = make_float4(o[a][0], o[a][1], o[a][2], o[a][3]);
    }
    }   // !dead
    // The shared direction's gradient is summed over the channel chunks with float atomics.  A lane owns four CONSECUTIVE
    // positions, so adding them straight from the registers is four instructions of 4-byte pieces 16 bytes apart; transposed
    // through a wave-private LDS patch every instruction covers 256 contiguous bytes instead (the shape the atomic units
    // process at full rate -- the same finding as in the sampled HypHC kernel, DESIGN.md section 8 "Round 3").
    __shared__ float tp[TPB / 64][3][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) tp[wv][a][lane * 4 + q] = gdsum[a][q];
    __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0); the patch is wave-private
    const int64_t l_wave = l - (int64_t)lane * 4;             // first position of this wave
    float* Od = GPD + ((int64_t)b * Ctot * 3) * L + (int64_t)Cout * 3 * L + l_wave;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pos = q * 64 + lane;
            if (l_wave + pos < L) atomicAdd(Od + (int64_t)a * L + pos, tp[wv][a][pos]);
        }
}

// ------------------------------------------------------------------------------ Conv1d heads: BN1d + act + mask
__global__ void bn_stats_kernel(const float* __restrict__ Y, int B, int C, int64_t L, double* __restrict__ sums, BnFin fin) {
    __shared__ double red[16];
    const int c = blockIdx.y;
    const int64_t total = (int64_t)B * L;
    const int64_t e0 = (int64_t)blockIdx.x * (TPB * ITEMS) + threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t e = e0 + (int64_t)it * TPB;
        if (e < total) {
            const float y = Y[((e / L) * C + c) * L + e % L];
            s1 += y;
            s2 += y * y;
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(sums + c, r1);
        atomicAdd(sums + C + c, r2);
        bn_finalize_tail_channel(fin, sums, C, c, gridDim.x);
    }
}

__global__ void bn_act_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ mask, int B, int C, int64_t L,
                                  float slope, float* __restrict__ out, unsigned long long dseed,
                                  const unsigned long long* __restrict__ dseed_dev, unsigned dthr, float dinv) {
    if (dseed_dev) dseed = *dseed_dev;
    const int c = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)B * L) return;
    const int64_t off = ((e / L) * C + c) * L + e % L;
    const float z = (Y[off] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    float o = z > 0.f ? z : z * slope;
    if (mask) o *= mask[off];
    else if (dthr) o *= hpcs_dropout_mult(dseed, off, dthr, dinv);
    out[off] = o;
}

__global__ void bn_act_bwd_reduce_kernel(const float* __restrict__ Y, const float* __restrict__ gout,
                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const float* __restrict__ mask, int B, int C, int64_t L, float slope,
                                         double* __restrict__ red_out, unsigned long long dseed,
                                         const unsigned long long* __restrict__ dseed_dev, unsigned dthr, float dinv) {
    __shared__ double red[16];
    if (dseed_dev) dseed = *dseed_dev;
    const int c = blockIdx.y;
    const int64_t total = (int64_t)B * L;
    const int64_t e0 = (int64_t)blockIdx.x * (TPB * ITEMS) + threadIdx.x;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t e = e0 + (int64_t)it * TPB;
        if (e < total) {
            const int64_t off = ((e / L) * C + c) * L + e % L;
            const float xh = (Y[off] - mu) * rs;
            const float z = xh * ga + be;
            float g = gout[off];
            if (mask) g *= mask[off];
            else if (dthr) g *= hpcs_dropout_mult(dseed, off, dthr, dinv);
            g *= z > 0.f ? 1.f : slope;
            s1 += g;
            s2 += g * xh;
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(red_out + c, r1);
        atomicAdd(red_out + C + c, r2);
    }
}

__global__ void bn_act_bwd_apply_kernel(const float* __restrict__ Y, const float* __restrict__ gout,
                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ mask, const double* __restrict__ red, int training,
                                        int B, int C, int64_t L, float slope, float* __restrict__ gY,
                                        unsigned long long dseed, const unsigned long long* __restrict__ dseed_dev,
                                        unsigned dthr, float dinv, float* __restrict__ ggamma, float* __restrict__ gbeta) {
    if (dseed_dev) dseed = *dseed_dev;
    const int c = blockIdx.y;
    if (ggamma && blockIdx.x == 0 && threadIdx.x == 0) {       // red = [sum_beta(C) | sum_gamma(C)]: the parameter gradients
        ggamma[c] += (float)red[C + c];
        gbeta[c] += (float)red[c];
    }
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)B * L) return;
    const double count = (double)B * (double)L;
    const int64_t off = ((e / L) * C + c) * L + e % L;
    const float rs = rstd[c], ga = gamma[c];
    const float xh = (Y[off] - mean[c]) * rs;
    const float z = xh * ga + beta[c];
    float g = gout[off];
    if (mask) g *= mask[off];
    else if (dthr) g *= hpcs_dropout_mult(dseed, off, dthr, dinv);
    g *= z > 0.f ? 1.f : slope;
    float r;
    if (training) {
        const float m1 = (float)(red[c] / count), m2 = (float)(red[C + c] / count);
        r = rs * ga * (g - m1 - xh * m2);
    } else {
        r = rs * ga * g;
    }
    gY[off] = r;
}

// ---- 16-byte forms of the four Conv1d-head kernels (L % 4 == 0, aligned bases).  The scalar forms above pay two 64-bit
// divisions per element and, in the reductions, a ~3 us block-sum + atomics tail per 2048 elements: 33 MB passes ran at
// 1.3-1.9 TB/s (r03).  Here blockIdx.y = channel, blockIdx.z = cloud (elementwise) or a group of `bpw` clouds (reductions,
// sized by the launcher so that ~2000 workgroups share the tensor), a thread owns four consecutive positions.
__device__ __forceinline__ void bn_drop4(float (&m)[4], const float* __restrict__ mask, int64_t off, unsigned long long dseed,
                                         unsigned dthr, float dinv) {
    if (mask) {
        const float4 q = *reinterpret_cast<const float4*>(mask + off);
        m[0] = q.x; m[1] = q.y; m[2] = q.z; m[3] = q.w;
    } else if (dthr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = hpcs_dropout_mult(dseed, off + i, dthr, dinv);
    } else {
        m[0] = m[1] = m[2] = m[3] = 1.f;
    }
}
__global__ __launch_bounds__(TPB) void bn_stats_vec_kernel(const float* __restrict__ Y, int B, int C, int64_t L, int bpw,
                                                          double* __restrict__ sums, BnFin fin) {
    __shared__ double red[16];
    const int c = blockIdx.y;
    const int b0 = blockIdx.x * bpw, b1 = min(B, b0 + bpw);
    float s1 = 0.f, s2 = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float* row = Y + ((int64_t)b * C + c) * L;
        for (int64_t l = (int64_t)threadIdx.x * 4; l < L; l += TPB * 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + l);
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(sums + c, r1);
        atomicAdd(sums + C + c, r2);
        bn_finalize_tail_channel(fin, sums, C, c, gridDim.x);
    }
}
__global__ __launch_bounds__(TPB) void bn_act_fwd_vec_kernel(
    const float* __restrict__ Y, const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mask, int C, int64_t L, float slope, float* __restrict__ out,
    unsigned long long dseed, const unsigned long long* __restrict__ dseed_dev, unsigned dthr, float dinv) {
    if (dseed_dev) dseed = *dseed_dev;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t l = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 4;
    if (l >= L) return;
    const int64_t off = ((int64_t)b * C + c) * L + l;
    const float mu = mean[c], sc = rstd[c] * gamma[c], be = beta[c], rs = rstd[c], ga = gamma[c];
    (void)sc;
    const float4 y = *reinterpret_cast<const float4*>(Y + off);
    float m[4];
    bn_drop4(m, mask, off, dseed, dthr, dinv);
    const float yv[4] = {y.x, y.y, y.z, y.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float z = (yv[i] - mu) * rs * ga + be;           // the scalar kernel's expression, same rounding
        o[i] = (z > 0.f ? z : z * slope);
        if (mask || dthr) o[i] *= m[i];
    }
    *reinterpret_cast<float4*>(out + off) = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(TPB) void bn_act_bwd_reduce_vec_kernel(
    const float* __restrict__ Y, const float* __restrict__ gout, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask, int B, int C, int64_t L,
    int bpw, float slope, double* __restrict__ red_out, unsigned long long dseed, const unsigned long long* __restrict__ dseed_dev,
    unsigned dthr, float dinv) {
    __shared__ double red[16];
    if (dseed_dev) dseed = *dseed_dev;
    const int c = blockIdx.y;
    const int b0 = blockIdx.x * bpw, b1 = min(B, b0 + bpw);
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    float s1 = 0.f, s2 = 0.f;
    for (int b = b0; b < b1; ++b) {
        const int64_t rowoff = ((int64_t)b * C + c) * L;
        for (int64_t l = (int64_t)threadIdx.x * 4; l < L; l += TPB * 4) {
            const int64_t off = rowoff + l;
            const float4 y = *reinterpret_cast<const float4*>(Y + off);
            const float4 gq = *reinterpret_cast<const float4*>(gout + off);
            float m[4];
            bn_drop4(m, mask, off, dseed, dthr, dinv);
            const float yv[4] = {y.x, y.y, y.z, y.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (yv[i] - mu) * rs;
                const float z = xh * ga + be;
                float g = gv[i];
                if (mask || dthr) g *= m[i];
                g *= z > 0.f ? 1.f : slope;
                s1 += g;
                s2 += g * xh;
            }
        }
    }
    const double r1 = hpcs_block_sum<double>((double)s1, red);
    const double r2 = hpcs_block_sum<double>((double)s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(red_out + c, r1);
        atomicAdd(red_out + C + c, r2);
    }
}
__global__ __launch_bounds__(TPB) void bn_act_bwd_apply_vec_kernel(
    const float* __restrict__ Y, const float* __restrict__ gout, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask, const double* __restrict__ red,
    int training, int B, int C, int64_t L, float slope, float* __restrict__ gY, unsigned long long dseed,
    const unsigned long long* __restrict__ dseed_dev, unsigned dthr, float dinv, float* __restrict__ ggamma,
    float* __restrict__ gbeta) {
    if (dseed_dev) dseed = *dseed_dev;
    const int c = blockIdx.y, b = blockIdx.z;
    if (ggamma && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
        ggamma[c] += (float)red[C + c];
        gbeta[c] += (float)red[c];
    }
    const int64_t l = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 4;
    if (l >= L) return;
    const double count = (double)B * (double)L;
    const int64_t off = ((int64_t)b * C + c) * L + l;
    const float rs = rstd[c], ga = gamma[c], mu = mean[c], be = beta[c];
    const float m1 = training ? (float)(red[c] / count) : 0.f, m2 = training ? (float)(red[C + c] / count) : 0.f;
    const float4 y = *reinterpret_cast<const float4*>(Y + off);
    const float4 gq = *reinterpret_cast<const float4*>(gout + off);
    float m[4];
    bn_drop4(m, mask, off, dseed, dthr, dinv);
    const float yv[4] = {y.x, y.y, y.z, y.w}, gv[4] = {gq.x, gq.y, gq.z, gq.w};
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xh = (yv[i] - mu) * rs;
        const float z = xh * ga + be;
        float g = gv[i];
        if (mask || dthr) g *= m[i];
        g *= z > 0.f ? 1.f : slope;
        r[i] = training ? rs * ga * (g - m1 - xh * m2) : rs * ga * g;
    }
    *reinterpret_cast<float4*>(gY + off) = make_float4(r[0], r[1], r[2], r[3]);
}
// clouds per workgroup of the two reductions: ~2000 workgroups over the C x B rows
static inline int bn_bpw(int B, int C) {
    int bpw = (int)(((int64_t)B * C + 1999) / 2000);
    if (bpw < 1) bpw = 1;
    if (bpw > B) bpw = B;
    return bpw;
}
static inline bool bn_vec_ok(int B, int64_t L, const void* a, const void* b, const void* c, const void* d) {
    return L % 4 == 0 && B <= 65535 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15u) == 0);
}

// the multiplier tensor the bn_act kernels apply implicitly for (seed, p): element i = 1/(1-p) or 0 (tests, debugging)
__global__ void dropout_mask_kernel(unsigned long long seed, unsigned thr, float inv, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = thr ? hpcs_dropout_mult(seed, i, thr, inv) : 1.f;
}

// ------------------------------------------------------------------------------ glue
// one workgroup per (b,c,a) row
__global__ void concat_mean_fwd_kernel(const float* __restrict__ x, int C, int N, float* __restrict__ out) {
    __shared__ float red[16];
    __shared__ float mval;
    const int64_t row = blockIdx.x;            // over B*C*3
    const int64_t b = row / (3 * C), r = row % (3 * C);
    const float* xr = x + row * N;
    float* o1 = out + (b * 2 * C * 3 + r) * N;
    float* o2 = out + (b * 2 * C * 3 + 3 * C + r) * N;
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float v = xr[n];
        o1[n] = v;
        s += v;
    }
    const float t = hpcs_block_sum<float>(s, red);
    if (threadIdx.x == 0) mval = t / (float)N;
    __syncthreads();
    const float m = mval;
    for (int n = threadIdx.x; n < N; n += blockDim.x) o2[n] = m;
}

__global__ void concat_mean_bwd_kernel(const float* __restrict__ gout, int C, int N, float* __restrict__ gx) {
    __shared__ float red[16];
    __shared__ float mval;
    const int64_t row = blockIdx.x;
    const int64_t b = row / (3 * C), r = row % (3 * C);
    const float* g1 = gout + (b * 2 * C * 3 + r) * N;
    const float* g2 = gout + (b * 2 * C * 3 + 3 * C + r) * N;
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) s += g2[n];
    const float t = hpcs_block_sum<float>(s, red);
    if (threadIdx.x == 0) mval = t / (float)N;
    __syncthreads();
    const float m = mval;
    for (int n = threadIdx.x; n < N; n += blockDim.x) gx[row * N + n] = g1[n] + m;
}

__global__ void frame_project_fwd_kernel(const float* __restrict__ x, const float* __restrict__ z, int C, int N,
                                         float* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    const int i = blockIdx.y, b = blockIdx.z;
    const float* xp = x + ((int64_t)(b * C + i) * 3) * N + m;
    const float* zp = z + ((int64_t)b * 9) * N + m;
    const float x0 = xp[0], x1 = xp[N], x2 = xp[2 * (int64_t)N];
    float* op = out + ((int64_t)(b * C + i) * 3) * N + m;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        op[(int64_t)k * N] = x0 * zp[(int64_t)(0 * 3 + k) * N] + x1 * zp[(int64_t)(1 * 3 + k) * N] + x2 * zp[(int64_t)(2 * 3 + k) * N];
}

// one thread per (b, m): loops channels, writes gx, accumulates gz
__global__ void frame_project_bwd_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                         const float* __restrict__ gout, int C, int N, float* __restrict__ gx,
                                         float* __restrict__ gz) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    const int b = blockIdx.y;
    float zz[9], gzz[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        zz[t] = z[((int64_t)b * 9 + t) * N + m];
        gzz[t] = 0.f;
    }
    for (int i = 0; i < C; ++i) {
        const int64_t base = ((int64_t)(b * C + i) * 3) * N + m;
        float xv[3], gv[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            xv[a] = x[base + (int64_t)a * N];
            gv[a] = gout[base + (int64_t)a * N];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gx[base + (int64_t)j * N] = gv[0] * zz[j * 3 + 0] + gv[1] * zz[j * 3 + 1] + gv[2] * zz[j * 3 + 2];
#pragma unroll
            for (int k = 0; k < 3; ++k) gzz[j * 3 + k] += xv[j] * gv[k];
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) gz[((int64_t)b * 9 + t) * N + m] = gzz[t];
}

// Fused invariant projection + max over N (vn_layers.py:197 einsum followed by vn_dgcnn_partseg.py:86-87's max):
// out[b, i, k] = max_n sum_j xi[b,i,j,(n)] z[b,j,k,n], where channel i < C reads x [B,C,3,N] and channel i >= C reads
// the per-cloud constant xc [B,Cc,3] (the mean half of cat(x, mean_N(x))).  The [B,C+Cc,3,N] projected tensor is never
// written; argmax (first index on ties) is kept for the sparse backward.  One workgroup per (b, i).
__global__ void project_max_fwd_kernel(const float* __restrict__ x, const float* __restrict__ xc, const float* __restrict__ z,
                                       int C, int Cc, int N, float* __restrict__ out, int32_t* __restrict__ arg) {
    __shared__ float sv[3][4];
    __shared__ int si[3][4];
    const int i = blockIdx.x, b = blockIdx.y;
    const float* zp = z + (int64_t)b * 9 * N;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    const float* xp = nullptr;
    if (i < C) {
        xp = x + ((int64_t)(b * C + i) * 3) * N;
    } else {
        const float* q = xc + (int64_t)(b * Cc + (i - C)) * 3;
        c0 = q[0]; c1 = q[1]; c2 = q[2];
    }
    float bv[3] = {-INFINITY, -INFINITY, -INFINITY};
    int bi[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float x0 = xp ? xp[n] : c0, x1 = xp ? xp[N + n] : c1, x2 = xp ? xp[2 * (int64_t)N + n] : c2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = x0 * zp[(int64_t)(0 * 3 + k) * N + n] + x1 * zp[(int64_t)(1 * 3 + k) * N + n] +
                            x2 * zp[(int64_t)(2 * 3 + k) * N + n];
            if (v > bv[k]) { bv[k] = v; bi[k] = n; }
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = bv[k];
        int ix = bi[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(ix, o, 64);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        if (lane == 0) { sv[k][w] = v; si[k][w] = ix; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float v = sv[k][0];
        int ix = si[k][0];
        for (int j = 1; j < nw; ++j)
            if (sv[k][j] > v || (sv[k][j] == v && si[k][j] < ix)) { v = sv[k][j]; ix = si[k][j]; }
        const int64_t o = ((int64_t)b * (C + Cc) + i) * 3 + k;
        out[o] = v;
        arg[o] = ix;
    }
}

// sparse backward: thread per (b, i); gx / gz must be zero-filled (or, for gx, hold the gradient to add to) by the
// caller (both via atomics); a null gx, or a null gz (then gxc is skipped too), leaves that part out
__global__ void project_max_bwd_kernel(const float* __restrict__ x, const float* __restrict__ xc, const float* __restrict__ z,
                                       const int32_t* __restrict__ arg, const float* __restrict__ gout, int B, int C, int Cc,
                                       int N, float* __restrict__ gx, float* __restrict__ gxc, float* __restrict__ gz) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ct = C + Cc;
    if (t >= (int64_t)B * Ct) return;
    const int b = (int)(t / Ct), i = (int)(t % Ct);
    const float* zp = z + (int64_t)b * 9 * N;
    float gc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int n = arg[t * 3 + k];
        const float g = gout[t * 3 + k];
        float xv[3];
        if (i < C) {
            const float* xp = x + ((int64_t)(b * C + i) * 3) * N;
            xv[0] = xp[n]; xv[1] = xp[N + n]; xv[2] = xp[2 * (int64_t)N + n];
        } else {
            const float* q = xc + (int64_t)(b * Cc + (i - C)) * 3;
            xv[0] = q[0]; xv[1] = q[1]; xv[2] = q[2];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float zjk = zp[(int64_t)(j * 3 + k) * N + n];
            if (i < C) {
                if (gx) atomicAdd(gx + ((int64_t)(b * C + i) * 3 + j) * N + n, g * zjk);   // the 3 k's may share n
            } else {
                gc[j] += g * zjk;
            }
            if (gz) atomicAdd(gz + ((int64_t)b * 9 + j * 3 + k) * N + n, g * xv[j]);
        }
    }
    if (i >= C && gz) {
        float* q = gxc + (int64_t)(b * Cc + (i - C)) * 3;
        q[0] = gc[0]; q[1] = gc[1]; q[2] = gc[2];
    }
}

__global__ void rowmax_fwd_kernel(const float* __restrict__ x, int64_t L, float* __restrict__ out, int32_t* __restrict__ arg) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int64_t row = blockIdx.x;
    const float* xr = x + row * L;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t l = threadIdx.x; l < L; l += blockDim.x) {
        const float v = xr[l];
        if (v > bv || (v == bv && (int)l < bi)) { bv = v; bi = (int)l; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { sv[w] = bv; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i)
            if (sv[i] > bv || (sv[i] == bv && si[i] < bi)) { bv = sv[i]; bi = si[i]; }
        out[row] = bv;
        if (arg) arg[row] = bi;
    }
}

__global__ void rowmax_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ arg, int64_t rows, int64_t L,
                                  float* __restrict__ gx) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * L) return;
    const int64_t row = e / L, l = e % L;
    gx[e] = (int)l == arg[row] ? gout[row] : 0.f;
}

__global__ void rowsum_kernel(const float* __restrict__ x, int64_t L, float scale, float* __restrict__ out) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const float* xr = x + row * L;
    float s = 0.f;
    for (int64_t l = threadIdx.x; l < L; l += blockDim.x) s += xr[l];
    const float t = hpcs_block_sum<float>(s, red);
    if (threadIdx.x == 0) out[row] = t * scale;
}

// many medium rows (256 <= L, L % 4 == 0): one WAVE per row, 16-byte loads, no workgroup barrier
__global__ __launch_bounds__(256) void rowsum_wave_kernel(const float* __restrict__ x, int64_t rows, int64_t L, float scale,
                                                         float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * L;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t q = lane; q < L / 4; q += 64) {
        const float4 v = *reinterpret_cast<const float4*>(xr + 4 * q);
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    const float t = hpcs_wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) out[row] = t * scale;
}

// a few very long rows (e.g. the mean of 32768 per-row losses): 1024 threads, 16-byte loads
__global__ __launch_bounds__(1024) void rowsum_long_kernel(const float* __restrict__ x, int64_t L, float scale,
                                                          float* __restrict__ out) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const float* xr = x + row * L;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int64_t L4 = L / 4;
    for (int64_t q = threadIdx.x; q < L4; q += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(xr + 4 * q);
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
    float s = (s0 + s1) + (s2 + s3);
    for (int64_t l = 4 * L4 + threadIdx.x; l < L; l += 1024) s += xr[l];
    const float t = hpcs_block_sum<float>(s, red);
    if (threadIdx.x == 0) out[row] = t * scale;
}

__global__ void vn_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ d, int C, int N, int k,
                                      float* __restrict__ out, int32_t* __restrict__ arg) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t L = (int64_t)N * k;
    const int64_t base = ((int64_t)(b * C + c) * 3) * L + (int64_t)n * k;
    float bv = -INFINITY;
    int bj = 0;
    for (int j = 0; j < k; ++j) {
        const float dot = x[base + j] * d[base + j] + x[base + L + j] * d[base + L + j] + x[base + 2 * L + j] * d[base + 2 * L + j];
        if (dot > bv) { bv = dot; bj = j; }
    }
    const int64_t ob = ((int64_t)(b * C + c) * 3) * N + n;
    out[ob] = x[base + bj];
    out[ob + N] = x[base + L + bj];
    out[ob + 2 * (int64_t)N] = x[base + 2 * L + bj];
    arg[(int64_t)(b * C + c) * N + n] = bj;
}

__global__ void vn_maxpool_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ arg, int C, int N, int k,
                                      float* __restrict__ gx) {
    const int64_t L = (int64_t)N * k;
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const int n = (int)(l / k), j = (int)(l % k);
    const bool hit = arg[(int64_t)(b * C + c) * N + n] == j;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        gx[((int64_t)(b * C + c) * 3 + a) * L + l] = hit ? gout[((int64_t)(b * C + c) * 3 + a) * N + n] : 0.f;
}

__global__ void scale_inplace_kernel(float* __restrict__ x, int64_t n, float s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}

inline bool bad_dims(int B, int C, int64_t L) { return B < 0 || C <= 0 || L <= 0; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hpcs_version(void) { return 100; }

extern "C" int hpcs_edge_gather_fwd(const float* x, const int64_t* idx, int B, int C, int N, int k, float* out,
                                    hpcs_stream_t stream) {
    if (!x || !idx || !out) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || k <= 0 || C > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    const int64_t L = (int64_t)N * k;
    hipLaunchKernelGGL(edge_gather_fwd_kernel, dim3(hpcs_cdiv(L, TPB), C, B), dim3(TPB), 0, ST, x, idx, C, N, k, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

namespace {

// ---- backward of get_graph_feature through the INVERSE graph (N <= 1024, N*k <= 20480: the BASELINE geometry) ------------
// The scatter-add over idx is a gather over the transposed adjacency: gx[m] = sum over the edges e = (n, j) with
// idx[n, j] = m of g_diff[e], plus the centre part sum_j (g_ctr[m, j] - g_diff[m, j]).  One workgroup per cloud builds the
// transposed lists once per call (LDS counting sort, every list sorted by edge id so that the sums have a fixed order);
// the reduction kernel then stages one N*k plane of `gout` in LDS per step (the next plane is already in flight in
// registers) and every point adds up its own incoming edges: no atomics, no scratch memset, no fold pass, every byte of
// `gout` read once.  Replaces centre pass + memset + atomic scatter + fold (57 + 10 + 160-200 + 15 us per 330 MB layer).
constexpr int INV_T = 1024;
constexpr int INV_MAXL = 20480;
constexpr int INV_HUB = 64;                          // in-degree above which a point is reduced by a whole wave
constexpr int INV_MAXHUB = INV_MAXL / (INV_HUB + 1) + 1;
// A feature-space kNN graph can have HUBS (early in training thousands of points share a nearest neighbour): a
// thread-per-point loop over a 5000-long list held the whole workgroup for 540 us.  Points with more than INV_HUB incoming
// edges are therefore listed separately and summed by one wave each (lanes stride the list, shuffle tree: still a fixed
// order).
struct InvGraph {                                    // per-call scratch inside the caller's workspace
    int* in_ptr;                                     // [B][N+1]
    unsigned short* in_edge;                         // [B][L]  edge ids grouped by target, ascending inside a group
    int* hub_cnt;                                    // [B]
    unsigned short* hub_ids;                         // [B][INV_MAXHUB]
    short* hub_slot;                                 // [B][N]  -1 or the point's position in hub_ids
    size_t bytes;
};
constexpr int INV_MAXSLAB = 4;                       // slabs per cloud (N*k <= 81920)
InvGraph inv_graph_layout(void* ws, int B, int N, int64_t L) {
    InvGraph g;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 255) & ~(size_t)255; return (char*)ws + r; };
    g.in_ptr = (int*)take((size_t)B * (N + 1) * sizeof(int));
    g.in_edge = (unsigned short*)take((size_t)B * L * sizeof(unsigned short));
    g.hub_cnt = (int*)take((size_t)B * sizeof(int));
    g.hub_ids = (unsigned short*)take((size_t)B * INV_MAXHUB * sizeof(unsigned short));
    g.hub_slot = (short*)take((size_t)B * N * sizeof(short));
    g.bytes = o;
    return g;
}

// One workgroup per (cloud, slab).  A slab is a run of N/S source points = Ls = N*k/S consecutive edges (S = 1 for
// N*k <= INV_MAXL; the N = 2048 / k = 40 geometry uses S = 4 slabs of 512 points): every slab gets its own transposed
// graph over ALL N targets, with slab-local edge ids, so that the reduction can stage one slab of a plane in LDS at a time.
// NT = targets per thread (N <= NT * 1024).
template <int NT>
__global__ __launch_bounds__(INV_T) void knn_inverse_kernel(const int64_t* __restrict__ idx, int N, int Ls, int S,
                                                            int* __restrict__ in_ptr, unsigned short* __restrict__ in_edge,
                                                            int* __restrict__ hub_cnt, unsigned short* __restrict__ hub_ids,
                                                            short* __restrict__ hub_slot) {
    // Counting sort of the edges by target with every list ascending in the edge id (= a fixed summation order later):
    //   short lists: slots handed out by LDS atomics, then a per-thread insertion sort of <= INV_HUB entries;
    //   hub lists (thousands of entries): placed STABLY, 1024 edges at a time -- an edge's slot is the list cursor + the
    //   same-hub edges of lower waves (per-wave counters) + the same-hub lanes below it in its wave (ballot/popcount).
    __shared__ int cur[NT * INV_T];                  // in-degree histogram, then the running list cursors
    __shared__ int wsum[INV_T / 64];
    __shared__ unsigned short ine[INV_MAXL];
    __shared__ unsigned short hubs[INV_MAXHUB];
    __shared__ short slot_of[NT * INV_T];            // hub slot of a point or -1
    __shared__ unsigned char wcount[INV_T / 64][INV_MAXHUB];
    const int bs = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int L = Ls;
    const int64_t* ib = idx + (int64_t)bs * Ls;      // clouds and slabs are consecutive runs of the edge array
    (void)S;
#pragma unroll
    for (int j = 0; j < NT; ++j) cur[tid + j * INV_T] = 0;
    for (int i = tid; i < (INV_T / 64) * INV_MAXHUB; i += INV_T) (&wcount[0][0])[i] = 0;
    constexpr int CH = INV_MAXL / INV_T;             // edges per thread: all targets are fetched once, up front
    short tgt[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = c * INV_T + tid;
        const int m = e < L ? (int)ib[e] : -1;
        tgt[c] = (unsigned)m < (unsigned)N ? (short)m : (short)-1;    // kNN indices are cloud-local; anything else is dropped
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH; ++c)
        if (tgt[c] >= 0) atomicAdd(&cur[tgt[c]], 1);
    __syncthreads();
    // exclusive scans over the N counts (target t = tid + j*1024, ascending in j): list starts and hub slots
    int cnt[NT], start[NT], slot[NT], is_hub[NT];
    int run = 0, hrun = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = tid + j * INV_T;
        cnt[j] = t < N ? cur[t] : 0;
        is_hub[j] = cnt[j] > INV_HUB ? 1 : 0;
        int incl = cnt[j], hincl = is_hub[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64), uh = __shfl_up(hincl, o, 64);
            if (lane >= o) { incl += u; hincl += uh; }
        }
        __syncthreads();                             // wsum free again
        if (lane == 63) wsum[wv] = incl | (hincl << 20);               // counts < 2^20, hubs < 2^11
        __syncthreads();
        int base = 0, hbase = 0, tot = 0, htot = 0;
        for (int w = 0; w < INV_T / 64; ++w) {
            const int v = wsum[w];
            if (w < wv) { base += v & 0xfffff; hbase += v >> 20; }
            tot += v & 0xfffff;
            htot += v >> 20;
        }
        start[j] = run + base + incl - cnt[j];
        slot[j] = hrun + hbase + hincl - is_hub[j];
        run += tot;
        hrun += htot;
    }
    const int nh = hrun;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = tid + j * INV_T;
        cur[t] = start[j];
        slot_of[t] = is_hub[j] ? (short)slot[j] : (short)-1;
        if (is_hub[j]) hubs[slot[j]] = (unsigned short)t;
        if (t < N) {
            in_ptr[(int64_t)bs * (N + 1) + t] = start[j];
            if (t == N - 1) in_ptr[(int64_t)bs * (N + 1) + N] = start[j] + cnt[j];
            hub_slot[(int64_t)bs * N + t] = is_hub[j] ? (short)slot[j] : (short)-1;
        }
    }
    __syncthreads();
    // short lists
    short hs[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        hs[c] = tgt[c] >= 0 ? slot_of[tgt[c]] : (short)-1;
        if (tgt[c] >= 0 && hs[c] < 0) ine[atomicAdd(&cur[tgt[c]], 1)] = (unsigned short)(c * INV_T + tid);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j)
        if (tid + j * INV_T < N && !is_hub[j])
            for (int i = start[j] + 1; i < start[j] + cnt[j]; ++i) {
                const unsigned short v = ine[i];
                int q = i - 1;
                while (q >= start[j] && ine[q] > v) { ine[q + 1] = ine[q]; --q; }
                ine[q + 1] = v;
            }
    // hub lists (nh is uniform)
    if (nh > 0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c * INV_T >= L) break;               // uniform
            const int h = hs[c];
            int rank = 0, tot = 0;
            unsigned long long todo = __ballot(h >= 0);
            while (todo) {                           // one round per distinct hub among the wave's 64 edges
                const int lead = __ffsll((long long)todo) - 1;
                const int hl = __builtin_amdgcn_readlane(h, lead);
                const unsigned long long mask = __ballot(h == hl);
                if (h == hl) {
                    rank = __popcll(mask & ((1ull << lane) - 1ull));
                    tot = __popcll(mask);
                }
                todo &= ~mask;
            }
            const bool leader = h >= 0 && rank == 0;
            if (leader) wcount[wv][h] = (unsigned char)tot;           // tot <= 64
            __syncthreads();
            if (h >= 0) {
                int off = 0;
                for (int w = 0; w < wv; ++w) off += wcount[w][h];
                ine[cur[tgt[c]] + off + rank] = (unsigned short)(c * INV_T + tid);
            }
            __syncthreads();
            if (leader) {
                atomicAdd(&cur[tgt[c]], tot);
                wcount[wv][h] = 0;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int e = tid; e < L; e += INV_T) in_edge[(int64_t)bs * L + e] = ine[e];
    for (int h = tid; h < nh; h += INV_T) hub_ids[(int64_t)bs * INV_MAXHUB + h] = hubs[h];
    if (tid == 0) hub_cnt[bs] = nh;
}

// LDS image shared by the two reduction kernels: one plane of N*k floats, the transposed edge list and the hub tables.
struct InvLds {
    float* plane;                // [L]
    unsigned short* ine;         // [L]
    float* hub_g;                // [INV_MAXHUB]
    int* hub_e0;                 // [INV_MAXHUB]
    int* hub_e1;                 // [INV_MAXHUB]
};
constexpr size_t INV_LDS_BYTES = (size_t)INV_MAXL * 6 + (size_t)INV_MAXHUB * 12 + 64;
__device__ __forceinline__ InvLds inv_lds_layout(float4* raw, int L) {
    InvLds l;
    l.plane = reinterpret_cast<float*>(raw);
    l.hub_g = l.plane + L;
    l.hub_e0 = reinterpret_cast<int*>(l.hub_g + INV_MAXHUB);
    l.hub_e1 = l.hub_e0 + INV_MAXHUB;
    l.ine = reinterpret_cast<unsigned short*>(l.hub_e1 + INV_MAXHUB);
    return l;
}
struct InvOwn { int e0, e1, hs, nh; };
__device__ __forceinline__ InvOwn inv_load_graph(const InvLds& l, const int* in_ptr, const unsigned short* in_edge,
                                                 const int* hub_cnt, const short* hub_slot, int b, int N, int L) {
    const int tid = threadIdx.x;
    for (int e = tid; e < L; e += INV_T) l.ine[e] = in_edge[(int64_t)b * L + e];
    InvOwn o{0, 0, -1, hub_cnt[b]};
    if (tid < N) {
        o.e0 = in_ptr[(int64_t)b * (N + 1) + tid];
        o.e1 = in_ptr[(int64_t)b * (N + 1) + tid + 1];
        o.hs = hub_slot[(int64_t)b * N + tid];
        if (o.hs >= 0) { l.hub_e0[o.hs] = o.e0; l.hub_e1[o.hs] = o.e1; }
    }
    return o;
}
// sum of the staged plane over the incoming edges of this thread's point; hubs are summed by whole waves into hub_g and
// picked up after the barrier this function contains
__device__ __forceinline__ float inv_gather(const InvLds& l, const InvOwn& o) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int h = wv; h < o.nh; h += INV_T / 64) {
        float acc = 0.f;
        for (int e = l.hub_e0[h] + lane; e < l.hub_e1[h]; e += 64) acc += l.plane[l.ine[e]];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
        if (lane == 0) l.hub_g[h] = acc;
    }
    float g = 0.f;
    if (o.hs < 0)
        for (int e = o.e0; e < o.e1; ++e) g += l.plane[l.ine[e]];
    if (o.nh > 0) {                                  // uniform across the workgroup
        __syncthreads();
        if (o.hs >= 0) g = l.hub_g[o.hs];
    }
    return g;
}
__device__ __forceinline__ float inv_own_sum(const float* plane, int N, int k) {       // sum_j plane[tid*k + j], k % 4 == 0
    float s = 0.f;
    if ((int)threadIdx.x < N) {
        const float4* r = reinterpret_cast<const float4*>(plane + threadIdx.x * k);
        for (int j = 0; j < (k >> 2); ++j) {
            const float4 v = r[j];
            s += (v.x + v.y) + (v.z + v.w);
        }
    }
    return s;
}
constexpr int INV_PF = INV_MAXL / 4 / INV_T;         // float4 prefetch registers per thread (5)

// grid (groups, B): the workgroup walks planes p = group, group + groups, ... of cloud b (p = c*3 + a).
__global__ __launch_bounds__(INV_T) void edge_gather_bwd_inv_kernel(const float* __restrict__ gout, const int* __restrict__ in_ptr,
                                                                    const unsigned short* __restrict__ in_edge,
                                                                    const int* __restrict__ hub_cnt, const short* __restrict__ hub_slot,
                                                                    int C, int N, int k, float* __restrict__ gx) {
    extern __shared__ float4 inv_lds[];
    const int L = N * k, L4 = L >> 2;                // host guarantees L % 4 == 0
    const InvLds l = inv_lds_layout(inv_lds, L);
    const int b = blockIdx.y, tid = threadIdx.x;
    const InvOwn own = inv_load_graph(l, in_ptr, in_edge, hub_cnt, hub_slot, b, N, L);
    float4 pre[INV_PF];
    auto fetch = [&](int p, int half) {              // half 0: difference channels [0,C), 1: centre channels [C,2C)
        const int c = p / 3, a = p % 3;
        const float4* src = reinterpret_cast<const float4*>(gout + (((int64_t)b * 2 * C + half * C + c) * 3 + a) * L);
#pragma unroll
        for (int i = 0; i < INV_PF; ++i) {
            const int q = tid + i * INV_T;
            pre[i] = q < L4 ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < INV_PF; ++i) {
            const int q = tid + i * INV_T;
            if (q < L4) reinterpret_cast<float4*>(l.plane)[q] = pre[i];
        }
    };
    const int planes = C * 3;
    int p = blockIdx.x;
    if (p < planes) fetch(p, 1);
    for (; p < planes; p += gridDim.x) {
        __syncthreads();                             // previous plane fully consumed (and the graph tables are in place)
        stash();                                     // centre plane
        fetch(p, 0);
        __syncthreads();
        float s = inv_own_sum(l.plane, N, k);
        __syncthreads();
        stash();                                     // difference plane
        if (p + (int)gridDim.x < planes) fetch(p + gridDim.x, 1);
        __syncthreads();
        s -= inv_own_sum(l.plane, N, k);
        const float g = inv_gather(l, own);
        if (tid < N) gx[((int64_t)b * planes + p) * N + tid] = g + s;
    }
}

// Same transposed-adjacency reduction for a tensor that has NO centre half: GPD [B,Ctot,3,N*k] -> gUV [B,2*Ctot,3,N] with
// gU[m] = sum over the edges pointing at m (channels [0,Ctot)) and gV[n] = sum over the k edges leaving n (channels
// [Ctot,2Ctot)).  One staged plane serves both sums.
__global__ __launch_bounds__(INV_T) void edge_grad_reduce_inv_kernel(const float* __restrict__ GPD, const int* __restrict__ in_ptr,
                                                                     const unsigned short* __restrict__ in_edge,
                                                                     const int* __restrict__ hub_cnt, const short* __restrict__ hub_slot,
                                                                     int Ctot, int N, int k, float* __restrict__ gUV) {
    extern __shared__ float4 inv_lds[];
    const int L = N * k, L4 = L >> 2;
    const InvLds l = inv_lds_layout(inv_lds, L);
    const int b = blockIdx.y, tid = threadIdx.x;
    const InvOwn own = inv_load_graph(l, in_ptr, in_edge, hub_cnt, hub_slot, b, N, L);
    float4 pre[INV_PF];
    auto fetch = [&](int p) {
        const float4* src = reinterpret_cast<const float4*>(GPD + ((int64_t)b * Ctot * 3 + p) * L);
#pragma unroll
        for (int i = 0; i < INV_PF; ++i) {
            const int q = tid + i * INV_T;
            pre[i] = q < L4 ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int planes = Ctot * 3;
    int p = blockIdx.x;
    if (p < planes) fetch(p);
    for (; p < planes; p += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < INV_PF; ++i) {
            const int q = tid + i * INV_T;
            if (q < L4) reinterpret_cast<float4*>(l.plane)[q] = pre[i];
        }
        if (p + (int)gridDim.x < planes) fetch(p + gridDim.x);
        __syncthreads();
        const float s = inv_own_sum(l.plane, N, k);
        const float g = inv_gather(l, own);
        if (tid < N) {
            gUV[((int64_t)b * 2 * planes + p) * N + tid] = g;
            gUV[((int64_t)b * 2 * planes + planes + p) * N + tid] = s;
        }
    }
}

// Slab form of edge_grad_reduce_inv_kernel for clouds whose N*k plane does not fit LDS (N = 2048, k = 40: 4 slabs of 512
// source points = 20480 edges): for each slab the workgroup loads that slab's transposed graph, then stages the slab of
// each of its (<= INV_PLMAX) planes in turn; gU partial sums stay in registers across the slabs, gV is complete per slab.
constexpr int INV_PLMAX = 8;
template <int NT>
__global__ __launch_bounds__(INV_T) void edge_grad_reduce_slab_kernel(const float* __restrict__ GPD, const int* __restrict__ in_ptr,
                                                                      const unsigned short* __restrict__ in_edge,
                                                                      const int* __restrict__ hub_cnt,
                                                                      const short* __restrict__ hub_slot, int Ctot, int N, int k,
                                                                      int S, float* __restrict__ gUV) {
    extern __shared__ float4 inv_lds[];
    const int Ns = N / S, Ls = Ns * k, L4 = Ls >> 2;
    const int64_t Ltot = (int64_t)N * k;
    const InvLds l = inv_lds_layout(inv_lds, Ls);
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int planes = Ctot * 3;
    const int npl = (planes - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // <= INV_PLMAX (host)
    float acc[INV_PLMAX][NT];
#pragma unroll
    for (int i = 0; i < INV_PLMAX; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = 0.f;
    float4 pre[INV_PF];
    auto fetch = [&](int q) {                        // q = s * npl + i
        const int sN = q / npl, p = blockIdx.x + (q % npl) * gridDim.x;
        const float4* src = reinterpret_cast<const float4*>(GPD + ((int64_t)b * planes + p) * Ltot + (int64_t)sN * Ls);
#pragma unroll
        for (int i = 0; i < INV_PF; ++i) {
            const int e4 = tid + i * INV_T;
            pre[i] = e4 < L4 ? src[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    const int total = S * npl;
    if (total > 0) fetch(0);
    for (int sl = 0; sl < S; ++sl) {
        const int bs = b * S + sl;
        __syncthreads();                             // the previous slab's graph and plane are no longer read
        for (int e = tid; e < Ls; e += INV_T) l.ine[e] = in_edge[(int64_t)bs * Ls + e];
        int e0[NT], e1[NT], hs[NT];
        const int nh = hub_cnt[bs];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = tid + j * INV_T;
            e0[j] = e1[j] = 0;
            hs[j] = -1;
            if (t < N) {
                e0[j] = in_ptr[(int64_t)bs * (N + 1) + t];
                e1[j] = in_ptr[(int64_t)bs * (N + 1) + t + 1];
                hs[j] = hub_slot[(int64_t)bs * N + t];
                if (hs[j] >= 0) { l.hub_e0[hs[j]] = e0[j]; l.hub_e1[hs[j]] = e1[j]; }
            }
        }
#pragma unroll
        for (int i = 0; i < INV_PLMAX; ++i) {
            if (i >= npl) break;                     // uniform
            const int p = blockIdx.x + i * gridDim.x;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < INV_PF; ++u) {
                const int e4 = tid + u * INV_T;
                if (e4 < L4) reinterpret_cast<float4*>(l.plane)[e4] = pre[u];
            }
            if (sl * npl + i + 1 < total) fetch(sl * npl + i + 1);
            __syncthreads();
            if (tid < Ns) {                          // gV of this slab's own points
                float sv = 0.f;
                const float4* r = reinterpret_cast<const float4*>(l.plane + tid * k);
                for (int j = 0; j < (k >> 2); ++j) {
                    const float4 v = r[j];
                    sv += (v.x + v.y) + (v.z + v.w);
                }
                gUV[((int64_t)b * 2 * planes + planes + p) * N + sl * Ns + tid] = sv;
            }
            for (int h = wv; h < nh; h += INV_T / 64) {
                float a = 0.f;
                for (int e = l.hub_e0[h] + lane; e < l.hub_e1[h]; e += 64) a += l.plane[l.ine[e]];
#pragma unroll
                for (int sh = 32; sh > 0; sh >>= 1) a += __shfl_xor(a, sh, 64);
                if (lane == 0) l.hub_g[h] = a;
            }
            float g[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                g[j] = 0.f;
                if (hs[j] < 0)
                    for (int e = e0[j]; e < e1[j]; ++e) g[j] += l.plane[l.ine[e]];
            }
            if (nh > 0) {                            // uniform
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (hs[j] >= 0) g[j] = l.hub_g[hs[j]];
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] += g[j];
        }
    }
#pragma unroll
    for (int i = 0; i < INV_PLMAX; ++i) {
        if (i >= npl) break;
        const int p = blockIdx.x + i * gridDim.x;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = tid + j * INV_T;
            if (t < N) gUV[((int64_t)b * 2 * planes + p) * N + t] = acc[i][j];
        }
    }
}

// builds the transposed graph(s) in `ws` and returns the tables; false when the geometry does not fit the LDS plans.
// *slabs = 1: one graph per cloud (N <= 1024, N*k <= INV_MAXL); > 1: N*k is a multiple of INV_MAXL and every slab of
// N / slabs source points owns exactly INV_MAXL edges (N <= 2048), allowed only when the caller handles slabs.
bool inv_graph_build(const int64_t* idx, int B, int N, int k, bool vec, void* ws, size_t ws_bytes, hipStream_t st, InvGraph* out,
                     int* slabs = nullptr, bool launch = true) {
    const bool off = hpcs_tuning().edge_bwd_atomic != 0;
    const int64_t L = (int64_t)N * k;
    if (off || !vec || (L % 4) != 0) return false;
    int S = 1;
    if (N > INV_T || L > INV_MAXL) {
        if (!slabs || N > 2 * INV_T || L % INV_MAXL != 0) return false;
        S = (int)(L / INV_MAXL);
        if (S > INV_MAXSLAB || N % S != 0 || (int64_t)(N / S) * k != INV_MAXL) return false;
    }
    const int Ls = (int)(L / S);
    const InvGraph g = inv_graph_layout(ws, B * S, N, Ls);
    if (ws_bytes < g.bytes) return false;
    if (!launch) {                                   // the tables were built earlier (hpcs_edge_graph_transpose)
    } else if (N <= INV_T)
        hipLaunchKernelGGL((knn_inverse_kernel<1>), dim3(B * S), dim3(INV_T), 0, st, idx, N, Ls, S, g.in_ptr, g.in_edge, g.hub_cnt,
                           g.hub_ids, g.hub_slot);
    else
        hipLaunchKernelGGL((knn_inverse_kernel<2>), dim3(B * S), dim3(INV_T), 0, st, idx, N, Ls, S, g.in_ptr, g.in_edge, g.hub_cnt,
                           g.hub_ids, g.hub_slot);
    *out = g;
    if (slabs) *slabs = S;
    return true;
}
int inv_groups(int B, int planes) {
    int groups = 256 / B;                            // ~one workgroup per CU; each walks planes/groups planes of its cloud
    if (groups < 1) groups = 1;
    return groups > planes ? planes : groups;
}

}  // namespace

extern "C" size_t hpcs_edge_gather_bwd_workspace(int B, int C, int N) {
    if (B <= 0 || C <= 0 || N <= 0) return 64;
    const int CP = (C * 3 + 15) / 16 * 16;
    return (size_t)B * N * CP * sizeof(float);
}

extern "C" int hpcs_edge_gather_bwd(const float* gout, const int64_t* idx, int B, int C, int N, int k, float* gx, void* ws,
                                    size_t ws_bytes, hpcs_stream_t stream) {
    if (!gout || !idx || !gx || !ws) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || k <= 0 || C > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (ws_bytes < hpcs_edge_gather_bwd_workspace(B, C, N)) return HPCS_ERR_WORKSPACE;
    const int64_t L = (int64_t)N * k;
    const int CP = (C * 3 + 15) / 16 * 16;
    const bool vec = (k % 4 == 0) && (((uintptr_t)gout & 15u) == 0);
    {   // inverse-graph path: the whole cloud's plane fits one workgroup's LDS
        InvGraph ig;
        if (inv_graph_build(idx, B, N, k, vec, ws, ws_bytes, ST, &ig)) {
            HPCS_CHECK_LAUNCH();
            (void)hipFuncSetAttribute((const void*)edge_gather_bwd_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)INV_LDS_BYTES);
            hipLaunchKernelGGL(edge_gather_bwd_inv_kernel, dim3(inv_groups(B, C * 3), B), dim3(INV_T), INV_LDS_BYTES, ST, gout,
                               (const int*)ig.in_ptr, (const unsigned short*)ig.in_edge, (const int*)ig.hub_cnt,
                               (const short*)ig.hub_slot, C, N, k, gx);
            HPCS_CHECK_LAUNCH();
            return HPCS_OK;
        }
    }
    if (vec && k <= 64)
        hipLaunchKernelGGL(edge_gather_bwd_center_tile_kernel, dim3(hpcs_cdiv(N, 64), C, B), dim3(16 * k), 0, ST, gout, C, N, k, gx);
    else if (vec)
        hipLaunchKernelGGL((edge_gather_bwd_center_kernel<true>), dim3(hpcs_cdiv(N, 64), C, B), dim3(64), 0, ST, gout, C, N, k, gx);
    else
        hipLaunchKernelGGL((edge_gather_bwd_center_kernel<false>), dim3(hpcs_cdiv(N, 64), C, B), dim3(64), 0, ST, gout, C, N, k, gx);
    HPCS_CHECK_LAUNCH();
    if (const int zr = hpcs_zero_async(ST, ws, (size_t)B * N * CP * sizeof(float))) return zr;
    const size_t lds = (size_t)ET * (C * 3 + 1) * sizeof(float);
    if (lds > 64 * 1024) return HPCS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(edge_scatter_tile_kernel, dim3(hpcs_cdiv(L, ET), B), dim3(256), lds, ST, gout, idx, C, 2 * C, N, k, CP, (float*)ws);
    HPCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(edge_scatter_fold_kernel, dim3(hpcs_cdiv(N, 64), B), dim3(64), 0, ST, (const float*)ws, C, C, N, CP, 1, gx);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

static BnFin bn_fin(unsigned* counter, int64_t count, float eps, float momentum, float* rmean, float* rvar, float* mean, float* rstd) {
    BnFin f;
    f.counter = counter;
    f.count = (double)count;
    f.eps = eps;
    f.momentum = momentum;
    f.rmean = rmean;
    f.rvar = rvar;
    f.mean = mean;
    f.rstd = rstd;
    return f;
}
static const BnFin NO_FIN = bn_fin(nullptr, 1, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr);

static int vn_norm_stats_impl(PdRef pd, int B, int Cout, double* sums, int zeroed, hipStream_t st, BnFin fin = NO_FIN) {
    if (!zeroed)
        if (const int zr = hpcs_zero_async(st, sums, sizeof(double) * 2 * Cout)) return zr;
    if (B == 0) return HPCS_OK;
    const bool vec_ok = pd.idx == nullptr && pd.L % 4 == 0 && (((uintptr_t)pd.base & 15u) == 0) && B <= 65535 && Cout <= 65535;
    // positions per workgroup: as many 1024-blocks as still leave ~2500 workgroups for the chip (at least VEC_ITEMS, at most 8)
    int items = VEC_ITEMS;
    {
        const int64_t blocks1k = hpcs_cdiv(pd.L, TPB * 4);
        const int64_t rows = (int64_t)B * Cout;
        while (items < 8 && hpcs_cdiv(blocks1k, items + 1) * rows >= 2500) ++items;
        items = hpcs_cdiv(blocks1k, hpcs_cdiv(blocks1k, items));           // same workgroup count, evenly filled
        if (items < 1) items = 1;
    }
    if (vec_ok && pd.w2 == nullptr)
        hipLaunchKernelGGL(vn_norm_stats_vec_kernel<false>, dim3(hpcs_cdiv(pd.L, TPB * 4 * items), Cout, B), dim3(TPB), 0, st,
                           pd.base, pd.Ctot, pd.L, Cout, sums, (const float*)nullptr, 0, fin, items);
    else if (vec_ok && Cout == 21 && pd.Ctot == 42)
        hipLaunchKernelGGL(vn_norm_stats_k2all_kernel<21>, dim3(hpcs_cdiv(pd.L, TPB * 4), B), dim3(TPB), 0, st, pd.base, pd.w2,
                           pd.ldw, pd.L, sums, fin);
    else if (vec_ok)
        hipLaunchKernelGGL(vn_norm_stats_vec_kernel<true>, dim3(hpcs_cdiv(pd.L, TPB * 4 * items), Cout, B), dim3(TPB), 0, st,
                           pd.base, pd.Ctot, pd.L, Cout, sums, pd.w2, pd.ldw, fin, items);
    else
        hipLaunchKernelGGL(vn_norm_stats_kernel, dim3(hpcs_cdiv((int64_t)B * pd.L, TPB * ITEMS), Cout), dim3(TPB), 0, st, pd, B,
                           pd.Ctot, pd.L, Cout, sums, fin);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_norm_stats(const float* PD, int B, int Cout, int Ctot, int64_t L, double* sums, int sums_zeroed, hpcs_stream_t stream) {
    if (!PD || !sums) return HPCS_ERR_NULL;
    if (bad_dims(B, Cout, L) || Ctot < Cout || Cout > 65535) return HPCS_ERR_SHAPE;
    return vn_norm_stats_impl(PdRef{PD, nullptr, Ctot, L, 0, 0}, B, Cout, sums, sums_zeroed, ST);
}

extern "C" int hpcs_vn_edge_norm_stats(const float* UV, const int64_t* idx, int B, int Cout, int Ctot, int N, int k,
                                       double* sums, int sums_zeroed, hpcs_stream_t stream) {
    if (!UV || !idx || !sums) return HPCS_ERR_NULL;
    if (B < 0 || Cout <= 0 || N <= 0 || k <= 0 || Ctot < Cout || Cout > 65535) return HPCS_ERR_SHAPE;
    return vn_norm_stats_impl(PdRef{UV, idx, Ctot, (int64_t)N * k, N, k}, B, Cout, sums, sums_zeroed, ST);
}

extern "C" int hpcs_bn_finalize(const double* sums, int C, int64_t count, float eps, float momentum, float* running_mean,
                                float* running_var, float* mean, float* rstd, int training, hpcs_stream_t stream) {
    if (!mean || !rstd || (training && !sums) || (!training && (!running_mean || !running_var))) return HPCS_ERR_NULL;
    if (C <= 0 || (training && count <= 0)) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(hpcs_cdiv(C, 128)), dim3(128), 0, ST, sums, C, (double)count, eps, momentum,
                       running_mean, running_var, mean, rstd, training);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

static int vn_args_ok(const void* PD, const void* mean, const void* rstd, const void* gamma, const void* beta, int B,
                      int Cout, int Cd, int64_t L, int pool_k, int flags) {
    if (!PD) return HPCS_ERR_NULL;
    if ((flags & HPCS_VN_BN) && (!mean || !rstd || !gamma || !beta)) return HPCS_ERR_NULL;
    if (bad_dims(B, Cout, L) || Cout > 65535) return HPCS_ERR_SHAPE;
    if ((flags & HPCS_VN_LRELU) ? !(Cd == Cout || Cd == 1) : Cd < 0) return HPCS_ERR_SHAPE;
    if (pool_k < 0 || (pool_k > 0 && L % pool_k != 0)) return HPCS_ERR_SHAPE;
    return HPCS_OK;
}

static int vn_apply_fwd_impl(PdRef pd, const float* mean, const float* rstd, const float* gamma, const float* beta, int B,
                             int Cout, int Cd, int pool_k, float slope, int flags, float* out, hipStream_t st) {
    const int64_t L = pd.L;
    if (pool_k > 0) {
        const bool vec = pd_materialised(pd) && (pool_k % 4 == 0) && (L % 4 == 0) && (((uintptr_t)pd.base & 15u) == 0);
        const dim3 grid(hpcs_cdiv((int64_t)B * (L / pool_k), 64), Cout);
        if (vec && pool_k <= 64 && B <= 65535 && Cout <= 65535)
            hipLaunchKernelGGL(vn_apply_pool_tile_kernel, dim3(hpcs_cdiv(L / pool_k, 64), Cout, B), dim3(16 * pool_k), 0, st,
                               pd.base, mean, rstd, gamma, beta, Cout, Cd, L, pool_k, slope, flags, out);
        else if (vec)
            hipLaunchKernelGGL((vn_apply_pool_fwd_kernel<true>), grid, dim3(64), 0, st, pd, mean, rstd, gamma, beta, B, Cout,
                               Cd, L, pool_k, slope, flags, out);
        else
            hipLaunchKernelGGL((vn_apply_pool_fwd_kernel<false>), grid, dim3(64), 0, st, pd, mean, rstd, gamma, beta, B, Cout,
                               Cd, L, pool_k, slope, flags, out);
    } else if (pd.idx == nullptr && L % 4 == 0 && L < (int64_t)1 << 31 && B <= 65535 && Cout <= 65535 &&
               ((((uintptr_t)pd.base | (uintptr_t)out) & 15u) == 0)) {
        const dim3 grid(hpcs_cdiv(L, (int64_t)TPB * 4), Cout, B);
        if (pd.w2 != nullptr)
            hipLaunchKernelGGL((vn_apply_fwd_vec_kernel<true>), grid, dim3(TPB), 0, st, pd.base, mean, rstd, gamma, beta, Cout, Cd,
                               (int)L, slope, flags, out, pd.w2, pd.ldw);
        else
            hipLaunchKernelGGL((vn_apply_fwd_vec_kernel<false>), grid, dim3(TPB), 0, st, pd.base, mean, rstd, gamma, beta, Cout,
                               Cd, (int)L, slope, flags, out, (const float*)nullptr, 0);
    } else {
        hipLaunchKernelGGL(vn_apply_fwd_kernel, dim3(hpcs_cdiv((int64_t)B * L, TPB), Cout), dim3(TPB), 0, st, pd, mean,
                           rstd, gamma, beta, B, Cout, Cd, L, slope, flags, out);
    }
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

static int vn_bwd_reduce_impl(PdRef pd, const float* gout, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, int B, int Cout, int Cd, int pool_k, float slope, int flags, double* red,
                              int zeroed, hipStream_t st) {
    if (!(flags & HPCS_VN_BN)) return HPCS_ERR_UNSUPPORTED;
    if (!zeroed)
        if (const int zr = hpcs_zero_async(st, red, sizeof(double) * 2 * Cout)) return zr;
    if (B == 0) return HPCS_OK;
    const bool vec = pd.idx == nullptr && pd.L % 4 == 0 && pd.L < ((int64_t)1 << 31) && (((uintptr_t)pd.base & 15u) == 0) && B <= 65535 &&
                     Cout <= 65535 &&
                     (pool_k > 0 ? pool_k % 4 == 0 : (((uintptr_t)gout & 15u) == 0));
    if (vec && pd.w2 == nullptr)
        hipLaunchKernelGGL(vn_bwd_reduce_vec_kernel<false>, dim3(hpcs_cdiv(pd.L, TPB * 4 * VEC_ITEMS), Cout, B), dim3(TPB), 0, st,
                           pd.base, gout, mean, rstd, gamma, beta, Cout, Cd, pd.L, pool_k, slope, flags, red, (const float*)nullptr, 0);
    else if (vec && pool_k == 0 && Cout == 21 && Cd == 21 && flags == (HPCS_VN_BN | HPCS_VN_LRELU))
        hipLaunchKernelGGL(vn_bwd_reduce_k2all_kernel<21>, dim3(hpcs_cdiv(pd.L, TPB * 4), B), dim3(TPB), 0, st, pd.base, pd.w2,
                           pd.ldw, gout, mean, rstd, gamma, beta, pd.L, slope, flags, red);
    else if (vec)
        hipLaunchKernelGGL(vn_bwd_reduce_vec_kernel<true>, dim3(hpcs_cdiv(pd.L, TPB * 4 * VEC_ITEMS), Cout, B), dim3(TPB), 0, st,
                           pd.base, gout, mean, rstd, gamma, beta, Cout, Cd, pd.L, pool_k, slope, flags, red, pd.w2, pd.ldw);
    else
        hipLaunchKernelGGL(vn_bwd_reduce_kernel, dim3(hpcs_cdiv((int64_t)B * pd.L, TPB * ITEMS), Cout), dim3(TPB), 0, st, pd, gout,
                           mean, rstd, gamma, beta, B, Cout, Cd, pd.L, pool_k, slope, flags, red);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

static int vn_bwd_apply_impl(PdRef pd, const float* gout, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, const double* red, int training, int B, int Cout, int Cd, int pool_k,
                             float slope, int flags, float* GPD, hipStream_t st, float* ggamma = nullptr,
                             float* gbeta = nullptr) {
    if (B == 0) return HPCS_OK;
    const int64_t L = pd.L;
    if (Cd == 1 && (flags & HPCS_VN_LRELU)) {
        const int Ctot = Cout + Cd;
        const int chunks = Cout < 16 ? Cout : 16, c_chunk = hpcs_cdiv(Cout, chunks);
        // zero the B direction blocks [3][L] (row pitch Ctot*3*L floats) that the channel chunks add into
        if (const int zr = hpcs_zero_rows_async(st, GPD + (int64_t)Cout * 3 * L, (size_t)Ctot * 3 * L, 3 * (size_t)L, B)) return zr;
        if (pd_materialised(pd) && pool_k == 0 && L % 4 == 0 && B <= 65535 &&
            ((((uintptr_t)pd.base | (uintptr_t)gout | (uintptr_t)GPD) & 15u) == 0)) {
            // enough channel chunks to put >= ~1500 workgroups on the chip, at least 4 channels each
            const int64_t per = hpcs_cdiv(L, (int64_t)TPB * 4) * B;
            int ch = (int)hpcs_cdiv((int64_t)1536, per);
            ch = ch < 1 ? 1 : ch;
            int cc = hpcs_cdiv(Cout, ch);
            cc = cc < 4 ? (Cout < 4 ? Cout : 4) : cc;
            hipLaunchKernelGGL(vn_bwd_apply_shared_vec_kernel, dim3(hpcs_cdiv(L, (int64_t)TPB * 4), B, hpcs_cdiv(Cout, cc)), dim3(TPB), 0,
                               st, pd.base, gout, mean, rstd, gamma, beta, red, training, B, Cout, L, slope, flags, GPD, cc, ggamma,
                               gbeta);
        } else {
            hipLaunchKernelGGL((vn_bwd_apply_kernel<true>), dim3(hpcs_cdiv((int64_t)B * L, TPB), hpcs_cdiv(Cout, c_chunk)), dim3(TPB),
                               0, st, pd, gout, mean, rstd, gamma, beta, red, training, B, Cout, Cd, L, pool_k, slope, flags, GPD,
                               c_chunk, ggamma, gbeta);
        }
    } else if (pd.idx == nullptr && L % 4 == 0 && L < (int64_t)1 << 31 && B <= 65535 && Cout <= 65535 && (Cd == 0 || Cd == Cout) &&
               ((((uintptr_t)pd.base | (uintptr_t)GPD) & 15u) == 0) &&
               (pool_k > 0 ? pool_k % 4 == 0 : (((uintptr_t)gout & 15u) == 0))) {
        const dim3 grid(hpcs_cdiv(L, (int64_t)TPB * 4), Cout, B);
        if (pd.w2 != nullptr)
            hipLaunchKernelGGL((vn_bwd_apply_vec_kernel<true>), grid, dim3(TPB), 0, st, pd.base, gout, mean, rstd, gamma, beta, red,
                               training, B, Cout, Cd, (int)L, pool_k, slope, flags, GPD, ggamma, gbeta, pd.w2, pd.ldw);
        else
            hipLaunchKernelGGL((vn_bwd_apply_vec_kernel<false>), grid, dim3(TPB), 0, st, pd.base, gout, mean, rstd, gamma, beta,
                               red, training, B, Cout, Cd, (int)L, pool_k, slope, flags, GPD, ggamma, gbeta,
                               (const float*)nullptr, 0);
    } else {
        hipLaunchKernelGGL((vn_bwd_apply_kernel<false>), dim3(hpcs_cdiv((int64_t)B * L, TPB), Cout), dim3(TPB), 0, st, pd,
                           gout, mean, rstd, gamma, beta, red, training, B, Cout, Cd, L, pool_k, slope, flags, GPD, 1, ggamma, gbeta);
    }
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_apply_fwd(const float* PD, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, int B, int Cout, int Cd, int64_t L, int pool_k, float slope, int flags,
                                 float* out, hpcs_stream_t stream) {
    int rc = vn_args_ok(PD, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, flags);
    if (rc) return rc;
    if (!out) return HPCS_ERR_NULL;
    if (B == 0) return HPCS_OK;
    return vn_apply_fwd_impl(PdRef{PD, nullptr, Cout + Cd, L, 0, 0}, mean, rstd, gamma, beta, B, Cout, Cd, pool_k, slope,
                             flags, out, ST);
}

extern "C" int hpcs_vn_bwd_reduce(const float* PD, const float* gout, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, int B, int Cout, int Cd, int64_t L, int pool_k,
                                  float slope, int flags, double* red, int red_zeroed, hpcs_stream_t stream) {
    int rc = vn_args_ok(PD, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, flags);
    if (rc) return rc;
    if (!gout || !red) return HPCS_ERR_NULL;
    return vn_bwd_reduce_impl(PdRef{PD, nullptr, Cout + Cd, L, 0, 0}, gout, mean, rstd, gamma, beta, B, Cout, Cd, pool_k,
                              slope, flags, red, red_zeroed, ST);
}

extern "C" int hpcs_vn_bwd_apply(const float* PD, const float* gout, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, const double* red, int training, int B, int Cout,
                                 int Cd, int64_t L, int pool_k, float slope, int flags, float* GPD, float* ggamma,
                                 float* gbeta, hpcs_stream_t stream) {
    int rc = vn_args_ok(PD, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, flags);
    if (rc) return rc;
    if (!gout || !GPD) return HPCS_ERR_NULL;
    if ((flags & HPCS_VN_BN) && training && !red) return HPCS_ERR_NULL;
    if ((ggamma == nullptr) != (gbeta == nullptr) || (ggamma && (!(flags & HPCS_VN_BN) || !red))) return HPCS_ERR_NULL;
    return vn_bwd_apply_impl(PdRef{PD, nullptr, Cout + Cd, L, 0, 0}, gout, mean, rstd, gamma, beta, red, training, B, Cout,
                             Cd, pool_k, slope, flags, GPD, ST, ggamma, gbeta);
}

// ---- two-input-channel layers (conv1): PD = W E formed on the fly from the edge tensor E [B,2,3,L] and the stacked weight
// W2 [Cout+Cd rows, pitch ldw >= 2] (PdRef::w2); same kernels, same point-wise code, bit-identical values.
static int k2_args_ok(const void* E, const void* W2, int ldw) { return (!E || !W2) ? HPCS_ERR_NULL : (ldw < 2 ? HPCS_ERR_SHAPE : HPCS_OK); }

extern "C" int hpcs_vn_norm_stats_k2(const float* E, const float* W2, int ldw, int B, int Cout, int Ctot, int64_t L,
                                     double* sums, int sums_zeroed, hpcs_stream_t stream) {
    int rc = k2_args_ok(E, W2, ldw);
    if (rc) return rc;
    if (!sums) return HPCS_ERR_NULL;
    if (bad_dims(B, Cout, L) || Ctot < Cout) return HPCS_ERR_SHAPE;
    PdRef pd{E, nullptr, Ctot, L, 0, 0};
    pd.w2 = W2;
    pd.ldw = ldw;
    return vn_norm_stats_impl(pd, B, Cout, sums, sums_zeroed, ST);
}

extern "C" int hpcs_vn_apply_fwd_k2(const float* E, const float* W2, int ldw, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, int B, int Cout, int Cd, int64_t L, int pool_k,
                                    float slope, int flags, float* out, hpcs_stream_t stream) {
    int rc = k2_args_ok(E, W2, ldw);
    if (rc) return rc;
    rc = vn_args_ok(E, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, flags);
    if (rc) return rc;
    if (!out) return HPCS_ERR_NULL;
    PdRef pd{E, nullptr, Cout + Cd, L, 0, 0};
    pd.w2 = W2;
    pd.ldw = ldw;
    return vn_apply_fwd_impl(pd, mean, rstd, gamma, beta, B, Cout, Cd, pool_k, slope, flags, out, ST);
}

extern "C" int hpcs_vn_bwd_reduce_k2(const float* E, const float* W2, int ldw, const float* gout, const float* mean,
                                     const float* rstd, const float* gamma, const float* beta, int B, int Cout, int Cd,
                                     int64_t L, int pool_k, float slope, int flags, double* red, int red_zeroed,
                                     hpcs_stream_t stream) {
    int rc = k2_args_ok(E, W2, ldw);
    if (rc) return rc;
    rc = vn_args_ok(E, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, flags);
    if (rc) return rc;
    if (!gout || !red) return HPCS_ERR_NULL;
    PdRef pd{E, nullptr, Cout + Cd, L, 0, 0};
    pd.w2 = W2;
    pd.ldw = ldw;
    return vn_bwd_reduce_impl(pd, gout, mean, rstd, gamma, beta, B, Cout, Cd, pool_k, slope, flags, red, red_zeroed, ST);
}

extern "C" int hpcs_vn_bwd_apply_k2(const float* E, const float* W2, int ldw, const float* gout, const float* mean,
                                    const float* rstd, const float* gamma, const float* beta, const double* red,
                                    int training, int B, int Cout, int Cd, int64_t L, int pool_k, float slope, int flags,
                                    float* GPD, float* ggamma, float* gbeta, hpcs_stream_t stream) {
    int rc = k2_args_ok(E, W2, ldw);
    if (rc) return rc;
    rc = vn_args_ok(E, mean, rstd, gamma, beta, B, Cout, Cd, L, pool_k, flags);
    if (rc) return rc;
    if (!gout || !GPD) return HPCS_ERR_NULL;
    if ((flags & HPCS_VN_BN) && training && !red) return HPCS_ERR_NULL;
    if ((ggamma == nullptr) != (gbeta == nullptr) || (ggamma && (!(flags & HPCS_VN_BN) || !red))) return HPCS_ERR_NULL;
    PdRef pd{E, nullptr, Cout + Cd, L, 0, 0};
    pd.w2 = W2;
    pd.ldw = ldw;
    return vn_bwd_apply_impl(pd, gout, mean, rstd, gamma, beta, red, training, B, Cout, Cd, pool_k, slope, flags, GPD, ST,
                             ggamma, gbeta);
}

static int edge_args_ok(const void* UV, const void* idx, int B, int N, int k) {
    if (!UV || !idx) return HPCS_ERR_NULL;
    if (B < 0 || N <= 0 || k <= 0) return HPCS_ERR_SHAPE;
    return HPCS_OK;
}

extern "C" int hpcs_vn_edge_apply_fwd(const float* UV, const int64_t* idx, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int B, int Cout, int Cd, int N, int k,
                                      int pool, float slope, int flags, float* out, hpcs_stream_t stream) {
    int rc = edge_args_ok(UV, idx, B, N, k);
    if (rc) return rc;
    rc = vn_args_ok(UV, mean, rstd, gamma, beta, B, Cout, Cd, (int64_t)N * k, pool ? k : 0, flags);
    if (rc) return rc;
    if (!out) return HPCS_ERR_NULL;
    if (B == 0) return HPCS_OK;
    return vn_apply_fwd_impl(PdRef{UV, idx, Cout + Cd, (int64_t)N * k, N, k}, mean, rstd, gamma, beta, B, Cout, Cd,
                             pool ? k : 0, slope, flags, out, ST);
}

extern "C" int hpcs_vn_edge_bwd_reduce(const float* UV, const int64_t* idx, const float* gout, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta, int B, int Cout, int Cd,
                                       int N, int k, int pool, float slope, int flags, double* red, int red_zeroed, hpcs_stream_t stream) {
    int rc = edge_args_ok(UV, idx, B, N, k);
    if (rc) return rc;
    rc = vn_args_ok(UV, mean, rstd, gamma, beta, B, Cout, Cd, (int64_t)N * k, pool ? k : 0, flags);
    if (rc) return rc;
    if (!gout || !red) return HPCS_ERR_NULL;
    return vn_bwd_reduce_impl(PdRef{UV, idx, Cout + Cd, (int64_t)N * k, N, k}, gout, mean, rstd, gamma, beta, B, Cout, Cd,
                              pool ? k : 0, slope, flags, red, red_zeroed, ST);
}

extern "C" int hpcs_vn_edge_bwd_apply(const float* UV, const int64_t* idx, const float* gout, const float* mean,
                                      const float* rstd, const float* gamma, const float* beta, const double* red,
                                      int training, int B, int Cout, int Cd, int N, int k, int pool, float slope, int flags,
                                      float* GPD, hpcs_stream_t stream) {
    int rc = edge_args_ok(UV, idx, B, N, k);
    if (rc) return rc;
    rc = vn_args_ok(UV, mean, rstd, gamma, beta, B, Cout, Cd, (int64_t)N * k, pool ? k : 0, flags);
    if (rc) return rc;
    if (!gout || !GPD) return HPCS_ERR_NULL;
    if ((flags & HPCS_VN_BN) && training && !red) return HPCS_ERR_NULL;
    return vn_bwd_apply_impl(PdRef{UV, idx, Cout + Cd, (int64_t)N * k, N, k}, gout, mean, rstd, gamma, beta, red, training,
                             B, Cout, Cd, pool ? k : 0, slope, flags, GPD, ST);
}

extern "C" size_t hpcs_edge_grad_reduce_workspace(int B, int Ctot, int N) {
    if (B <= 0 || Ctot <= 0 || N <= 0) return 64;
    const int CP = (Ctot * 3 + 15) / 16 * 16;
    const size_t atomic_path = (size_t)B * N * CP * sizeof(float);
    // transposed graphs: k is not known here, so room for the largest plan (INV_MAXSLAB slabs of INV_MAXL edges per cloud)
    const size_t graphs = N <= 2 * INV_T ? inv_graph_layout(nullptr, B * INV_MAXSLAB, N, INV_MAXL).bytes : 0;
    return atomic_path > graphs ? atomic_path : graphs;
}

// GPD [B,Ctot,3,N*k] -> gUV [B,2*Ctot,3,N]: gU = scatter-add over idx (row-contiguous atomics), gV = sum over k
static int edge_grad_reduce_impl(const float* GPD, const int64_t* idx, int B, int Ctot, int N, int k, float* gUV, void* ws,
                                 size_t ws_bytes, hpcs_stream_t stream, bool prebuilt) {
    if (!GPD || (!idx && !prebuilt) || !gUV || !ws) return HPCS_ERR_NULL;
    if (B < 0 || Ctot <= 0 || N <= 0 || k <= 0 || Ctot > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (ws_bytes < hpcs_edge_grad_reduce_workspace(B, Ctot, N)) return HPCS_ERR_WORKSPACE;
    const int64_t L = (int64_t)N * k;
    const int CP = (Ctot * 3 + 15) / 16 * 16;
    const bool vec = (k % 4 == 0) && (((uintptr_t)GPD & 15u) == 0);
    {   // inverse-graph path (see edge_gather_bwd_inv_kernel): no atomics, no scratch memset, GPD read once
        InvGraph ig;
        int slabs = 1;
        if (inv_graph_build(idx, B, N, k, vec, ws, ws_bytes, ST, &ig, &slabs, !prebuilt)) {
            HPCS_CHECK_LAUNCH();
            if (slabs > 1) {
                int groups = inv_groups(B, Ctot * 3);
                const int need = hpcs_cdiv(Ctot * 3, INV_PLMAX);
                if (groups < need) groups = need;
                const dim3 grid(groups, B);
                if (N <= INV_T) {
                    (void)hipFuncSetAttribute((const void*)edge_grad_reduce_slab_kernel<1>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_LDS_BYTES);
                    hipLaunchKernelGGL((edge_grad_reduce_slab_kernel<1>), grid, dim3(INV_T), INV_LDS_BYTES, ST, GPD,
                                       (const int*)ig.in_ptr, (const unsigned short*)ig.in_edge, (const int*)ig.hub_cnt,
                                       (const short*)ig.hub_slot, Ctot, N, k, slabs, gUV);
                } else {
                    (void)hipFuncSetAttribute((const void*)edge_grad_reduce_slab_kernel<2>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_LDS_BYTES);
                    hipLaunchKernelGGL((edge_grad_reduce_slab_kernel<2>), grid, dim3(INV_T), INV_LDS_BYTES, ST, GPD,
                                       (const int*)ig.in_ptr, (const unsigned short*)ig.in_edge, (const int*)ig.hub_cnt,
                                       (const short*)ig.hub_slot, Ctot, N, k, slabs, gUV);
                }
                HPCS_CHECK_LAUNCH();
                return HPCS_OK;
            }
            (void)hipFuncSetAttribute((const void*)edge_grad_reduce_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)INV_LDS_BYTES);
            hipLaunchKernelGGL(edge_grad_reduce_inv_kernel, dim3(inv_groups(B, Ctot * 3), B), dim3(INV_T), INV_LDS_BYTES, ST, GPD,
                               (const int*)ig.in_ptr, (const unsigned short*)ig.in_edge, (const int*)ig.hub_cnt,
                               (const short*)ig.hub_slot, Ctot, N, k, gUV);
            HPCS_CHECK_LAUNCH();
            return HPCS_OK;
        }
    }
    if (prebuilt) return HPCS_ERR_UNSUPPORTED;           // the transposed graph does not serve this call (alignment / geometry)
    if (vec)
        hipLaunchKernelGGL((edge_segsum_kernel<true>), dim3(hpcs_cdiv(N, 64), Ctot, B), dim3(64), 0, ST, GPD, Ctot, N, k, gUV);
    else
        hipLaunchKernelGGL((edge_segsum_kernel<false>), dim3(hpcs_cdiv(N, 64), Ctot, B), dim3(64), 0, ST, GPD, Ctot, N, k, gUV);
    HPCS_CHECK_LAUNCH();
    if (const int zr = hpcs_zero_async(ST, ws, (size_t)B * N * CP * sizeof(float))) return zr;
    const size_t lds = (size_t)ET * (Ctot * 3 + 1) * sizeof(float);
    if (lds > 64 * 1024) return HPCS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(edge_scatter_tile_kernel, dim3(hpcs_cdiv(L, ET), B), dim3(256), lds, ST, GPD, idx, Ctot, Ctot, N, k, CP,
                       (float*)ws);
    HPCS_CHECK_LAUNCH();
    hipLaunchKernelGGL(edge_scatter_fold_kernel, dim3(hpcs_cdiv(N, 64), B), dim3(64), 0, ST, (const float*)ws, Ctot, 2 * Ctot, N,
                       CP, 0, gUV);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_edge_grad_reduce(const float* GPD, const int64_t* idx, int B, int Ctot, int N, int k, float* gUV,
                                     void* ws, size_t ws_bytes, hpcs_stream_t stream) {
    return edge_grad_reduce_impl(GPD, idx, B, Ctot, N, k, gUV, ws, ws_bytes, stream, false);
}

// The transposed graph depends on the kNN indices only: it can be built as soon as they exist (in the forward pass, on a side
// stream) and handed to the backward pass in the workspace.
extern "C" int hpcs_edge_graph_transpose(const int64_t* idx, int B, int Ctot, int N, int k, void* ws, size_t ws_bytes,
                                         hpcs_stream_t stream) {
    if (!idx || !ws) return HPCS_ERR_NULL;
    if (B < 0 || Ctot <= 0 || N <= 0 || k <= 0 || Ctot > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (ws_bytes < hpcs_edge_grad_reduce_workspace(B, Ctot, N)) return HPCS_ERR_WORKSPACE;
    InvGraph ig;
    int slabs = 1;
    if (!inv_graph_build(idx, B, N, k, k % 4 == 0, ws, ws_bytes, ST, &ig, &slabs)) return HPCS_ERR_UNSUPPORTED;
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_edge_grad_reduce_prebuilt(const float* GPD, int B, int Ctot, int N, int k, float* gUV, void* ws, size_t ws_bytes,
                                              hpcs_stream_t stream) {
    return edge_grad_reduce_impl(GPD, nullptr, B, Ctot, N, k, gUV, ws, ws_bytes, stream, true);
}

extern "C" int hpcs_bn_stats(const float* Y, int B, int C, int64_t L, double* sums, int sums_zeroed, hpcs_stream_t stream) {
    if (!Y || !sums) return HPCS_ERR_NULL;
    if (bad_dims(B, C, L) || C > 65535) return HPCS_ERR_SHAPE;
    if (!sums_zeroed)
        if (const int zr = hpcs_zero_async(ST, sums, sizeof(double) * 2 * C)) return zr;
    if (B == 0) return HPCS_OK;
    if (bn_vec_ok(B, L, Y, nullptr, nullptr, nullptr)) {
        const int bpw = bn_bpw(B, C);
        hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(hpcs_cdiv(B, bpw), C), dim3(TPB), 0, ST, Y, B, C, L, bpw, sums, NO_FIN);
    } else {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(hpcs_cdiv((int64_t)B * L, TPB * ITEMS), C), dim3(TPB), 0, ST, Y, B, C, L, sums, NO_FIN);
    }
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

// dropout by counter hash (mask == NULL and 0 < drop_p < 1): keep iff hash24(seed, element offset) >= p * 2^24
static inline unsigned drop_thresh(float p) { return p > 0.f && p < 1.f ? (unsigned)(p * 16777216.0f + 0.5f) : 0u; }
static inline float drop_inv(float p) { return p > 0.f && p < 1.f ? 1.0f / (1.0f - p) : 1.0f; }

extern "C" int hpcs_bn_act_fwd(const float* Y, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const float* mask, uint64_t drop_seed, const uint64_t* drop_seed_dev,
                               float drop_p, int B, int C, int64_t L, float slope, float* out, hpcs_stream_t stream) {
    if (!Y || !mean || !rstd || !gamma || !beta || !out) return HPCS_ERR_NULL;
    if (bad_dims(B, C, L) || C > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (bn_vec_ok(B, L, Y, out, mask, nullptr))
        hipLaunchKernelGGL(bn_act_fwd_vec_kernel, dim3(hpcs_cdiv(L, TPB * 4), C, B), dim3(TPB), 0, ST, Y, mean, rstd, gamma, beta,
                           mask, C, L, slope, out, (unsigned long long)drop_seed, (const unsigned long long*)drop_seed_dev,
                           drop_thresh(drop_p), drop_inv(drop_p));
    else
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(hpcs_cdiv((int64_t)B * L, TPB), C), dim3(TPB), 0, ST, Y, mean, rstd, gamma,
                       beta, mask, B, C, L, slope, out, (unsigned long long)drop_seed,
                       (const unsigned long long*)drop_seed_dev, drop_thresh(drop_p), drop_inv(drop_p));
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_bn_act_bwd_reduce(const float* Y, const float* gout, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, const float* mask, uint64_t drop_seed,
                                      const uint64_t* drop_seed_dev, float drop_p, int B, int C, int64_t L, float slope,
                                      double* red, int red_zeroed,
                                      hpcs_stream_t stream) {
    if (!Y || !gout || !mean || !rstd || !gamma || !beta || !red) return HPCS_ERR_NULL;
    if (bad_dims(B, C, L) || C > 65535) return HPCS_ERR_SHAPE;
    if (!red_zeroed)
        if (const int zr = hpcs_zero_async(ST, red, sizeof(double) * 2 * C)) return zr;
    if (B == 0) return HPCS_OK;
    if (bn_vec_ok(B, L, Y, gout, mask, nullptr)) {
        const int bpw = bn_bpw(B, C);
        hipLaunchKernelGGL(bn_act_bwd_reduce_vec_kernel, dim3(hpcs_cdiv(B, bpw), C), dim3(TPB), 0, ST, Y, gout, mean, rstd, gamma,
                           beta, mask, B, C, L, bpw, slope, red, (unsigned long long)drop_seed,
                           (const unsigned long long*)drop_seed_dev, drop_thresh(drop_p), drop_inv(drop_p));
    } else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(hpcs_cdiv((int64_t)B * L, TPB * ITEMS), C), dim3(TPB), 0, ST, Y, gout,
                       mean, rstd, gamma, beta, mask, B, C, L, slope, red, (unsigned long long)drop_seed,
                       (const unsigned long long*)drop_seed_dev, drop_thresh(drop_p), drop_inv(drop_p));
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_bn_act_bwd_apply(const float* Y, const float* gout, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, const float* mask, uint64_t drop_seed,
                                     const uint64_t* drop_seed_dev, float drop_p, const double* red, int training, int B,
                                     int C, int64_t L, float slope, float* gY, float* ggamma, float* gbeta,
                                     hpcs_stream_t stream) {
    if (!Y || !gout || !mean || !rstd || !gamma || !beta || !gY || (training && !red)) return HPCS_ERR_NULL;
    if ((ggamma == nullptr) != (gbeta == nullptr) || (ggamma && !red)) return HPCS_ERR_NULL;
    if (bad_dims(B, C, L) || C > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (bn_vec_ok(B, L, Y, gout, mask, gY))
        hipLaunchKernelGGL(bn_act_bwd_apply_vec_kernel, dim3(hpcs_cdiv(L, TPB * 4), C, B), dim3(TPB), 0, ST, Y, gout, mean, rstd,
                           gamma, beta, mask, red, training, B, C, L, slope, gY, (unsigned long long)drop_seed,
                           (const unsigned long long*)drop_seed_dev, drop_thresh(drop_p), drop_inv(drop_p), ggamma, gbeta);
    else
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(hpcs_cdiv((int64_t)B * L, TPB), C), dim3(TPB), 0, ST, Y, gout, mean,
                       rstd, gamma, beta, mask, red, training, B, C, L, slope, gY, (unsigned long long)drop_seed,
                       (const unsigned long long*)drop_seed_dev, drop_thresh(drop_p), drop_inv(drop_p), ggamma, gbeta);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_dropout_mask(uint64_t drop_seed, float drop_p, int64_t n, float* out, hpcs_stream_t stream) {
    if (!out) return HPCS_ERR_NULL;
    if (n < 0 || !(drop_p >= 0.f && drop_p < 1.f)) return HPCS_ERR_SHAPE;
    if (n == 0) return HPCS_OK;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(hpcs_cdiv(n, TPB)), dim3(TPB), 0, ST, (unsigned long long)drop_seed,
                       drop_thresh(drop_p), drop_inv(drop_p), n, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_concat_mean_fwd(const float* x, int B, int C, int N, float* out, hpcs_stream_t stream) {
    if (!x || !out) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(concat_mean_fwd_kernel, dim3((unsigned)((int64_t)B * C * 3)), dim3(TPB), 0, ST, x, C, N, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_concat_mean_bwd(const float* gout, int B, int C, int N, float* gx, hpcs_stream_t stream) {
    if (!gout || !gx) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(concat_mean_bwd_kernel, dim3((unsigned)((int64_t)B * C * 3)), dim3(TPB), 0, ST, gout, C, N, gx);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_frame_project_fwd(const float* x, const float* z, int B, int C, int N, float* out,
                                      hpcs_stream_t stream) {
    if (!x || !z || !out) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || C > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(frame_project_fwd_kernel, dim3(hpcs_cdiv(N, TPB), C, B), dim3(TPB), 0, ST, x, z, C, N, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_frame_project_bwd(const float* x, const float* z, const float* gout, int B, int C, int N, float* gx,
                                      float* gz, hpcs_stream_t stream) {
    if (!x || !z || !gout || !gx || !gz) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(frame_project_bwd_kernel, dim3(hpcs_cdiv(N, 64), B), dim3(64), 0, ST, x, z, gout, C, N, gx, gz);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_project_max_fwd(const float* x, const float* xconst, const float* z, int B, int C, int Cc, int N,
                                    float* out, int32_t* arg, hpcs_stream_t stream) {
    if (!x || !z || !out || !arg || (Cc > 0 && !xconst)) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || Cc < 0 || N <= 0 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(project_max_fwd_kernel, dim3(C + Cc, B), dim3(256), 0, ST, x, xconst, z, C, Cc, N, out, arg);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_project_max_bwd(const float* x, const float* xconst, const float* z, const int32_t* arg,
                                    const float* gout, int B, int C, int Cc, int N, float* gx, int gx_accumulate,
                                    float* gxconst, float* gz, hpcs_stream_t stream) {
    if (!x || !z || !arg || !gout || (!gx && !gz) || (Cc > 0 && (!xconst || (gz && !gxconst)))) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || Cc < 0 || N <= 0) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    if (const int zr = hpcs_zero_async(ST, (gx && !gx_accumulate) ? gx : nullptr, sizeof(float) * (size_t)B * C * 3 * N, gz,
                                       sizeof(float) * (size_t)B * 9 * N))
        return zr;
    hipLaunchKernelGGL(project_max_bwd_kernel, dim3(hpcs_cdiv((int64_t)B * (C + Cc), 128)), dim3(128), 0, ST, x, xconst, z, arg,
                       gout, B, C, Cc, N, gx, gxconst, gz);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_rowmax_fwd(const float* x, int64_t rows, int64_t L, float* out, int32_t* arg, hpcs_stream_t stream) {
    if (!x || !out) return HPCS_ERR_NULL;
    if (rows < 0 || L <= 0 || L > 0x7fffffff) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(rowmax_fwd_kernel, dim3((unsigned)rows), dim3(TPB), 0, ST, x, L, out, arg);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_rowmax_bwd(const float* gout, const int32_t* arg, int64_t rows, int64_t L, float* gx,
                               hpcs_stream_t stream) {
    if (!gout || !arg || !gx) return HPCS_ERR_NULL;
    if (rows < 0 || L <= 0) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(rowmax_bwd_kernel, dim3(hpcs_cdiv(rows * L, TPB)), dim3(TPB), 0, ST, gout, arg, rows, L, gx);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_rowsum(const float* x, int64_t rows, int64_t L, float scale, float* out, hpcs_stream_t stream) {
    if (!x || !out) return HPCS_ERR_NULL;
    if (rows < 0 || L <= 0) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    if (rows <= 64 && L >= 8192 && L % 4 == 0 && (((uintptr_t)x & 15u) == 0))
        hipLaunchKernelGGL(rowsum_long_kernel, dim3((unsigned)rows), dim3(1024), 0, ST, x, L, scale, out);
    else if (rows >= 1024 && L >= 256 && L % 4 == 0 && (((uintptr_t)x & 15u) == 0))
        hipLaunchKernelGGL(rowsum_wave_kernel, dim3((unsigned)hpcs_cdiv(rows, 4)), dim3(256), 0, ST, x, rows, L, scale, out);
    else
        hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)rows), dim3(L >= 256 ? 256 : 64), 0, ST, x, L, scale, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_maxpool_fwd(const float* x, const float* d, int B, int C, int N, int k, float* out, int32_t* arg,
                                   hpcs_stream_t stream) {
    if (!x || !d || !out || !arg) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || k <= 0 || C > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(vn_maxpool_fwd_kernel, dim3(hpcs_cdiv(N, TPB), C, B), dim3(TPB), 0, ST, x, d, C, N, k, out, arg);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_vn_maxpool_bwd(const float* gout, const int32_t* arg, int B, int C, int N, int k, float* gx,
                                   hpcs_stream_t stream) {
    if (!gout || !arg || !gx) return HPCS_ERR_NULL;
    if (B < 0 || C <= 0 || N <= 0 || k <= 0 || C > 65535 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(vn_maxpool_bwd_kernel, dim3(hpcs_cdiv((int64_t)N * k, TPB), C, B), dim3(TPB), 0, ST, gout, arg, C, N,
                       k, gx);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_scale_inplace(float* x, int64_t n, float s, hpcs_stream_t stream) {
    if (!x) return HPCS_ERR_NULL;
    if (n < 0) return HPCS_ERR_SHAPE;
    if (n == 0) return HPCS_OK;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3(hpcs_cdiv(n, TPB)), dim3(TPB), 0, ST, x, n, s);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

// ---- statistics + BatchNorm finalisation in ONE launch (training mode): the last workgroup finalises --------------------------
static int fin_args_ok(const void* sums, const void* counter, const void* mean, const void* rstd, int64_t count) {
    if (!sums || !counter || !mean || !rstd) return HPCS_ERR_NULL;
    if (count <= 0) return HPCS_ERR_SHAPE;
    return HPCS_OK;
}

extern "C" int hpcs_vn_norm_stats_finalize(const float* PD, int B, int Cout, int Ctot, int64_t L, double* sums, uint32_t* counter,
                                           float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                           float* rstd, hpcs_stream_t stream) {
    if (!PD) return HPCS_ERR_NULL;
    int rc = fin_args_ok(sums, counter, mean, rstd, (int64_t)B * L);
    if (rc) return rc;
    if (bad_dims(B, Cout, L) || Ctot < Cout || Cout > 65535 || B == 0) return HPCS_ERR_SHAPE;
    return vn_norm_stats_impl(PdRef{PD, nullptr, Ctot, L, 0, 0}, B, Cout, sums, 1, ST,
                              bn_fin(counter, (int64_t)B * L, eps, momentum, running_mean, running_var, mean, rstd));
}

extern "C" int hpcs_vn_norm_stats_k2_finalize(const float* E, const float* W2, int ldw, int B, int Cout, int Ctot, int64_t L,
                                              double* sums, uint32_t* counter, float eps, float momentum, float* running_mean,
                                              float* running_var, float* mean, float* rstd, hpcs_stream_t stream) {
    int rc = k2_args_ok(E, W2, ldw);
    if (rc) return rc;
    rc = fin_args_ok(sums, counter, mean, rstd, (int64_t)B * L);
    if (rc) return rc;
    if (bad_dims(B, Cout, L) || Ctot < Cout || B == 0) return HPCS_ERR_SHAPE;
    PdRef pd{E, nullptr, Ctot, L, 0, 0};
    pd.w2 = W2;
    pd.ldw = ldw;
    return vn_norm_stats_impl(pd, B, Cout, sums, 1, ST,
                              bn_fin(counter, (int64_t)B * L, eps, momentum, running_mean, running_var, mean, rstd));
}

extern "C" int hpcs_bn_stats_finalize(const float* Y, int B, int C, int64_t L, double* sums, uint32_t* counter, float eps,
                                      float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                                      hpcs_stream_t stream) {
    if (!Y) return HPCS_ERR_NULL;
    int rc = fin_args_ok(sums, counter, mean, rstd, (int64_t)B * L);
    if (rc) return rc;
    if (bad_dims(B, C, L) || C > 65535 || B == 0) return HPCS_ERR_SHAPE;
    if (bn_vec_ok(B, L, Y, nullptr, nullptr, nullptr)) {
        const int bpw = bn_bpw(B, C);
        hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(hpcs_cdiv(B, bpw), C), dim3(TPB), 0, ST, Y, B, C, L, bpw, sums,
                           bn_fin(counter, (int64_t)B * L, eps, momentum, running_mean, running_var, mean, rstd));
    } else
    hipLaunchKernelGGL(bn_stats_kernel, dim3(hpcs_cdiv((int64_t)B * L, TPB * ITEMS), C), dim3(TPB), 0, ST, Y, B, C, L, sums,
                       bn_fin(counter, (int64_t)B * L, eps, momentum, running_mean, running_var, mean, rstd));
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
