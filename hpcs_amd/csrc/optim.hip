// next row f1: hpcs.optim.RAdam.step (hpcs/optim/radam.py:59-134) as ONE launch over a table of tensors.
//
// The reference walks ~50 parameter tensors in a Python loop with ~25 small torch kernels each.  Every tensor is
// treated as a set of "points" = rows over its last dim (conv weights [Co,Ci,1] are Co*Ci one-dimensional points,
// BatchNorm vectors and `scale` are a single point).  8 lanes cooperate on one row; per row:
//   g += wd*p ; rg = g / lambda_p^2 (egrad2rgrad) ; m = b1 m + (1-b1) rg ; v = b2 v + (1-b2) lambda_p^2 <rg,rg>
//   u = -step * m / (sqrt(v)+eps) ; p' = project(expmap(u, p)) ; m = ptransp(p, p', m) ; p = p'
// step sizes (with the reference's double increment of group["step"] per parameter) are computed on the host.
#include "common.h"

namespace {
constexpr int SUB = 8;
constexpr int TPB = 256;
constexpr float MIN_NORM = 1e-15f;
constexpr float BALL_EPS_F32 = 4e-3f;

template <int SUBW>
__device__ __forceinline__ float sub_sum(float v) {
    if (SUBW == 1) return v;
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (SUBW == 64) {
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
    }
    return v;
}

// row_start[i] = first global row of tensor i (n_tensors+1 entries) lives right after the descriptors
// SUBW lanes per row: 8 for vector points; 1 for tables of one-dimensional points (Conv1d weights [Co,Ci,1] are
// Co*Ci such rows -- with 8 lanes each, 7 of 8 lanes idled through five passes).  Same arithmetic either way.
template <int SUBW>
__global__ void radam_kernel(const hpcs_radam_desc* __restrict__ desc, const int64_t* __restrict__ row_start, int n_tensors,
                             int64_t total_rows, float b1, float b2, float eps, float wd) {
    const int64_t grow = (int64_t)blockIdx.x * (TPB / SUBW) + threadIdx.x / SUBW;
    const int s = threadIdx.x % SUBW;
    if (grow >= total_rows) return;
    int lo = 0, hi = n_tensors;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row_start[mid] <= grow) lo = mid; else hi = mid;
    }
    const hpcs_radam_desc d = desc[lo];
    const int cols = d.cols;
    const int64_t base = (grow - row_start[lo]) * cols;
    float* P = (float*)d.param + base;
    float* G = (float*)d.grad + base;
    float* Mo = (float*)d.exp_avg + base;
    float* V = (float*)d.exp_avg_sq + base;

    // pass 1: |p|^2 and (with the decayed gradient) <g,g>
    float pp = 0.f, gg = 0.f;
    for (int i = s; i < cols; i += SUBW) {
        const float p = P[i];
        const float g = G[i] + wd * p;
        pp += p * p;
        gg += g * g;
    }
    pp = sub_sum<SUBW>(pp);
    gg = sub_sum<SUBW>(gg);
    const float lam_p = 2.f / fmaxf(1.f - pp, MIN_NORM);
    const float inv_l2 = 1.f / (lam_p * lam_p);
    const float inner = lam_p * lam_p * (gg * inv_l2 * inv_l2);        // lambda^2 <rg,rg>

    // pass 2: moments, step direction u; reductions for the exp-map
    float uu = 0.f, up = 0.f;
    for (int i = s; i < cols; i += SUBW) {
        const float p = P[i];
        const float rg = (G[i] + wd * p) * inv_l2;
        G[i] = rg;                                   // the reference mutates .grad in place too (radam.py:104-105)
        const float m = b1 * Mo[i] + (1.f - b1) * rg;
        const float v = b2 * V[i] + (1.f - b2) * inner;
        Mo[i] = m;
        V[i] = v;
        const float u = -d.step_size * (m / (sqrtf(v) + eps));
        uu += u * u;
        up += u * p;
    }
    uu = sub_sum<SUBW>(uu);
    up = sub_sum<SUBW>(up);
    // expmap(u, p) = mobius_add(p, y), y = tanh(lam_p |u| / 2) u / |u|
    const float un = fmaxf(sqrtf(uu), MIN_NORM);
    const float ysc = tanhf(fminf(fmaxf(lam_p * un / 2.f, -15.f), 15.f)) / un;
    const float y2 = ysc * ysc * uu, xy = ysc * up, x2 = pp;
    const float cx = 1.f + 2.f * xy + y2, cy = 1.f - x2;
    const float den = fmaxf(1.f + 2.f * xy + x2 * y2, MIN_NORM);

    // pass 3: |q|^2 of the un-projected new point -> projection factor; dots for the gyration
    float qq = 0.f;
    for (int i = s; i < cols; i += SUBW) {
        const float p = P[i];
        const float u = -d.step_size * (Mo[i] / (sqrtf(V[i]) + eps));
        const float q = (cx * p + cy * (ysc * u)) / den;
        qq += q * q;
    }
    qq = sub_sum<SUBW>(qq);
    const float qn = fmaxf(sqrtf(qq), MIN_NORM);
    const float maxn = 1.f - BALL_EPS_F32;
    const bool clip = qn > maxn;
    // new point np = clip ? q/qn*maxn : q ; gyration(u_=np, v_=-p, w=m)
    float u2 = 0.f, uv = 0.f, uw = 0.f, vw = 0.f;
    for (int i = s; i < cols; i += SUBW) {
        const float p = P[i];
        const float m = Mo[i];
        const float u = -d.step_size * (m / (sqrtf(V[i]) + eps));
        float q = (cx * p + cy * (ysc * u)) / den;
        if (clip) q = q / qn * maxn;
        u2 += q * q;
        uv += q * (-p);
        uw += q * m;
        vw += (-p) * m;
    }
    u2 = sub_sum<SUBW>(u2); uv = sub_sum<SUBW>(uv); uw = sub_sum<SUBW>(uw); vw = sub_sum<SUBW>(vw);
    const float v2 = pp;
    const float ga = -uw * v2 + vw + 2.f * uv * vw;
    const float gb = -vw * u2 - uw;
    const float gd = fmaxf(1.f + 2.f * uv + u2 * v2, MIN_NORM);
    const float lam_n = 2.f / fmaxf(1.f - u2, MIN_NORM);
    const float ratio = lam_p / lam_n;
    for (int i = s; i < cols; i += SUBW) {
        const float p = P[i];
        const float m = Mo[i];
        const float u = -d.step_size * (m / (sqrtf(V[i]) + eps));
        float q = (cx * p + cy * (ysc * u)) / den;
        if (clip) q = q / qn * maxn;
        const float gy = m + 2.f * (ga * q + gb * (-p)) / gd;
        Mo[i] = gy * ratio;
        P[i] = q;
    }
}
}  // namespace

extern "C" int hpcs_radam_step(const hpcs_radam_desc* desc_dev, int n_tensors, int64_t total_rows_hint, int lanes_per_row,
                               float beta1, float beta2, float eps, float weight_decay, hpcs_stream_t stream) {
    if (!desc_dev) return HPCS_ERR_NULL;
    if (n_tensors <= 0 || total_rows_hint <= 0) return HPCS_ERR_SHAPE;
    if (lanes_per_row != 1 && lanes_per_row != SUB && lanes_per_row != 64) return HPCS_ERR_UNSUPPORTED;
    // layout of the table (built by the host side): n_tensors descriptors, then (n_tensors+1) int64 row starts
    const int64_t* row_start = (const int64_t*)(desc_dev + n_tensors);
    if (lanes_per_row == 1) {
        const int64_t blocks = (total_rows_hint + TPB - 1) / TPB;
        hipLaunchKernelGGL(radam_kernel<1>, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, desc_dev, row_start,
                           n_tensors, total_rows_hint, beta1, beta2, eps, weight_decay);
    } else if (lanes_per_row == 64) {        // wide rows (hundreds of columns): a wave per row
        const int64_t blocks = (total_rows_hint + (TPB / 64) - 1) / (TPB / 64);
        hipLaunchKernelGGL(radam_kernel<64>, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, desc_dev, row_start,
                           n_tensors, total_rows_hint, beta1, beta2, eps, weight_decay);
    } else {
        const int64_t blocks = (total_rows_hint + (TPB / SUB) - 1) / (TPB / SUB);
        hipLaunchKernelGGL(radam_kernel<SUB>, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)stream, desc_dev, row_start,
                           n_tensors, total_rows_hint, beta1, beta2, eps, weight_decay);
    }
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
