// a7, a11, a15 + Poincare toolbox: row-wise kernels over [rows, E].
//
// Replaces expmap_1 / lambda_ / mobius_add / project / hyp_dist_o (hpcs/utils/poincare.py:16-136,
// hpcs/distances/poincare.py:13-136), Artanh (hpcs/utils/math.py:61-78: clamp +-(1-1e-5), fp64 inside),
// hyp_lca (hpcs/distances/lca.py:8-52) and HyperbolicDistance.compute_mat (hpcs/distances/poincare.py:144-176).
//
// Mapping: 8 lanes cooperate on one row (lane s handles elements s, s+8, ...; a 32-wide row is one 128-byte
// segment per load), row reductions are three xor-shuffles.  These tensors are a few MB: the kernels are
// launch/latency-scale, so the code favours exactness (fp64 for artanh and the LCA scalar algebra).
//
// hyp_lca: the reference's reflection construction is cancellation-bound in fp32 (2% median error at
// scale 1e-3); we evaluate the same quantity in closed form: the geodesic through a,b is a
// circle orthogonal to the unit sphere with centre c, c.a=(1+|a|^2)/2, c.b=(1+|b|^2)/2, and the projection of
// the origin on it has norm m = |c| - sqrt(|c|^2-1), |c|^2 = |(1+aa)b-(1+bb)a|^2 / (4(aa*bb-ab^2)).
#include "common.h"

namespace {

constexpr float MIN_NORM = 1e-15f;
constexpr float BALL_EPS_F32 = 4e-3f;
constexpr int SUB = 8;            // lanes per row
constexpr int TPB = 256;
constexpr int ROWS_PER_BLOCK = TPB / SUB;

__device__ __forceinline__ float sub_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__device__ __forceinline__ double artanh_clamped(double x, double* dx_out) {
    const double lim = 1.0 - 1e-5;
    // the reference clamps the fp32 input, then evaluates in fp64
    float xf = (float)x;
    xf = fminf(fmaxf(xf, -1.f + 1e-5f), 1.f - 1e-5f);
    (void)lim;
    const double xc = (double)xf;
    if (dx_out) *dx_out = 1.0 / (1.0 - xc * xc);
    return 0.5 * (log1p(xc) - log1p(-xc));
}

#define ROW_PROLOGUE()                                                            \
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x / SUB); \
    const int s = threadIdx.x % SUB;                                              \
    const bool live = row < rows;                                                 \
    const int64_t base = (live ? row : 0) * E

__global__ void expmap0_fwd_kernel(const float* __restrict__ u, int64_t rows, int E, float* __restrict__ out) {
    ROW_PROLOGUE();
    float ss = 0.f;
    for (int i = s; i < E; i += SUB) { const float v = u[base + i]; ss += v * v; }
    ss = sub_sum(ss);
    const float un = fmaxf(sqrtf(ss), MIN_NORM);
    const float sc = tanhf(fminf(un, 15.f)) / un;
    if (live)
        for (int i = s; i < E; i += SUB) out[base + i] = u[base + i] * sc;
}

__global__ void expmap0_bwd_kernel(const float* __restrict__ u, const float* __restrict__ g, int64_t rows, int E,
                                   float* __restrict__ gu) {
    ROW_PROLOGUE();
    float ss = 0.f, ug = 0.f;
    for (int i = s; i < E; i += SUB) { const float v = u[base + i]; ss += v * v; ug += v * g[base + i]; }
    ss = sub_sum(ss);
    ug = sub_sum(ug);
    const float r = sqrtf(ss);
    const float un = fmaxf(r, MIN_NORM);
    const float t = tanhf(fminf(un, 15.f));
    const float sc = t / un;
    float coef = 0.f;
    if (r > MIN_NORM) {
        const float dt = un < 15.f ? (1.f - t * t) : 0.f;
        coef = (dt * un - t) / (un * un) * ug / r;
    }
    if (live)
        for (int i = s; i < E; i += SUB) gu[base + i] = sc * g[base + i] + coef * u[base + i];
}

// general exp-map at p: mobius_add(p, tanh(lambda_p*|u|/2) u/|u|)
__global__ void expmap_fwd_kernel(const float* __restrict__ u, const float* __restrict__ p, int64_t rows, int E,
                                  float* __restrict__ out) {
    ROW_PROLOGUE();
    float uu = 0.f, pp = 0.f, up = 0.f;
    for (int i = s; i < E; i += SUB) {
        const float a = u[base + i], b = p[base + i];
        uu += a * a; pp += b * b; up += a * b;
    }
    uu = sub_sum(uu); pp = sub_sum(pp); up = sub_sum(up);
    const float un = fmaxf(sqrtf(uu), MIN_NORM);
    const float lam = 2.f / fmaxf(1.f - pp, MIN_NORM);
    const float sc = tanhf(fminf(fmaxf(lam * un / 2.f, -15.f), 15.f)) / un;      // y = sc*u
    const float y2 = sc * sc * uu, xy = sc * up, x2 = pp;
    const float cx = 1.f + 2.f * xy + y2, cy = 1.f - x2;
    const float den = fmaxf(1.f + 2.f * xy + x2 * y2, MIN_NORM);
    if (live)
        for (int i = s; i < E; i += SUB) out[base + i] = (cx * p[base + i] + cy * (sc * u[base + i])) / den;
}

__global__ void project_kernel(const float* __restrict__ x, int64_t rows, int E, float* __restrict__ out) {
    ROW_PROLOGUE();
    float ss = 0.f;
    for (int i = s; i < E; i += SUB) { const float v = x[base + i]; ss += v * v; }
    ss = sub_sum(ss);
    const float n = fmaxf(sqrtf(ss), MIN_NORM);
    const float maxn = 1.f - BALL_EPS_F32;
    if (live)
        for (int i = s; i < E; i += SUB) {
            const float v = x[base + i];
            out[base + i] = n > maxn ? v / n * maxn : v;
        }
}

__global__ void mobius_add_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t rows, int E,
                                  float* __restrict__ out) {
    ROW_PROLOGUE();
    float x2 = 0.f, y2 = 0.f, xy = 0.f;
    for (int i = s; i < E; i += SUB) {
        const float a = x[base + i], b = y[base + i];
        x2 += a * a; y2 += b * b; xy += a * b;
    }
    x2 = sub_sum(x2); y2 = sub_sum(y2); xy = sub_sum(xy);
    const float cx = 1.f + 2.f * xy + y2, cy = 1.f - x2;
    const float den = fmaxf(1.f + 2.f * xy + x2 * y2, MIN_NORM);
    if (live)
        for (int i = s; i < E; i += SUB) out[base + i] = (cx * x[base + i] + cy * y[base + i]) / den;
}

__global__ void hyp_dist_o_kernel(const float* __restrict__ x, int64_t rows, int E, float* __restrict__ out) {
    ROW_PROLOGUE();
    float ss = 0.f;
    for (int i = s; i < E; i += SUB) { const float v = x[base + i]; ss += v * v; }
    ss = sub_sum(ss);
    if (live && s == 0) out[row] = (float)(2.0 * artanh_clamped((double)sqrtf(ss), nullptr));
}

// ---- hyp_lca, general position ---------------------------------------------------------------------------
struct LcaScalars { double aa, ab, bb, N, det, C2, m, d, dd_dm; };

__device__ __forceinline__ LcaScalars lca_scalars(double aa, double ab, double bb) {
    LcaScalars q;
    q.aa = aa; q.ab = ab; q.bb = bb;
    q.N = bb * (1 + aa) * (1 + aa) + aa * (1 + bb) * (1 + bb) - 2 * ab * (1 + aa) * (1 + bb);
    q.det = aa * bb - ab * ab;
    if (q.det <= 0.0 || q.N <= 0.0) {        // a, b, origin collinear: the geodesic is a diameter
        q.C2 = INFINITY; q.m = 0.0; q.d = 0.0; q.dd_dm = 2.0;
        return q;
    }
    q.C2 = fmax(q.N / (4.0 * q.det), 1.0);
    const double C = sqrt(q.C2);
    q.m = 1.0 / (C + sqrt(q.C2 - 1.0));
    double dx;
    q.d = 2.0 * artanh_clamped(q.m, &dx);
    q.dd_dm = 2.0 * dx;
    return q;
}

__global__ void hyp_lca_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t rows, int E,
                                   float* __restrict__ dist, float* __restrict__ coord) {
    ROW_PROLOGUE();
    float aa = 0.f, ab = 0.f, bb = 0.f;
    for (int i = s; i < E; i += SUB) {
        const float x = a[base + i], y = b[base + i];
        aa += x * x; ab += x * y; bb += y * y;
    }
    aa = sub_sum(aa); ab = sub_sum(ab); bb = sub_sum(bb);
    const LcaScalars q = lca_scalars(aa, ab, bb);
    if (!live) return;
    if (s == 0 && dist) dist[row] = (float)q.d;
    if (coord) {
        // centre c = al*a + be*b ; proj = m * c/|c|
        double al = 0.0, be = 0.0, cn = 1.0;
        if (q.m > 0.0) {
            al = (q.bb * (1 + q.aa) - q.ab * (1 + q.bb)) / (2.0 * q.det);
            be = (q.aa * (1 + q.bb) - q.ab * (1 + q.aa)) / (2.0 * q.det);
            cn = sqrt(q.C2);
        }
        const double f = q.m / cn;
        for (int i = s; i < E; i += SUB) coord[base + i] = (float)(f * (al * a[base + i] + be * b[base + i]));
    }
}

__global__ void hyp_lca_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gdist,
                                   int64_t rows, int E, float* __restrict__ ga, float* __restrict__ gb) {
    ROW_PROLOGUE();
    float aaf = 0.f, abf = 0.f, bbf = 0.f;
    for (int i = s; i < E; i += SUB) {
        const float x = a[base + i], y = b[base + i];
        aaf += x * x; abf += x * y; bbf += y * y;
    }
    aaf = sub_sum(aaf); abf = sub_sum(abf); bbf = sub_sum(bbf);
    const LcaScalars q = lca_scalars(aaf, abf, bbf);
    if (!live) return;
    double Faa = 0.0, Fab = 0.0, Fbb = 0.0;
    if (q.m > 0.0 && q.C2 > 1.0) {
        const double aa = q.aa, ab = q.ab, bb = q.bb;
        // d = d(m), m = C - sqrt(C^2-1), C = sqrt(C2), C2 = N/(4 det)
        const double C = sqrt(q.C2);
        const double dm_dC = -q.m / sqrt(q.C2 - 1.0);
        const double dC_dC2 = 0.5 / C;
        const double g = (double)gdist[row] * q.dd_dm * dm_dC * dC_dC2;       // dL/dC2
        const double N_aa = 2 * bb * (1 + aa) + (1 + bb) * (1 + bb) - 2 * ab * (1 + bb);
        const double N_bb = (1 + aa) * (1 + aa) + 2 * aa * (1 + bb) - 2 * ab * (1 + aa);
        const double N_ab = -2 * (1 + aa) * (1 + bb);
        const double inv4det = 1.0 / (4.0 * q.det);
        Faa = g * (N_aa * inv4det - q.C2 / q.det * bb);
        Fbb = g * (N_bb * inv4det - q.C2 / q.det * aa);
        Fab = g * (N_ab * inv4det + q.C2 / q.det * 2 * ab);
    }
    for (int i = s; i < E; i += SUB) {
        const double x = a[base + i], y = b[base + i];
        ga[base + i] = (float)(2 * Faa * x + Fab * y);
        gb[base + i] = (float)(2 * Fbb * y + Fab * x);
    }
}

// Backward of the COORDINATE output (hyp_lca(a, b, return_coord=True), hpcs/distances/lca.py:37-52 is differentiable in the
// reference).  p = F(q) c with c = al*a + be*b the centre of the geodesic's circle (G [al; be] = h, G the Gram matrix of
// (a, b), h = ((1+aa)/2, (1+bb)/2)), q = |c|^2, F(q) = 1 - sqrt(1 - 1/q).  With g = dL/dp:
//   dL/dc = F g + 2 F'(q) (g.c) c,  dL/dal = dL/dc . a,  dL/dbe = dL/dc . b   (c.a = h1, c.b = h2 by construction),
//   lam = G^-1 (dL/dal, dL/dbe)   (adjoint of the 2x2 solve: dL/dh = lam, dL/dG = -lam [al be]),
//   dL/da = al dL/dc + lam1 (1 - 2 al) a - (lam1 be + lam2 al) b,   dL/db symmetric.   fp64 scalars, like the forward.
__global__ void hyp_lca_coord_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                         const float* __restrict__ gcoord, int64_t rows, int E, float* __restrict__ ga,
                                         float* __restrict__ gb) {
    ROW_PROLOGUE();
    float aaf = 0.f, abf = 0.f, bbf = 0.f, gaf = 0.f, gbf = 0.f;
    for (int i = s; i < E; i += SUB) {
        const float x = a[base + i], y = b[base + i], g = gcoord[base + i];
        aaf += x * x; abf += x * y; bbf += y * y; gaf += g * x; gbf += g * y;
    }
    aaf = sub_sum(aaf); abf = sub_sum(abf); bbf = sub_sum(bbf); gaf = sub_sum(gaf); gbf = sub_sum(gbf);
    const LcaScalars q = lca_scalars(aaf, abf, bbf);
    if (!live) return;
    double F = 0.0, k_c = 0.0, al = 0.0, be = 0.0, lam1 = 0.0, lam2 = 0.0;
    if (q.m > 0.0 && q.C2 > 1.0) {
        al = (q.bb * (1 + q.aa) - q.ab * (1 + q.bb)) / (2.0 * q.det);
        be = (q.aa * (1 + q.bb) - q.ab * (1 + q.aa)) / (2.0 * q.det);
        const double root = sqrt(1.0 - 1.0 / q.C2);
        F = q.m / sqrt(q.C2);                                     // = 1 - root, without the cancellation
        const double dF = -1.0 / (2.0 * q.C2 * q.C2 * root);
        const double g_c = al * (double)gaf + be * (double)gbf;   // g . c
        k_c = 2.0 * dF * g_c;                                     // dL/dc = F g + k_c c
        const double d_al = F * (double)gaf + k_c * 0.5 * (1 + q.aa);
        const double d_be = F * (double)gbf + k_c * 0.5 * (1 + q.bb);
        lam1 = (q.bb * d_al - q.ab * d_be) / q.det;
        lam2 = (q.aa * d_be - q.ab * d_al) / q.det;
    }
    const double cross = lam1 * be + lam2 * al;
    for (int i = s; i < E; i += SUB) {
        const double x = a[base + i], y = b[base + i], g = gcoord[base + i];
        const double gc = F * g + k_c * (al * x + be * y);
        ga[base + i] = (float)(al * gc + lam1 * (1.0 - 2.0 * al) * x - cross * y);
        gb[base + i] = (float)(be * gc + lam2 * (1.0 - 2.0 * be) * y - cross * x);
    }
}

// ---- a15 / a8 dense matrices (API completeness; the training path never builds them) ----------------------
__device__ __forceinline__ float hyperbolic_sim_pair(const float* __restrict__ xr, const float* __restrict__ yr, int E, float eps) {
    const float maxn = 1.f - BALL_EPS_F32;
    float xx = 0.f, yy = 0.f;
    for (int e = 0; e < E; ++e) { const float a = xr[e], b = yr[e]; xx += a * a; yy += b * b; }
    const float nx = fmaxf(sqrtf(xx), MIN_NORM), ny = fmaxf(sqrtf(yy), MIN_NORM);
    const float sx = nx > maxn ? maxn / nx : 1.f, sy = ny > maxn ? maxn / ny : 1.f;
    float d2 = 0.f;
    for (int e = 0; e < E; ++e) { const float df = (xr[e] * sx - yr[e] * sy) + eps; d2 += df * df; }
    const float ax = 1.f - xx * sx * sx, ay = 1.f - yy * sy * sy;
    const float dxy = 1.f + 2.f * (d2 / (ax * ay));
    return expf(-acoshf(dxy));
}
__global__ void hyperbolic_sim_mat_kernel(const float* __restrict__ x, const float* __restrict__ y, int M, int Mp, int E,
                                          float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= Mp) return;
    out[(int64_t)i * Mp + j] = hyperbolic_sim_pair(x + (int64_t)i * E, y + (int64_t)j * E, E, 0.f);
}
// row-aligned pairs: HyperbolicDistance.pairwise_distance (hpcs/distances/poincare.py:178-187); its squared distance
// comes from F.pairwise_distance, which adds eps = 1e-6 to every component of the difference
__global__ void hyperbolic_sim_pairs_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t M, int E,
                                            float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    out[i] = hyperbolic_sim_pair(x + i * E, y + i * E, E, 1e-6f);
}

// HyperbolicLCA (hpcs/distances/poincare.py:190-237), E = 2 read as complex numbers.  For a pair (x, y) of projected
// points: w = M_x(y) = (y - x) / (1 - conj(x) y) sends x to the origin; the geodesic midpoint is mid = tanh(artanh|w| / 2)
// w / |w|; m = M_x^{-1}(mid) = (mid + x) / (1 + conj(x) mid); result = min(d(o,x), d(o,y), d(o,m)), d(o,p) = 2 artanh|p|
// with the reference's fp32 input clamp.  pairs != 0: row-aligned pairs (out [M]); else the full [M, Mp] matrix.
__device__ __forceinline__ void project2(const float* __restrict__ p, float& re, float& im) {
    const float a = p[0], b = p[1];
    const float n = fmaxf(sqrtf(a * a + b * b), MIN_NORM);
    const float maxn = 1.f - BALL_EPS_F32;
    const float sc = n > maxn ? maxn / n : 1.f;
    re = a * sc;
    im = b * sc;
}
__device__ __forceinline__ float lca_pair(float xr, float xi, float yr, float yi) {
    const float dox = (float)(2.0 * artanh_clamped((double)sqrtf(xr * xr + xi * xi), nullptr));
    const float doy = (float)(2.0 * artanh_clamped((double)sqrtf(yr * yr + yi * yi), nullptr));
    // w = (y - x) / (1 - conj(x) y)
    const float nr = yr - xr, ni = yi - xi;
    const float dr = 1.f - (xr * yr + xi * yi), di = -(xr * yi - xi * yr);
    const float dd = dr * dr + di * di;
    const float wr = (nr * dr + ni * di) / dd, wi = (ni * dr - nr * di) / dd;
    // midpoint with the origin: mobius_mul(w, 0.5)
    const float wn = fmaxf(sqrtf(wr * wr + wi * wi), MIN_NORM);
    const float half = (float)(0.5 * artanh_clamped((double)wn, nullptr));
    const float t = tanhf(fminf(fmaxf(half, -15.f), 15.f)) / wn;
    const float mr = t * wr, mi = t * wi;
    // m = (mid + x) / (1 + conj(x) mid)
    const float pr = mr + xr, pi = mi + xi;
    const float qr = 1.f + (xr * mr + xi * mi), qi = xr * mi - xi * mr;
    const float qq = qr * qr + qi * qi;
    const float zr = (pr * qr + pi * qi) / qq, zi = (pi * qr - pr * qi) / qq;
    const float dom = (float)(2.0 * artanh_clamped((double)sqrtf(zr * zr + zi * zi), nullptr));
    return fminf(fminf(dox, doy), dom);
}
__global__ void hyperbolic_lca_kernel(const float* __restrict__ x, const float* __restrict__ y, int M, int Mp, int pairs,
                                      float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = pairs ? j : blockIdx.y;
    if (j >= (pairs ? M : Mp)) return;
    float xr, xi, yr, yi;
    project2(x + 2 * (int64_t)i, xr, xi);
    project2(y + 2 * (int64_t)j, yr, yi);
    out[pairs ? (int64_t)j : (int64_t)i * Mp + j] = lca_pair(xr, xi, yr, yi);
}

__global__ void cosine_mat_kernel(const float* __restrict__ xn, const float* __restrict__ yn, int M, int Mp, int E,
                                  float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= Mp) return;
    float d = 0.f;
    for (int e = 0; e < E; ++e) d = __builtin_fmaf(xn[(int64_t)i * E + e], yn[(int64_t)j * E + e], d);
    out[(int64_t)i * Mp + j] = 0.5f * (1.f + d);
}

// ---- l2 row normalisation (PML BaseDistance.maybe_normalize == F.normalize(p=2, dim=1, eps=1e-12)) --------
__global__ void l2_rows_fwd_kernel(const float* __restrict__ x, int64_t rows, int E, float* __restrict__ xn,
                                   float* __restrict__ inv_norm) {
    ROW_PROLOGUE();
    float ss = 0.f;
    for (int i = s; i < E; i += SUB) { const float v = x[base + i]; ss += v * v; }
    ss = sub_sum(ss);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    if (!live) return;
    for (int i = s; i < E; i += SUB) xn[base + i] = x[base + i] * inv;
    if (s == 0 && inv_norm) inv_norm[row] = inv;
}

__global__ void l2_rows_bwd_kernel(const float* __restrict__ xn, const float* __restrict__ inv_norm,
                                   const float* __restrict__ gxn, int64_t rows, int E, float* __restrict__ gx, int accumulate) {
    ROW_PROLOGUE();
    float dot = 0.f;
    for (int i = s; i < E; i += SUB) dot += xn[base + i] * gxn[base + i];
    dot = sub_sum(dot);
    if (!live) return;
    const float inv = inv_norm[row];
    for (int i = s; i < E; i += SUB) {
        const float v = (gxn[base + i] - xn[base + i] * dot) * inv;
        gx[base + i] = accumulate ? gx[base + i] + v : v;
    }
}

__global__ void colsum_kernel(const float* __restrict__ x, int64_t rows, int E, double* __restrict__ colsum) {
    // grid.x = E columns chunks of 64 lanes; each block reduces a slab of rows for 64 columns
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * 1024;
    const int64_t r1 = r0 + 1024 < rows ? r0 + 1024 : rows;
    double acc = 0.0;
    if (e < E)
        for (int64_t r = r0 + w; r < r1; r += nw) acc += (double)x[r * E + e];
    __shared__ double sh[4][64];
    sh[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && e < E) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += sh[i][threadIdx.x & 63];
        atomicAdd(colsum + e, t);
    }
}

// E a power of two <= 256: thread t owns column t % E and walks rows t / E, t / E + 256 / E, ... of a 256-row slab: every
// wave instruction reads 256 contiguous bytes (the kernel above leaves half the lanes idle at E = 32 and walks 256
// dependent loads per wave: 49 us at M = 32768 against 5 us here).
__global__ __launch_bounds__(256) void colsum_pow2_kernel(const float* __restrict__ x, int64_t rows, int E,
                                                        double* __restrict__ colsum) {
    __shared__ double sh[256];
    const int e = threadIdx.x % E, rsub = threadIdx.x / E, rpi = 256 / E;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int64_t r1 = r0 + 256 < rows ? r0 + 256 : rows;
    double acc = 0.0;
    for (int64_t r = r0 + rsub; r < r1; r += rpi) acc += (double)x[r * E + e];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < E) {
        double t = 0.0;
        for (int i = 0; i < rpi; ++i) t += sh[i * E + threadIdx.x];
        atomicAdd(colsum + threadIdx.x, t);
    }
}

// HypHC term, the scalar tail of the forward pass in one launch (was nine torch element-wise kernels):
//   out = loss_sum / kept + (0.5 + 0.5 * |colsum|^2 / M^2)   (ultrametric_loss.py: HypHCSampledFn.forward)
__global__ void hyphc_finalize_kernel(const double* __restrict__ loss_sum, const int64_t* __restrict__ kept,
                                      const double* __restrict__ colsum, int64_t M, int E, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int e = 0; e < E; ++e) s += colsum[e] * colsum[e];
    const double mean_sim = 0.5 + 0.5 * s / ((double)M * (double)M);
    out[0] = (float)(loss_sum[0] / (double)kept[0] + mean_sim);
}
// ... and of the backward pass (ten): gxn = colsum / M^2 * g + gxn_raw * inv in place, gscale = gscale_raw * inv,
// inv = g / max(kept, 1)
__global__ __launch_bounds__(256) void hyphc_grad_finalize_kernel(const float* __restrict__ g, const int64_t* __restrict__ kept,
                                                                 const double* __restrict__ colsum,
                                                                 const double* __restrict__ gscale_raw, int64_t M, int E,
                                                                 float* __restrict__ gxn, float* __restrict__ gscale) {
    const float g32 = g[0];
    const double k = (double)kept[0];
    const float inv = (float)((double)g32 / (k < 1.0 ? 1.0 : k));
    const int64_t total = M * E;
    const double mm = (double)M * (double)M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float row = (float)(colsum[i % E] / mm) * g32;
        gxn[i] = row + gxn[i] * inv;
    }
    if (gscale && blockIdx.x == 0 && threadIdx.x == 0) gscale[0] = (float)(gscale_raw[0] * (double)inv);
}

}  // namespace

#define ST ((hipStream_t)stream)
#define ROW_GRID(rows) dim3((unsigned)hpcs_cdiv((rows), ROWS_PER_BLOCK)), dim3(TPB), 0, ST
#define ROW_ARGS_OK(rows, E) ((rows) >= 0 && (E) > 0)

extern "C" int hpcs_expmap0_fwd(const float* u, int64_t rows, int E, float* out, hpcs_stream_t stream) {
    if (!u || !out) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(expmap0_fwd_kernel, ROW_GRID(rows), u, rows, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_expmap0_bwd(const float* u, const float* gout, int64_t rows, int E, float* gu, hpcs_stream_t stream) {
    if (!u || !gout || !gu) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(expmap0_bwd_kernel, ROW_GRID(rows), u, gout, rows, E, gu);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_expmap_fwd(const float* u, const float* p, int64_t rows, int E, float* out, hpcs_stream_t stream) {
    if (!u || !p || !out) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(expmap_fwd_kernel, ROW_GRID(rows), u, p, rows, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_project(const float* x, int64_t rows, int E, float* out, hpcs_stream_t stream) {
    if (!x || !out) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(project_kernel, ROW_GRID(rows), x, rows, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_mobius_add(const float* x, const float* y, int64_t rows, int E, float* out, hpcs_stream_t stream) {
    if (!x || !y || !out) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(mobius_add_kernel, ROW_GRID(rows), x, y, rows, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_hyp_dist_o(const float* x, int64_t rows, int E, float* out, hpcs_stream_t stream) {
    if (!x || !out) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(hyp_dist_o_kernel, ROW_GRID(rows), x, rows, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_hyp_lca_fwd(const float* a, const float* b, int64_t rows, int E, float* dist, float* coord,
                                hpcs_stream_t stream) {
    if (!a || !b || (!dist && !coord)) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(hyp_lca_fwd_kernel, ROW_GRID(rows), a, b, rows, E, dist, coord);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_hyp_lca_bwd(const float* a, const float* b, const float* gdist, int64_t rows, int E, float* ga,
                                float* gb, hpcs_stream_t stream) {
    if (!a || !b || !gdist || !ga || !gb) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(hyp_lca_bwd_kernel, ROW_GRID(rows), a, b, gdist, rows, E, ga, gb);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hyp_lca_coord_bwd(const float* a, const float* b, const float* gcoord, int64_t rows, int E, float* ga,
                                      float* gb, hpcs_stream_t stream) {
    if (!a || !b || !gcoord || !ga || !gb) return HPCS_ERR_NULL;
    if (rows < 0 || E <= 0) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(hyp_lca_coord_bwd_kernel, ROW_GRID(rows), a, b, gcoord, rows, E, ga, gb);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_hyperbolic_sim_mat(const float* x, const float* y, int M, int Mp, int E, float* out,
                                       hpcs_stream_t stream) {
    if (!x || !y || !out) return HPCS_ERR_NULL;
    if (M < 0 || Mp < 0 || E <= 0 || M > 65535) return HPCS_ERR_SHAPE;
    if (M == 0 || Mp == 0) return HPCS_OK;
    hipLaunchKernelGGL(hyperbolic_sim_mat_kernel, dim3(hpcs_cdiv(Mp, 256), M), dim3(256), 0, ST, x, y, M, Mp, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_hyperbolic_sim_pairs(const float* x, const float* y, int64_t M, int E, float* out, hpcs_stream_t stream) {
    if (!x || !y || !out) return HPCS_ERR_NULL;
    if (M < 0 || E <= 0) return HPCS_ERR_SHAPE;
    if (M == 0) return HPCS_OK;
    hipLaunchKernelGGL(hyperbolic_sim_pairs_kernel, dim3(hpcs_cdiv(M, 256)), dim3(256), 0, ST, x, y, M, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_hyperbolic_lca(const float* x, const float* y, int M, int Mp, int E, int pairs, float* out,
                                   hpcs_stream_t stream) {
    if (!x || !y || !out) return HPCS_ERR_NULL;
    if (M < 0 || Mp < 0 || M > 65535 || (pairs && M != Mp)) return HPCS_ERR_SHAPE;
    if (E != 2) return HPCS_ERR_UNSUPPORTED;               // the reference views the points as complex numbers
    if (M == 0 || Mp == 0) return HPCS_OK;
    if (pairs)
        hipLaunchKernelGGL(hyperbolic_lca_kernel, dim3(hpcs_cdiv(M, 256), 1), dim3(256), 0, ST, x, y, M, Mp, 1, out);
    else
        hipLaunchKernelGGL(hyperbolic_lca_kernel, dim3(hpcs_cdiv(Mp, 256), M), dim3(256), 0, ST, x, y, M, Mp, 0, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_cosine_mat(const float* xn, const float* yn, int M, int Mp, int E, float* out, hpcs_stream_t stream) {
    if (!xn || !yn || !out) return HPCS_ERR_NULL;
    if (M < 0 || Mp < 0 || E <= 0 || M > 65535) return HPCS_ERR_SHAPE;
    if (M == 0 || Mp == 0) return HPCS_OK;
    hipLaunchKernelGGL(cosine_mat_kernel, dim3(hpcs_cdiv(Mp, 256), M), dim3(256), 0, ST, xn, yn, M, Mp, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_l2_rows_fwd(const float* x, int64_t rows, int E, float* xn, float* inv_norm, hpcs_stream_t stream) {
    if (!x || !xn) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(l2_rows_fwd_kernel, ROW_GRID(rows), x, rows, E, xn, inv_norm);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_l2_rows_bwd(const float* xn, const float* inv_norm, const float* gxn, int64_t rows, int E, float* gx,
                                int accumulate, hpcs_stream_t stream) {
    if (!xn || !inv_norm || !gxn || !gx) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (rows == 0) return HPCS_OK;
    hipLaunchKernelGGL(l2_rows_bwd_kernel, ROW_GRID(rows), xn, inv_norm, gxn, rows, E, gx, accumulate);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
extern "C" int hpcs_colsum(const float* xn, int64_t rows, int E, double* colsum, hpcs_stream_t stream) {
    if (!xn || !colsum) return HPCS_ERR_NULL;
    if (!ROW_ARGS_OK(rows, E)) return HPCS_ERR_SHAPE;
    if (const int zr = hpcs_zero_async(ST, colsum, sizeof(double) * E)) return zr;
    if (rows == 0) return HPCS_OK;
    if (E <= 256 && (E & (E - 1)) == 0)
        hipLaunchKernelGGL(colsum_pow2_kernel, dim3(hpcs_cdiv(rows, 256)), dim3(256), 0, ST, xn, rows, E, colsum);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3(hpcs_cdiv(E, 64), hpcs_cdiv(rows, 1024)), dim3(256), 0, ST, xn, rows, E, colsum);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hyphc_finalize(const double* loss_sum, const int64_t* kept, const double* colsum, int64_t M, int E, float* out,
                                   hpcs_stream_t stream) {
    if (!loss_sum || !kept || !colsum || !out) return HPCS_ERR_NULL;
    if (M <= 0 || E <= 0) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(hyphc_finalize_kernel, dim3(1), dim3(64), 0, ST, loss_sum, kept, colsum, M, E, out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_hyphc_grad_finalize(const float* g, const int64_t* kept, const double* colsum, const double* gscale_raw,
                                        int64_t M, int E, float* gxn, float* gscale, hpcs_stream_t stream) {
    if (!g || !kept || !colsum || !gxn) return HPCS_ERR_NULL;
    if (gscale && !gscale_raw) return HPCS_ERR_NULL;
    if (M <= 0 || E <= 0) return HPCS_ERR_SHAPE;
    const int64_t total = M * E;
    const unsigned blocks = (unsigned)(total / 1024 < 1 ? 1 : (total / 1024 > 4096 ? 4096 : total / 1024));
    hipLaunchKernelGGL(hyphc_grad_finalize_kernel, dim3(blocks), dim3(256), 0, ST, g, kept, colsum, gscale_raw, M, E, gxn, gscale);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
