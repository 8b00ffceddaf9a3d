// Point-wise Vector-Neuron arithmetic shared by csrc/vn_layers.hip (streaming kernels over a materialised pre-activation
// tensor) and csrc/vn_recompute.hip (EdgeConv kernels that re-form the pre-activations on the matrix cores): VNBatchNorm
// (hpcs/nn/dgcnn/utils/vn_layers.py:112-132), the VN-LeakyReLU of VNLinearLeakyReLU (:66-77), their backward, and the
// in-kernel BatchNorm finalisation.  Included inside each file's anonymous namespace.
#pragma once

constexpr float VN_EPS = 1e-6f;

// a / b for b > 0 well inside the normal range (norms + eps, |d|^2 + eps) in the BACKWARD point functions: v_rcp_f32 and one
// Newton correction -- 4 instructions against the ~10 of the IEEE sequence (v_div_scale x2, v_rcp, 4-5 fma, v_div_fmas,
// v_div_fixup), and the same correctly rounded quotient on every one of 4.2 M random operand pairs (scripts/micro/div_ulp.hip).
// The backward executes eight of them per (channel, position).  The forward point functions keep '/': the quotient is
// identical, but the open-coded sequence changes how the compiler packs and contracts the surrounding sums, and the forward
// error ensemble (tests/test_gpu_round2.py) is pinned to the rounding pattern it was measured on.  -DHPCS_VN_IEEE_DIV
// restores '/' everywhere.
__device__ __forceinline__ float vn_div(float a, float b) {
#ifdef HPCS_VN_IEEE_DIV
    return a / b;
#else
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    const float e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
#endif
}

// ------------------------------------------------------------------------------------------------ a3
struct VnPoint {   // recomputed forward state of one (b,c,l) vector
    float p[3], d[3], n, xhat, nb, q[3], dot, dsq;
    bool keep;
};

// BWD: the recomputation inside the backward kernels -- p / n * nb as one quotient nb / n (rcp + Newton) and three products
// instead of three IEEE quotients (26 instructions fewer per point; q differs from the forward kernels' by an ulp, which only
// the backward's own arithmetic sees).  The forward kernels keep the reference's operation order.
template <bool BWD = false>
__device__ __forceinline__ void vn_forward_point(VnPoint& s, float mean, float rstd, float gamma, float beta, int flags) {
    const float pn = sqrtf(s.p[0] * s.p[0] + s.p[1] * s.p[1] + s.p[2] * s.p[2]);
    s.n = pn + VN_EPS;
    if (flags & HPCS_VN_BN) {
        s.xhat = (s.n - mean) * rstd;
        s.nb = s.xhat * gamma + beta;
        if (BWD) {
            const float sc = vn_div(s.nb, s.n);
#pragma unroll
            for (int a = 0; a < 3; ++a) s.q[a] = s.p[a] * sc;
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) s.q[a] = s.p[a] / s.n * s.nb;
        }
    } else {
        s.xhat = 0.f;
        s.nb = 1.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) s.q[a] = s.p[a];
    }
    s.dot = s.q[0] * s.d[0] + s.q[1] * s.d[1] + s.q[2] * s.d[2];
    s.dsq = s.d[0] * s.d[0] + s.d[1] * s.d[1] + s.d[2] * s.d[2];
    s.keep = !(flags & HPCS_VN_LRELU) || (s.dot >= 0.f);
}

__device__ __forceinline__ void vn_output_point(const VnPoint& s, float slope, int flags, float (&o)[3]) {
    if (!(flags & HPCS_VN_LRELU)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = s.q[a];
        return;
    }
    const float t = s.dot / (s.dsq + VN_EPS);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inner = s.keep ? s.q[a] : (s.q[a] - t * s.d[a]);
        o[a] = slope * s.q[a] + (1.f - slope) * inner;
    }
}

// gradient wrt q and d of one vector given g = dL/dout
__device__ __forceinline__ void vn_lrelu_bwd_point(const VnPoint& s, const float (&g)[3], float slope, int flags,
                                                   float (&gq)[3], float (&gd)[3]) {
    if (s.keep) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            gq[a] = g[a];
            gd[a] = 0.f;
        }
        return;
    }
    const float a1 = 1.f - slope;
    const float den = s.dsq + VN_EPS;
    const float t = vn_div(s.dot, den);
    const float gdot = g[0] * s.d[0] + g[1] * s.d[1] + g[2] * s.d[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gq[a] = g[a] - a1 * vn_div(gdot, den) * s.d[a];
        gd[a] = -a1 * (t * g[a] + vn_div(gdot * (s.q[a] - 2.f * t * s.d[a]), den));
    }
}

// gradient wrt p of one vector through the VN-BatchNorm (norm statistics), given gq = dL/dq.
// m1 = E[gnb] * gamma, m2 = E[gnb * xhat] * gamma from the backward reduction (train mode).
__device__ __forceinline__ void vn_bn_bwd_point(const VnPoint& s, const float (&gq)[3], bool bn, bool training, float m1,
                                                float m2, float ga, float rs, float (&gp)[3]) {
    if (!bn) {
#pragma unroll
        for (int a = 0; a < 3; ++a) gp[a] = gq[a];
        return;
    }
    const float gqp = gq[0] * s.p[0] + gq[1] * s.p[1] + gq[2] * s.p[2];
    const float gnb = vn_div(gqp, s.n);
    float gn = vn_div(-gqp * s.nb, s.n * s.n);
    const float gxh = gnb * ga;
    if (training) gn += rs * (gxh - m1 - s.xhat * m2);
    else gn += rs * gxh;
    const float pn = s.n - VN_EPS;
    const float sc = vn_div(s.nb, s.n);
    // pn is the only divisor without an epsilon: below the range where v_rcp_f32 stays finite the IEEE sequence takes over
    const float dirw = pn >= 1e-30f ? vn_div(gn, pn) : (pn > 0.f ? gn / pn : 0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a) gp[a] = gq[a] * sc + dirw * s.p[a];
}


// BatchNorm finalisation inside the statistics kernels (training mode): the LAST workgroup to finish -- a ticket counter next to
// the sums, zeroed with them -- turns the sums into mean / rstd and updates the running statistics, the arithmetic of
// bn_finalize_kernel below.  One launch per normalisation layer less (13 per training step).
struct BnFin {
    unsigned* counter;          // [C] tickets, zeroed (nullptr: no finalisation, the caller launches hpcs_bn_finalize)
    double count;
    float eps, momentum;
    float *rmean, *rvar, *mean, *rstd;
};
__device__ __forceinline__ void bn_finalize_channel(const double s1, const double s2, int c, double count, float eps, float momentum,
                                                    float* rmean, float* rvar, float* mean, float* rstd) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
    if (rvar) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
    }
}
// Call from every thread of every workgroup AFTER the workgroup's atomic adds into `sums` (whichever threads issued them).
// No __threadfence(): at agent scope it writes back and invalidates the XCD's L2 -- issued by thousands of workgroups it took
// the whole step from 10.9 to 17 ms.  Nothing but device-scope atomics is published here (performed at the coherence point,
// acknowledged through vmcnt), and the finalising workgroup reads the sums back with atomics; its own plain stores
// (mean / rstd / running statistics) are consumed by later kernels only.
__device__ __forceinline__ void bn_finalize_tail(const BnFin& f, double* sums, int C) {
    if (f.counter == nullptr) return;
    __shared__ int last_sh;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's atomic adds have been performed
    __syncthreads();                                          // ... and every other thread's of the workgroup
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        last_sh = __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1u;
    }
    __syncthreads();
    if (!last_sh) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double s1 = atomicAdd(sums + c, 0.0);           // read through the same path the adds took
        const double s2 = atomicAdd(sums + C + c, 0.0);
        bn_finalize_channel(s1, s2, c, f.count, f.eps, f.momentum, f.rmean, f.rvar, f.mean, f.rstd);
    }
}

// Per-channel form for the kernels whose workgroups each serve ONE channel (a single ticket counter for ~7000 workgroups
// serialises: 39 -> 110 us per statistics kernel): fin.counter[c] counts the workgroups of channel c, the last of them
// finalises that channel.  Only the thread that issued the workgroup's atomics calls this.
__device__ __forceinline__ void bn_finalize_tail_channel(const BnFin& f, double* sums, int C, int c, unsigned blocks_of_channel) {
    if (f.counter == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's adds into sums[c], sums[C + c] are performed
    if (__hip_atomic_fetch_add(f.counter + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != blocks_of_channel - 1u) return;
    const double s1 = atomicAdd(sums + c, 0.0);               // read through the same path the adds took
    const double s2 = atomicAdd(sums + C + c, 0.0);
    bn_finalize_channel(s1, s2, c, f.count, f.eps, f.momentum, f.rmean, f.rvar, f.mean, f.rstd);
}

