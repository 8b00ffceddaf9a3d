// Shared helpers for the gfx950 kernels of libhpcs_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hpcs_mi355x.h"

#define HPCS_WAVE 64

#define HPCS_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return HPCS_ERR_HIP - (int)e__; \
    } while (0)

static inline int hpcs_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning switches (A/B experiments and fallbacks; the defaults are what the bench runs).  Read from the environment ONCE, at the
// first entry-point call that needs one (csrc/debug.hip holds the table), and immutable afterwards: entry points keep no other
// global state.  hpcs_debug_set_tuning (tests only) replaces a value in place.
struct HpcsTuning {
    int nt_wg_target;      // HPCS_NT_WG_TARGET   weight-gradient GEMM: workgroups per launch aimed for (0 = built-in)
    int nt_big_min;        // HPCS_NT_BIG_MIN     smallest output side that takes the 128 x 128 weight-gradient kernel (0 = 128 / 300)
    int nt_big_wg;         // HPCS_NT_BIG_WG      workgroups of that kernel (512 = two per CU)
    int gemm_vecstore;     // HPCS_GEMM_VECSTORE  16-byte epilogue stores through LDS (1)
    int gemm_tn_guarded;   // HPCS_GEMM_TN_GUARDED bounds-tested operand loader (0)
    int gemm_x3;           // HPCS_GEMM_X3        large dense GEMMs as six bf16-piece products on the bf16 MFMA, fp32 accurate (1)
    int knn_kernel;        // HPCS_KNN_KERNEL     0 auto | 1 tiled (sorting network) | 2 queue | 4 bq | 5 bq2
    int sampled_v;         // HPCS_SAMPLED_V      sampled HypHC kernel: 4 (r03 form) | 8 (r02 form, also serves E = 64)
    int sampled_seg;       // HPCS_SAMPLED_SEG    triplets per 8-lane group (0 = chosen per launch)
    int sampled_lds_kb;    // HPCS_SAMPLED_LDS_KB LDS budget of its positive table (156)
    int edge_bwd_atomic;   // HPCS_EDGE_BWD_ATOMIC graph-layer gradient reduction by atomics instead of the transposed graph (0)
    int x3_nt_lds_kb;      // HPCS_X3_NT_LDS_KB    LDS request of the bf16x3 weight-gradient kernel in KB (56: two workgroups per CU; 36 = its stage = three)
};
extern "C" const HpcsTuning* hpcs_tuning_table(void);
static inline const HpcsTuning& hpcs_tuning() { return *hpcs_tuning_table(); }

// Zero-fill of up to four buffers as ONE ordinary kernel launch.  This library never calls hipMemsetAsync: on this runtime
// (ROCm 7.0 / 7.2, graph packet capture on) the memset NODES of a replayed hipGraph lose their arguments once a second
// graph has been instantiated -- after a device-wide synchronize, or after a few thousand node launches, the fill writes
// a stale 64-bit pattern (and, for other nodes, to a stale address: a GPU memory-access fault).  That was round 2's
// "stray store" (DESIGN.md); scripts/graph_lifetime_repro2.py reproduces it with hipMemsetAsync + two kernels.  Kernel
// nodes replay correctly, and one launch instead of three memset nodes is cheaper as well.
struct HpcsZeroJob {
    void* p[4];
    unsigned long long n[4];      // bytes, multiples of 4
};
static __global__ void hpcs_zero_kernel(HpcsZeroJob job) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        char* p = (char*)job.p[j];
        const size_t n = job.n[j];
        if (p == nullptr || n == 0) continue;
        if ((((size_t)p | n) & 15) == 0) {
            uint4* q = (uint4*)p;
            for (size_t i = tid; i < (n >> 4); i += nth) q[i] = make_uint4(0u, 0u, 0u, 0u);
        } else {
            unsigned* q = (unsigned*)p;
            for (size_t i = tid; i < (n >> 2); i += nth) q[i] = 0u;
        }
    }
}
static inline int hpcs_zero_async(hipStream_t st, void* p0, size_t n0, void* p1 = nullptr, size_t n1 = 0, void* p2 = nullptr,
                                  size_t n2 = 0, void* p3 = nullptr, size_t n3 = 0) {
    HpcsZeroJob job{{p0, p1, p2, p3}, {n0, n1, n2, n3}};
    size_t mx = 0;
    for (int j = 0; j < 4; ++j) {
        if (job.p[j] == nullptr) job.n[j] = 0;
        if (job.n[j] & 3) return HPCS_ERR_SHAPE;
        if (job.n[j] > mx) mx = job.n[j];
    }
    if (mx == 0) return HPCS_OK;
    long blocks = (long)((mx + 256 * 64 - 1) / (256 * 64));       // 64 bytes per thread per pass
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hpcs_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, job);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPCS_OK : HPCS_ERR_HIP - (int)e;
}

// Strided form: `rows` blocks of `width` floats, `pitch` floats apart (the shared-direction gradient block of every cloud of
// a [B, Cout + 1, 3, L] tensor).  Same reason: hipMemset2DAsync is a memset node as well.
static __global__ void hpcs_zero_rows_kernel(float* __restrict__ p, size_t pitch, size_t width) {
    float* q = p + (size_t)blockIdx.y * pitch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (size_t)gridDim.x * blockDim.x) q[i] = 0.f;
}
static inline int hpcs_zero_rows_async(hipStream_t st, float* p, size_t pitch, size_t width, int rows) {
    if (rows <= 0 || width == 0) return HPCS_OK;
    if (rows > 65535) return HPCS_ERR_SHAPE;
    long bx = (long)((width + 1023) / 1024);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(hpcs_zero_rows_kernel, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0, st, p, pitch, width);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPCS_OK : HPCS_ERR_HIP - (int)e;
}

// Bijective XCD-aware block remap (cdna_hip_programming.md T1): hardware deals consecutive
// block ids round-robin over the 8 XCDs; give each XCD a contiguous run of logical ids so that
// neighbouring tiles (which share operand panels) hit the same L2.
__device__ __forceinline__ unsigned hpcs_xcd_remap(unsigned id, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned xcd = id & 7u, slot = id >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + slot;
}

// 32-bit counter hash (two murmur3-style finalizer rounds over a 64-bit counter) and the inverted-dropout multiplier
// derived from it: element `off` of a tensor is kept with probability 1 - p and scaled by 1 / (1 - p).  The same
// (seed, off) gives the same value in forward and backward, so the mask never exists in memory.
__device__ __forceinline__ unsigned hpcs_hash32(unsigned a, unsigned b) {
    unsigned h = a * 0xCC9E2D51u + (b ^ 0x1B873593u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += b * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ float hpcs_dropout_mult(unsigned long long seed, long long off, unsigned thresh24, float inv_keep) {
    const unsigned h = hpcs_hash32((unsigned)off ^ (unsigned)seed, (unsigned)((unsigned long long)off >> 32) ^ (unsigned)(seed >> 32));
    return (h >> 8) >= thresh24 ? inv_keep : 0.f;
}

__device__ __forceinline__ float hpcs_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double hpcs_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024).  Result valid in thread 0.
template <typename T>
__device__ __forceinline__ T hpcs_block_sum(T v, T* smem /* >= 16 entries */) {
    v = hpcs_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}
