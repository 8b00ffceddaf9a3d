// SO(3) augmentation at the head of the training step: hpcs/models/shapenet_hyp_hc.py:63-69 draws one rotation per
// cloud with pytorch3d's random_rotations (normalised Gaussian quaternion -> matrix) on the CPU, applies it with
// Rotate.transform_points (row vectors times R) and transposes the batch to [B,3,N] for the extractor (:73).
// Here the three steps are one launch: quaternion from a counter RNG (or a caller-given matrix), rotate, transpose.
#include "common.h"

namespace {
__device__ __forceinline__ unsigned hash_u32(unsigned a, unsigned b) {
    unsigned h = a * 0xCC9E2D51u + (b ^ 0x1B873593u);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += b * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ __forceinline__ float unit_open(unsigned h) { return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

__global__ __launch_bounds__(256) void so3_rotate_kernel(const float* __restrict__ points, int N, uint64_t seed,
                                                         const uint64_t* __restrict__ seed_dev,
                                                         const float* __restrict__ rot_in, float* __restrict__ x_out,
                                                         float* __restrict__ rot_out) {
    __shared__ float R[9];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        if (rot_in) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = rot_in[b * 9 + i];
        } else {
            // four standard normals by Box-Muller from four counter-hash uniforms
            if (seed_dev) seed = *seed_dev;          // graph-replayable launches read the seed from device memory
            const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32);
            const float u0 = unit_open(hash_u32(s0 ^ 0xA511E9B3u, 4u * b + s1)), u1 = unit_open(hash_u32(s0 ^ 0x63D83595u, 4u * b + 1u + s1));
            const float u2 = unit_open(hash_u32(s0 ^ 0x9E3779B9u, 4u * b + 2u + s1)), u3 = unit_open(hash_u32(s0 ^ 0x7F4A7C15u, 4u * b + 3u + s1));
            const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
            const float t0 = 6.283185307179586f * u1, t1 = 6.283185307179586f * u3;
            float w = r0 * cosf(t0), x = r0 * sinf(t0), y = r1 * cosf(t1), z = r1 * sinf(t1);
            const float inv = 1.f / fmaxf(sqrtf(w * w + x * x + y * y + z * z), 1e-12f);
            w *= inv; x *= inv; y *= inv; z *= inv;
            R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - z * w);       R[2] = 2.f * (x * z + y * w);
            R[3] = 2.f * (x * y + z * w);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - x * w);
            R[6] = 2.f * (x * z - y * w);       R[7] = 2.f * (y * z + x * w);       R[8] = 1.f - 2.f * (x * x + y * y);
        }
        if (rot_out && blockIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) rot_out[b * 9 + i] = R[i];
        }
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* p = points + ((int64_t)b * N + n) * 3;
    const float p0 = p[0], p1 = p[1], p2 = p[2];
    float* o = x_out + (int64_t)b * 3 * N + n;
    // row vector times R: out_j = sum_i p_i R[i][j]
    o[0] = p0 * R[0] + p1 * R[3] + p2 * R[6];
    o[N] = p0 * R[1] + p1 * R[4] + p2 * R[7];
    o[2 * (int64_t)N] = p0 * R[2] + p1 * R[5] + p2 * R[8];
}
}  // namespace

extern "C" int hpcs_so3_rotate_points(const float* points, int B, int N, uint64_t seed, const uint64_t* seed_dev,
                                      const float* rot_in, float* x_out, float* rot_out, hpcs_stream_t stream) {
    if (!points || !x_out) return HPCS_ERR_NULL;
    if (B < 0 || N <= 0 || B > 65535) return HPCS_ERR_SHAPE;
    if (B == 0) return HPCS_OK;
    hipLaunchKernelGGL(so3_rotate_kernel, dim3(hpcs_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, points, N, seed, seed_dev,
                       rot_in, x_out, rot_out);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}
