// LDS float-atomic throughput on gfx950 (hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_atomic_bench scripts/lds_atomic_bench.hip):
// cycles per wave-instruction for the sampled HypHC kernel's positive-table pattern and its alternatives, 16 waves per CU.
// Result (r03, profiles/r03_microbenchmarks.txt): ds_add_f32 retires ONE LANE PER 3 CYCLES (193 cycles per 64-lane
// instruction); ds_add_u32 5.5, ds_add_f64 25, a ds_cmpst compare-and-swap loop 17.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int TS = 40, ROWS = 800;   // (u64 mode: rows of TS doubles do not fit: it wraps inside the table, timing only)

template <int MODE>
__global__ __launch_bounds__(1024) void k(const int* __restrict__ rows, float* __restrict__ out, int iters, long long* cyc) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < ROWS * TS; i += 1024) sm[i] = 0.f;
    __syncthreads();
    const int s = threadIdx.x & 7, g = threadIdx.x >> 3;
    const long long t0 = clock64();
    float v = 1.0f + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
        const int r = rows[(it * 128 + g) & 65535];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) atomicAdd(sm + r * TS + j * 8 + s, v);                           // table pattern, ds_add_f32
            if (MODE == 1) sm[r * TS + j * 8 + s] = v;                                      // same addresses, plain store
            if (MODE == 2) { float* p = sm + r * TS + j * 8 + s; *p = *p + v; }             // non-atomic RMW
            if (MODE == 3) atomicAdd(sm + ((threadIdx.x * 4 + j * 4096 / 4) % (ROWS * TS)), v);   // conflict-free consecutive
            if (MODE == 4) atomicAdd(reinterpret_cast<int*>(sm) + r * TS + j * 8 + s, 1);   // integer atomic, table pattern
            if (MODE == 5) __hip_atomic_fetch_add(sm + r * TS + j * 8 + s, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 6) {                                                                 // CAS loop
                int* p = reinterpret_cast<int*>(sm) + r * TS + j * 8 + s;
                int old = *p, assumed;
                do { assumed = old; old = atomicCAS(p, assumed, __float_as_int(__int_as_float(assumed) + v)); } while (old != assumed);
            }
            if (MODE == 7) atomicAdd(reinterpret_cast<double*>(sm) + ((r * TS + j * 8 + s) >> 1) + (s & 1), (double)v);   // ds_add_f64
            if (MODE == 8) { if (s < 4) atomicAdd(sm + r * TS + j * 8 + s, v); }             // half the lanes
            if (MODE == 9) { if (g % 8 == 0) atomicAdd(sm + r * TS + j * 8 + s, v); }        // 8 lanes of 64
            if (MODE == 10) atomicAdd(reinterpret_cast<unsigned long long*>(sm) + ((r * TS) >> 1) + j * 8 + s, (unsigned long long)(long long)(v * 68719476736.f));
        }
        v += 1e-7f;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float acc = 0.f;
    for (int i = threadIdx.x; i < ROWS * TS; i += 1024) acc += sm[i];
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const int* d_rows, float* d_out, long long* d_cyc, int iters) {
    const int grid = 256;
    const size_t lds = ROWS * TS * sizeof(float) * 2 > 160 * 1024 ? 160 * 1024 : ROWS * TS * sizeof(float) * 2;
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), lds, 0, d_rows, d_out, iters, d_cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), lds, 0, d_rows, d_out, iters, d_cyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(grid); CK(hipMemcpy(c.data(), d_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    // per CU: 16 waves x iters x 4 wave-instructions
    const double instr_per_cu = 16.0 * iters * 4;
    printf("%-42s %8.3f ms  clock64 ticks/WG %lld  -> %.1f ns per wave-instr per CU (%.1f cycles @2.4GHz)\n", name, ms, c[0],
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
}

int main() {
    std::vector<int> rows(65536);
    srand(1);
    for (auto& r : rows) r = rand() % ROWS;
    int* d_rows; float* d_out; long long* d_cyc;
    CK(hipMalloc(&d_rows, rows.size() * 4)); CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 256 * 1024 * 4)); CK(hipMalloc(&d_cyc, 256 * 8));
    const int iters = 4000;
    run<0>("ds_add_f32, table pattern (random rows)", d_rows, d_out, d_cyc, iters);
    run<1>("ds_write_b32, table pattern", d_rows, d_out, d_cyc, iters);
    run<2>("read+add+write (non-atomic), table pattern", d_rows, d_out, d_cyc, iters);
    run<3>("ds_add_f32, consecutive addresses", d_rows, d_out, d_cyc, iters);
    run<4>("ds_add_u32, table pattern", d_rows, d_out, d_cyc, iters);
    run<5>("fetch_add f32 workgroup scope, table pattern", d_rows, d_out, d_cyc, iters);
    run<6>("CAS loop f32, table pattern", d_rows, d_out, d_cyc, iters);
    run<7>("ds_add_f64, table pattern", d_rows, d_out, d_cyc, iters);
    run<8>("ds_add_f32, 32 of 64 lanes", d_rows, d_out, d_cyc, iters);
    run<9>("ds_add_f32, 8 of 64 lanes", d_rows, d_out, d_cyc, iters);
    run<10>("ds_add_u64 fixed point (incl. conversion)", d_rows, d_out, d_cyc, iters);
    return 0;
}
