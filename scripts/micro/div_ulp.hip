// Accuracy of the quotient sequences the VN point-wise code can use, in ulps of the fp64 quotient.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/div_ulp.hip -o gpurun_out/div_ulp && gpurun_out/div_ulp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const float* a, const float* b, float* o0, float* o1, float* o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    o0[i] = x / y;
    {
        const float r = __builtin_amdgcn_rcpf(y);
        const float q = x * r;
        const float e = __builtin_fmaf(-y, q, x);
        o1[i] = __builtin_fmaf(e, r, q);
    }
    o2[i] = x * __builtin_amdgcn_rcpf(y);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), b(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        a[i] = (float)((rand() / (double)RAND_MAX - 0.5) * 4.0);
        b[i] = (float)std::exp((rand() / (double)RAND_MAX - 0.5) * 14.0);
    }
    float *da, *db, *d0, *d1, *d2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, d0, d1, d2, n);
    std::vector<float> o(n);
    float* outs[3] = {d0, d1, d2};
    const char* names[3] = {"a / b", "rcp + newton", "a * rcp(b)"};
    for (int v = 0; v < 3; ++v) {
        hipMemcpy(o.data(), outs[v], n * 4, hipMemcpyDeviceToHost);
        double sum = 0, mx = 0; long wrong = 0;
        for (int i = 0; i < n; ++i) {
            const double q = (double)a[i] / (double)b[i];
            const float qf = (float)q;
            const double ulp = std::ldexp(1.0, std::ilogb(q) - 23);
            const double e = std::fabs((double)o[i] - q) / ulp;
            sum += e; if (e > mx) mx = e; wrong += (o[i] != qf);
        }
        printf("%-14s mean %.4f ulp  max %.4f ulp  not correctly rounded: %ld of %d\n", names[v], sum / n, mx, wrong, n);
    }
    return 0;
}
