// op_cost64.hip -- issue cost of the fp64 opcodes of dev_sincos (k_desc) per wave64 instruction per SIMD, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 2048
#define KERNEL(NAME, ASM)                                                                                       \
    __global__ __launch_bounds__(64) void NAME(double *out)                                                     \
    {                                                                                                           \
        double a[8];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-9 * (threadIdx.x * 17 + i * 3 + blockIdx.x); \
        double b = 1.0000001, c = 1e-12 * threadIdx.x;                                                          \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < 4; r++)                                                       \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));      \
        }                                                                                                       \
        double s = 0;                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 8; i++) s += a[i];                                                \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                                 \
    }
KERNEL(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
typedef void (*kern_t)(double *);
static void run(const char *name, kern_t kern, double *d_out)
{
    const int w = 8, grid = 256 * 4 * w;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double insts_per_simd = (double)w * ITERS * 32;
    printf("%-12s %7.3f ms  %5.2f ns/inst/simd = %5.2f cycles at 2.4 GHz\n", name, best, best * 1e6 / insts_per_simd, best * 1e6 / insts_per_simd * 2.4);
}
int main()
{
    double *d_out;
    CHECK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 8));
    run("v_mul_f64", k_mul_f64, d_out); run("v_add_f64", k_add_f64, d_out); run("v_fma_f64", k_fma_f64, d_out);
    return 0;
}
