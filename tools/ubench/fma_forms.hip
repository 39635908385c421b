// fma_forms.hip -- reconciles the issue costs of tools/ubench/op_cost.hip (v_fma_f32 4.0-4.4 cycles per wave-instruction per
// SIMD, add / logic 2.5) with the guide's "v_fma_f32 (wave64) 2 cyc; one wave alone 4" (MI355X_MICROARCH.md, per-instruction
// cycle constants): the same opcodes in their VOP2 / VOP3 / packed encodings, with two or three distinct VGPR sources, at 1, 2, 4
// and 8 waves per SIMD.  16 independent accumulators per wave (throughput, not latency); every CU busy.
//   hipcc --offload-arch=gfx950 -O3 -o fma_forms fma_forms.hip && ./fma_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 2048

#define KERNEL(NAME, ASM, ...)                                                                                  \
    __global__ __launch_bounds__(64) void NAME(float *out)                                                      \
    {                                                                                                           \
        float a[16];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.017f + i * 0.3f + blockIdx.x;     \
        float b = 1.0f + threadIdx.x * 1e-7f, c = threadIdx.x * 1e-9f;                                          \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < 2; r++)                                                       \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : __VA_ARGS__);        \
        }                                                                                                       \
        float s = 0;                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s += a[i];                                               \
        out[blockIdx.x * 64 + threadIdx.x] = s + b + c;                                                         \
    }
// packed forms work on register pairs: 8 accumulators of two floats
#define KERNEL2(NAME, ASM, ...)                                                                                 \
    __global__ __launch_bounds__(64) void NAME(float *out)                                                      \
    {                                                                                                           \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                   \
        f2 a[16];                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = f2{ threadIdx.x * 0.017f + i, 0.5f + blockIdx.x };\
        f2 b = f2{ 1.0f + threadIdx.x * 1e-7f, 1.0f }, c = f2{ threadIdx.x * 1e-9f, 0.f };                      \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < 2; r++)                                                       \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : __VA_ARGS__);        \
        }                                                                                                       \
        float s = 0;                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;                                    \
        out[blockIdx.x * 64 + threadIdx.x] = s + b.x + c.x;                                                     \
    }

KERNEL(k_add_f32_vop2, "v_add_f32 %0, %0, %1", "v"(b))                       // 2 VGPR sources
KERNEL(k_mul_f32_vop2, "v_mul_f32 %0, %0, %1", "v"(b))
KERNEL(k_fmac_f32_vop2, "v_fmac_f32 %0, %1, %2", "v"(b), "v"(c))             // VOP2 fused multiply-add: reads dst + 2
KERNEL(k_fma_f32_3src, "v_fma_f32 %0, %0, %1, %2", "v"(b), "v"(c))           // VOP3, three distinct VGPRs
KERNEL(k_fma_f32_2src, "v_fma_f32 %0, %0, %1, %1", "v"(b))                   // VOP3, two distinct VGPRs
KERNEL(k_fma_f32_const, "v_fma_f32 %0, %0, %1, 1.0", "v"(b))                 // VOP3, inline constant as third source
KERNEL(k_add_f32_vop3, "v_add_f32_e64 %0, %0, %1", "v"(b))                   // the VOP3 encoding of a two-source op
KERNEL(k_max_f32_vop2, "v_max_f32 %0, %0, %1", "v"(b))
KERNEL(k_add_u32_vop2, "v_add_u32 %0, %0, %1", "v"(b))
KERNEL(k_min_u32_vop2, "v_min_u32 %0, %0, %1", "v"(b))
KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL2(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2", "v"(b), "v"(c))          // two fp32 FMAs per lane per instruction
KERNEL2(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1", "v"(b))
KERNEL2(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1", "v"(b))

typedef void (*kern_t)(float *);
static double clock_ghz = 2.3;

static double run(kern_t kern, int w, float *d_out)
{
    const int grid = 256 * 4 * w;
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
    return best * 1e6 / ((double)w * ITERS * 32) * clock_ghz;    // cycles per wave-instruction per SIMD
}

__global__ void k_clock(unsigned long long *clk, unsigned *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned a = threadIdx.x;
    for (int i = 0; i < 200000; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(i));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main()
{
    float *d_out; unsigned long long *d_clk, h_clk[2];
    CHECK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4)); CHECK(hipMalloc(&d_clk, 16));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_clk, (unsigned *)d_out);
    CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
    clock_ghz = (double)h_clk[0] / (h_clk[1] * 10.0);
    printf("shader clock (idle chip, one wave): %.3f GHz; the loaded chip clocks lower, so the cycle figures below are upper bounds by the same ratio\n", clock_ghz);
    printf("cycles per wave64 instruction per SIMD (16 independent accumulators per wave), by resident waves per SIMD\n");
    printf("%-22s %8s %8s %8s %8s\n", "form", "1 wave", "2 waves", "4 waves", "8 waves");
    struct { const char *n; kern_t k; } ks[] = {
        { "v_add_f32 (VOP2)", k_add_f32_vop2 }, { "v_mul_f32 (VOP2)", k_mul_f32_vop2 }, { "v_max_f32 (VOP2)", k_max_f32_vop2 },
        { "v_fmac_f32 (VOP2)", k_fmac_f32_vop2 }, { "v_fma_f32 3 VGPR", k_fma_f32_3src }, { "v_fma_f32 2 VGPR", k_fma_f32_2src },
        { "v_fma_f32 v,v,1.0", k_fma_f32_const }, { "v_add_f32 (VOP3 e64)", k_add_f32_vop3 },
        { "v_add_u32 (VOP2)", k_add_u32_vop2 }, { "v_min_u32 (VOP2)", k_min_u32_vop2 }, { "v_add3_u32 (VOP3)", k_add3_u32 },
        { "v_pk_fma_f32", k_pk_fma_f32 }, { "v_pk_add_f32", k_pk_add_f32 }, { "v_pk_mul_f32", k_pk_mul_f32 } };
    for (auto &k : ks) {
        printf("%-22s", k.n);
        for (int w : { 1, 2, 4, 8 }) printf(" %8.2f", run(k.k, w, d_out));
        printf("\n");
    }
    return 0;
}
