// issue_rate.hip -- measures, on the GPU box, the per-SIMD issue rate of wave64 VALU / SALU / LDS instructions as a function of
// resident waves per SIMD: the "cycles per wave-instruction" constant behind roofline.issue in bench.py.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
// Each kernel runs ITERS iterations of 32 independent (or dependent) instructions; grid = 256 CUs x 4 SIMDs x W waves as
// one-wave workgroups so that waves spread over SIMDs; time from HIP events; rate = wave-instructions / SIMD / cycle at the
// clock measured by s_memtime / s_memrealtime inside the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 4096

__global__ __launch_bounds__(64) void k_valu_indep(unsigned *out, unsigned long long *clk)
{
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 17 + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "v"(it));
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(it));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(64) void k_valu_dep(unsigned *out, unsigned long long *clk)
{
    unsigned a = threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 32; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(it));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}

__global__ __launch_bounds__(64) void k_pk(unsigned *out, unsigned long long *clk)
{
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 17 + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 3) & 15]), "v"(0x0c010c00u));
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_salu(unsigned *out, unsigned long long *clk)
{
    unsigned s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(it));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(it));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(it));
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(s3) : "s"(it));
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s0 + s1 + s2 + s3;
}

// 16 VALU + 16 SALU interleaved: do they co-issue?
__global__ __launch_bounds__(64) void k_mix(unsigned *out, unsigned long long *clk)
{
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 17 + i;
    unsigned s0 = blockIdx.x, s1 = 3;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(it));
            if (i & 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(it)); else asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(it));
        }
    }
    unsigned s = s0 + s1;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

// LDS byte reads at random-ish addresses (the access pattern of the FAST ring) and dword reads
__global__ __launch_bounds__(64) void k_lds_u8(unsigned *out, unsigned long long *clk)
{
    __shared__ unsigned char tile[2304];
    for (int i = threadIdx.x; i < 2304; i += 64) tile[i] = (unsigned char)(i * 7);
    __syncthreads();
    unsigned idx = (threadIdx.x * 37 + blockIdx.x) % 2000, acc = 0;
    for (int it = 0; it < ITERS / 4; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc += tile[idx + i * 17];
        idx = (idx + acc) % 2000;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <class K>
static void run(const char *name, K kern, int insts_per_iter, int iters, unsigned *d_out, unsigned long long *d_clk)
{
    printf("%-14s", name);
    for (int w = 1; w <= 8; w *= 2) {
        const int grid = 256 * 4 * w;
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out, d_clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out, d_clk);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double insts_per_simd = (double)w * iters * insts_per_iter;
        printf("  w=%d: %7.3f ms %6.2f ns/inst/simd", w, ms, ms * 1e6 / insts_per_simd);
    }
    printf("\n");
}

int main()
{
    unsigned *d_out; unsigned long long *d_clk, h_clk[2];
    CHECK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4)); CHECK(hipMalloc(&d_clk, 16));
    run("valu_indep", k_valu_indep, 32, ITERS, d_out, d_clk);
    CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
    printf("clock: %llu shader cycles in %llu x 10 ns = %.3f GHz\n", h_clk[0], h_clk[1], (double)h_clk[0] / (h_clk[1] * 10.0));
    run("valu_dep", k_valu_dep, 32, ITERS, d_out, d_clk);
    run("pk+perm", k_pk, 32, ITERS, d_out, d_clk);
    run("salu", k_salu, 32, ITERS, d_out, d_clk);
    run("valu16+salu16", k_mix, 32, ITERS, d_out, d_clk);
    run("lds_u8 x16", k_lds_u8, 16, ITERS / 4, d_out, d_clk);
    return 0;
}
