// pk3_cost.hip -- gfx950's packed three-input f16 minimum / maximum (v_pk_minimum3_f16 / v_pk_maximum3_f16) as an INTEGER min / max of two
// 8-bit values per register: issue cost beside v_min3_u32, and exactness on the bit patterns 0x4000 + x (x = 0..255: positive NORMAL halves whose
// order is the order of x, so no denormal mode is involved).  Candidate for k_fast's arc network (two ring pixels per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o pk3_cost pk3_cost.hip && ./pk3_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 2048
#define KERNEL(NAME, ASM, ...)                                                                                  \
    __global__ __launch_bounds__(64) void NAME(unsigned *out)                                                   \
    {                                                                                                           \
        unsigned a[16];                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = 0x40004000u | ((threadIdx.x * 17 + i * 3 + blockIdx.x) & 0x00FF00FFu); \
        unsigned b = 0x40004000u | ((threadIdx.x ^ 5) & 0x00FF00FFu), c = 0x40004000u | ((threadIdx.x * 3 + 1) & 0x00FF00FFu);      \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < 2; r++)                                                       \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : __VA_ARGS__);        \
        }                                                                                                       \
        unsigned s = 0;                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s += a[i];                                               \
        out[blockIdx.x * 64 + threadIdx.x] = s + b + c;                                                         \
    }
KERNEL(k_min3_u32, "v_min3_u32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1", "v"(b))
KERNEL(k_pk_min3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_pk_max3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1", "v"(b))
KERNEL(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1", "v"(b))
KERNEL(k_min3_f16, "v_min3_f16 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_min3_u16, "v_min3_u16 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_or_lit, "v_or_b32 %0, 0x40004000, %0", "v"(b))
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_alignbit16, "v_alignbit_b32 %0, %0, %0, 16", "v"(b))

typedef void (*kern_t)(unsigned *);
static void run(const char *name, kern_t kern, unsigned *d_out)
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
    printf("%-16s %7.3f ms  %5.2f ns per wave-instruction per SIMD = %4.2f cycles at 2.4 GHz\n", name, best, best * 1e6 / insts_per_simd, best * 1e6 / insts_per_simd * 2.4);
}

// exactness: every (x, y, z) in 0..255 through both halves of one instruction against the integer min / max
__global__ void k_exact(unsigned *bad)
{
    const unsigned x = blockIdx.x, y = threadIdx.x;     // 256 x 256 threads, z loops
    unsigned nbad = 0;
    for (unsigned z = 0; z < 256; z++) {
        const unsigned a = 0x40004000u | x | ((255u - x) << 16), b = 0x40004000u | y | ((255u - y) << 16), c = 0x40004000u | z | ((255u - z) << 16);
        unsigned mn, mx;
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(a), "v"(b), "v"(c));
        const unsigned lo = min(min(x, y), z), hi = max(max(x, y), z);
        // low half: min / max of (x, y, z); high half: min / max of (255 - x, ...) = 255 - max / 255 - min
        if (mn != (0x40004000u | lo | ((255u - hi) << 16))) nbad++;
        if (mx != (0x40004000u | hi | ((255u - lo) << 16))) nbad++;
    }
    if (nbad) atomicAdd(bad, nbad);
    // the same WITHOUT the 0x4000 bias: the patterns 0x0000 .. 0x00FF are f16 subnormals -- exact only if the kernel's float mode keeps them
    unsigned nbad2 = 0;
    for (unsigned z = 0; z < 256; z++) {
        const unsigned a = x | ((255u - x) << 16), b = y | ((255u - y) << 16), c = z | ((255u - z) << 16);
        unsigned mn, mx;
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(a), "v"(b), "v"(c));
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(a), "v"(b), "v"(c));
        const unsigned lo = min(min(x, y), z), hi = max(max(x, y), z);
        if (mn != (lo | ((255u - hi) << 16))) nbad2++;
        if (mx != (hi | ((255u - lo) << 16))) nbad2++;
    }
    if (nbad2) atomicAdd(bad + 1, nbad2);
}

int main()
{
    unsigned *d_out, *d_bad, h_bad2[2] = { 0, 0 };
    CHECK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4)); CHECK(hipMalloc(&d_bad, 8)); CHECK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(k_exact, dim3(256), dim3(256), 0, 0, d_bad);
    CHECK(hipMemcpy(h_bad2, d_bad, 8, hipMemcpyDeviceToHost));
    const unsigned h_bad = h_bad2[0];
    printf("the same on the unbiased patterns 0x0000 + x (f16 subnormals, this build's float mode): %u mismatches\n", h_bad2[1]);
    printf("v_pk_minimum3_f16 / v_pk_maximum3_f16 on the patterns 0x4000 + x, all 2^24 triples x both halves: %u mismatches against the integer min / max\n", h_bad);
    run("v_add_u32", k_add_u32, d_out); run("v_min3_u32", k_min3_u32, d_out); run("v_pk_minimum3_f16", k_pk_min3_f16, d_out); run("v_pk_maximum3_f16", k_pk_max3_f16, d_out);
    run("v_pk_min_u16", k_pk_min_u16, d_out); run("v_pk_min_f16", k_pk_min_f16, d_out); run("v_min3_f16", k_min3_f16, d_out); run("v_min3_u16", k_min3_u16, d_out);
    run("v_or_b32 literal", k_or_lit, d_out); run("v_perm_b32", k_perm, d_out); run("v_alignbit 16", k_alignbit16, d_out);
    return h_bad != 0;
}
