// op_cost.hip -- cycles per wave64 instruction per SIMD for the opcodes the FAST / descriptor kernels are built from, at 8 waves
// per SIMD with 16 independent accumulators (throughput, not latency).  Prices roofline.issue.
//   hipcc --offload-arch=gfx950 -O3 -o op_cost op_cost.hip && ./op_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 2048

#define KERNEL(NAME, ASM, ...)                                                                                  \
    __global__ __launch_bounds__(64) void NAME(unsigned *out)                                                   \
    {                                                                                                           \
        unsigned a[16];                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 17 + i * 3 + blockIdx.x;            \
        unsigned b = threadIdx.x ^ 5, c = threadIdx.x * 3 + 1;                                                  \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < 2; r++)                                                       \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : __VA_ARGS__);        \
        }                                                                                                       \
        unsigned s = 0;                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; i++) s += a[i];                                               \
        out[blockIdx.x * 64 + threadIdx.x] = s + b + c;                                                         \
    }

KERNEL(k_add, "v_add_u32 %0, %0, %1", "v"(b))
KERNEL(k_min, "v_min_u32 %0, %0, %1", "v"(b))
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_min3_const, "v_min3_u32 %0, %0, %1, 7", "v"(b))
KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_min_u16, "v_min_u16 %0, %0, %1", "v"(b))
KERNEL(k_min3_u16, "v_min3_u16 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_pk_max, "v_pk_max_i16 %0, %0, %1", "v"(b))
KERNEL(k_pk_sub, "v_pk_sub_i16 %0, %0, %1", "v"(b))
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1", "v"(b))
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_perm_s, "v_perm_b32 %0, %0, %1, %2", "v"(b), "s"(0x0c010c00u))
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1", "v"(b))
KERNEL(k_and, "v_and_b32 %0, %0, %1", "v"(b))
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1", "v"(b))
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8", "v"(b))
KERNEL(k_lshr, "v_lshrrev_b32 %0, 8, %0", "v"(b))
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1", "v"(b))
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", "v"(b))
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1", "v"(b))
KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1", "v"(b))
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %0, %1", "v"(b))
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "v"(b))
KERNEL(k_sdwa_min, "v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", "v"(b))
KERNEL(k_sdwa_sub, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3", "v"(b))
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v"(b))
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_pk_fma, "v_pk_fma_f16 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_max_f32, "v_max_f32 %0, %0, %1", "v"(b))
KERNEL(k_max3_f32, "v_max3_f32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1", "v"(b))
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %1", "v"(b))
KERNEL(k_cmp_s, "v_cmp_lt_u32 s[20:21], %0, %1", "v"(b) : "s20", "s21")


KERNEL(k_sub, "v_sub_u32 %0, %0, %1", "v"(b))
KERNEL(k_or, "v_or_b32 %0, %0, %1", "v"(b))
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", "v"(b))
KERNEL(k_and_lit, "v_and_b32 %0, 0x7f7f7f7f, %0", "v"(b))
KERNEL(k_and_s, "v_and_b32 %0, %1, %0", "s"(0x7f7f7f7f))
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0", "v"(b))
KERNEL(k_ashr, "v_ashrrev_i32 %0, 1, %0", "v"(b))
KERNEL(k_mov, "v_mov_b32 %0, %1", "v"(b))
KERNEL(k_not, "v_not_b32 %0, %0", "v"(b))
KERNEL(k_max_u16, "v_max_u16 %0, %0, %1", "v"(b))
KERNEL(k_max_i16, "v_max_i16 %0, %0, %1", "v"(b))
KERNEL(k_min_i16, "v_min_i16 %0, %0, %1", "v"(b))
KERNEL(k_add_u16, "v_add_u16 %0, %0, %1", "v"(b))
KERNEL(k_sub_u16, "v_sub_u16 %0, %0, %1", "v"(b))
KERNEL(k_max_u32, "v_max_u32 %0, %0, %1", "v"(b))
KERNEL(k_max_i32, "v_max_i32 %0, %0, %1", "v"(b))
KERNEL(k_min_i32, "v_min_i32 %0, %0, %1", "v"(b))
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1", "v"(b))
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1", "v"(b))
KERNEL(k_add_f16, "v_add_f16 %0, %0, %1", "v"(b))
KERNEL(k_max_f16, "v_max_f16 %0, %0, %1", "v"(b))
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 8, %1", "v"(b))
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2", "v"(b), "v"(c))
KERNEL(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %0", "v"(b))
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %0, %1, s[20:21]", "v"(b) : "s20", "s21")
KERNEL(k_add_vop3, "v_add_u32_e64 %0, %0, %1", "v"(b))
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1", "v"(b) : "vcc")
KERNEL(k_min_u16_sdwa, "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1", "v"(b))
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD", "v"(b))
KERNEL(k_mul_u16, "v_mul_lo_u16 %0, %0, %1", "v"(b))
KERNEL(k_lshl_b16, "v_lshlrev_b16 %0, 1, %0", "v"(b))
KERNEL(k_ballot_like, "v_cmp_ne_u32 vcc, 0, %0", "v"(b) : "vcc")
KERNEL(k_readlane, "v_readfirstlane_b32 s20, %0", "v"(b) : "s20")

typedef void (*kern_t)(unsigned *);
static double clock_ghz = 2.3;

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
    printf("%-14s %7.3f ms  %5.2f ns/inst/simd  = %4.2f cycles at %.2f GHz\n", name, best, best * 1e6 / insts_per_simd,
           best * 1e6 / insts_per_simd * clock_ghz, clock_ghz);
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
    unsigned *d_out; unsigned long long *d_clk, h_clk[2];
    CHECK(hipMalloc(&d_out, 256 * 4 * 8 * 64 * 4)); CHECK(hipMalloc(&d_clk, 16));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d_clk, d_out);
    CHECK(hipMemcpy(h_clk, d_clk, 16, hipMemcpyDeviceToHost));
    clock_ghz = (double)h_clk[0] / (h_clk[1] * 10.0);
    printf("shader clock (idle chip, one wave): %.3f GHz\n", clock_ghz);
#define R(k) run(#k, k, d_out)
    R(k_add); R(k_min); R(k_min3); R(k_min3_const); R(k_max3_i32); R(k_min_u16); R(k_min3_u16); R(k_pk_max); R(k_pk_sub); R(k_pk_min_u16);
    R(k_perm); R(k_perm_s); R(k_alignbyte); R(k_and); R(k_and_or); R(k_lshl_add); R(k_add3); R(k_bfe); R(k_lshr); R(k_mul24); R(k_mad24);
    R(k_mul_lo); R(k_mul_hi); R(k_sad_u8); R(k_bcnt); R(k_mbcnt); R(k_cndmask); R(k_sdwa_min); R(k_sdwa_sub); R(k_dot4); R(k_med3); R(k_mov_dpp);
    R(k_sub); R(k_or); R(k_xor); R(k_and_lit); R(k_and_s); R(k_lshl); R(k_ashr); R(k_mov); R(k_not); R(k_max_u16); R(k_max_i16); R(k_min_i16); R(k_add_u16); R(k_sub_u16);
    R(k_max_u32); R(k_max_i32); R(k_min_i32); R(k_add_f32); R(k_mul_f32); R(k_add_f16); R(k_max_f16); R(k_or3); R(k_lshl_or); R(k_xad); R(k_cvt_ub0); R(k_cndmask_s);
    R(k_add_vop3); R(k_add_co); R(k_min_u16_sdwa); R(k_add_sdwa); R(k_mul_u16); R(k_lshl_b16); R(k_ballot_like); R(k_readlane);
    R(k_fma); R(k_pk_fma); R(k_max_f32); R(k_max3_f32); R(k_pk_max_f16); R(k_cmp); R(k_cmp_s);
    return 0;
}
