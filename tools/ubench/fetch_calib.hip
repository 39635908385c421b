// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the extraction kernels use:
// every kernel below streams a 512 MiB buffer exactly once (larger than the 32 MiB of L2 and the 256 MiB Infinity Cache), so the
// bytes that must cross the L2's memory side are known.  Run each counter in its own pass:
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
// tools/collect_pmc.py divides the counter (KiB) by the known KiB -> factor per pattern (MI355X_MICROARCH.md: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated").
//   k_rd_dword      global_load_dword, 4 B per lane          (generic loads)
//   k_rd_dwordx4    global_load_dwordx4, 16 B per lane
//   k_rd_lds_dword  global_load_lds_dword, 4 B per lane      (tile loads of k_fast / k_desc)
//   k_rd_lds_x4     global_load_lds_dwordx4, 16 B per lane   (tile loads of k_resize)
//   k_wr_dword / k_wr_dwordx4                                (stores)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define BYTES (512ull << 20)

__global__ __launch_bounds__(256) void k_rd_dword(const unsigned *__restrict__ p, unsigned *out, long long n)
{
    unsigned s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += p[i];
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_rd_dwordx4(const uint4 *__restrict__ p, unsigned *out, long long n)
{
    unsigned s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(64) void k_rd_lds_dword(const unsigned char *__restrict__ p, unsigned *out, long long nwave_chunks)
{
    __shared__ unsigned tile[64 * 8];
    unsigned s = 0;
    for (long long c = blockIdx.x; c < nwave_chunks; c += gridDim.x) {          // a chunk = 8 wave instructions x 256 B
        const unsigned char *b = p + c * 2048 + 4 * threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_amdgcn_global_load_lds((const unsigned *)(b + 256 * k), tile + 64 * k, 4, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        s += tile[threadIdx.x] ^ tile[64 * 7 + threadIdx.x];
        __syncthreads();
    }
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(64) void k_rd_lds_x4(const unsigned char *__restrict__ p, unsigned *out, long long nwave_chunks)
{
    __shared__ __align__(16) unsigned tile[256 * 4];
    unsigned s = 0;
    for (long long c = blockIdx.x; c < nwave_chunks; c += gridDim.x) {          // a chunk = 4 wave instructions x 1 KB
        const unsigned char *b = p + c * 4096 + 16 * threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; k++) __builtin_amdgcn_global_load_lds((const unsigned *)(b + 1024 * k), tile + 256 * k, 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        s += tile[threadIdx.x] ^ tile[256 * 3 + threadIdx.x];
        __syncthreads();
    }
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_wr_dword(unsigned *p, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = (unsigned)i;
}
__global__ __launch_bounds__(256) void k_wr_dwordx4(uint4 *p, long long n)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main()
{
    unsigned char *d; unsigned *out;
    CHECK(hipMalloc((void **)&d, BYTES + 4096)); CHECK(hipMalloc((void **)&out, 64));
    CHECK(hipMemset(d, 1, BYTES + 4096));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_rd_dword, dim3(8192), dim3(256), 0, 0, (const unsigned *)d, out, (long long)(BYTES / 4));
        hipLaunchKernelGGL(k_rd_dwordx4, dim3(8192), dim3(256), 0, 0, (const uint4 *)d, out, (long long)(BYTES / 16));
        hipLaunchKernelGGL(k_rd_lds_dword, dim3(16384), dim3(64), 0, 0, d, out, (long long)(BYTES / 2048));
        hipLaunchKernelGGL(k_rd_lds_x4, dim3(16384), dim3(64), 0, 0, d, out, (long long)(BYTES / 4096));
        hipLaunchKernelGGL(k_wr_dword, dim3(8192), dim3(256), 0, 0, (unsigned *)d, (long long)(BYTES / 4));
        hipLaunchKernelGGL(k_wr_dwordx4, dim3(8192), dim3(256), 0, 0, (uint4 *)d, (long long)(BYTES / 16));
        CHECK(hipDeviceSynchronize());
    }
    printf("{\"bytes\": %llu}\n", (unsigned long long)BYTES);
    return 0;
}
