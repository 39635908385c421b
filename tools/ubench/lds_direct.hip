// lds_direct.hip -- does global_load_lds_dword accept byte-unaligned global addresses on gfx950, and where do the lanes' dwords land?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k(const unsigned char *g, int byte_off, unsigned *out)
{
    __shared__ unsigned tile[256];
    for (int i = threadIdx.x; i < 256; i += 64) tile[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned char *p = g + byte_off + 4 * threadIdx.x;
    if (threadIdx.x % 16 < 11)   // masked lanes leave holes
        __builtin_amdgcn_global_load_lds((const unsigned *)p, tile + 64, 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = tile[i];
}

int main()
{
    unsigned char h[1024];
    for (int i = 0; i < 1024; i++) h[i] = (unsigned char)i;
    unsigned char *d; unsigned *d_out, o[256];
    CHECK(hipMalloc(&d, 1024)); CHECK(hipMalloc(&d_out, 1024));
    CHECK(hipMemcpy(d, h, 1024, hipMemcpyHostToDevice));
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, off, d_out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(o, d_out, 1024, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int l = 0; l < 64; l++) {
            unsigned exp = 0xdeadbeefu;
            if (l % 16 < 11) { const int b = off + 4 * l; exp = h[b] | (h[b + 1] << 8) | (h[b + 2] << 16) | ((unsigned)h[b + 3] << 24); }
            if (o[64 + l] != exp) { ok = 0; printf("off %d lane %d: got %08x expected %08x\n", off, l, o[64 + l], exp); if (l > 3) break; }
        }
        printf("byte offset %d: %s (untouched before/after: %08x %08x)\n", off, ok ? "OK" : "MISMATCH", o[63], o[128]);
    }
    return 0;
}
