// dispatch_rate.hip -- how long the chip needs to START the workgroups of a launch, as a function of their number and size: the floor
// of every small-batch launch of the frame chain (a k_fast / k_desc launch of one stereo frame is 3 400 / 8 800 one-wave workgroups
// whose waves live ~5 us, yet the launches take 12-17 us).
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
// Prints, per (workgroups, threads per workgroup, LDS per workgroup, body), the time from the FIRST wave's start to the LAST wave's
// end (s_memrealtime, 100 MHz, min / max over all waves by atomics) -- events around a launch have a 6 us floor that hides it.
// Body "exit" = stamp and leave; "5us" = every wave spins 5 us first (do running waves overlap the dispatch of later ones?).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

extern __shared__ int dyn[];
__global__ void k_body(int spin_ticks, unsigned long long *mm, int touch_lds)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (touch_lds) dyn[threadIdx.x] = spin_ticks;
    if (spin_ticks)
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(4);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], t0);
        atomicMax(&mm[1], __builtin_amdgcn_s_memrealtime());
    }
}

int main()
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    unsigned long long *mm, h[2]; CHECK(hipMalloc(&mm, 16));
    const int wgs[] = { 256, 1024, 2048, 4096, 8192, 16384 };
    const int nts[] = { 64, 256, 1024 };
    const int ldss[] = { 0, 5120, 20480 };
    // s_memtime runs at 100 MHz on this part: 500 ticks = 5 us
    for (int spin : { 0, 500 })
        for (int lds : ldss)
            for (int nt : nts)
                for (int g : wgs) {
                    if ((long long)g * nt > 8192LL * 256) continue;
                    std::vector<float> t;
                    for (int it = 0; it < 30; it++) {
                        h[0] = ~0ull; h[1] = 0;
                        CHECK(hipMemcpy(mm, h, 16, hipMemcpyHostToDevice));
                        hipLaunchKernelGGL(k_body, dim3(g), dim3(nt), lds, 0, spin, mm, lds ? 1 : 0);
                        CHECK(hipGetLastError());
                        CHECK(hipDeviceSynchronize());
                        CHECK(hipMemcpy(h, mm, 16, hipMemcpyDeviceToHost));
                        if (it >= 5) t.push_back((float)(h[1] - h[0]) / 100.f);
                    }
                    std::sort(t.begin(), t.end());
                    printf("body %-4s lds %6d  threads %5d  workgroups %6d  waves %7d : %7.1f us\n", spin ? "5us" : "exit", lds, nt, g, g * nt / 64, t[t.size() / 2]);
                }
    return 0;
}
