// call_latency.hip -- what one synchronous "host pointers in, host pointers out" call costs on this box, by transport:
// the design input for the per-call forms of the matchers (orbx_match.hip), which move ~90 KB in and ~4 KB out around
// a kernel of a few microseconds.
//   hipcc --offload-arch=gfx950 -O3 -o call_latency call_latency.hip && ./call_latency
// Variants (median / p90 of 2000 calls each, microseconds):
//   A  empty kernel + hipStreamSynchronize
//   B  empty kernel, completion by a flag in coherent pinned memory that the host polls
//   C  hipMemcpyAsync H2D (96 KB) + kernel (reads it all, writes 4 KB) + hipMemcpyAsync D2H + hipStreamSynchronize
//   D  H2D copy + kernel writing its 4 KB straight to pinned memory + flag poll (no D2H command)
//   E  kernel reads the 96 KB straight from pinned memory (coherent), writes to pinned, flag poll (no copy commands)
//   F  as E with non-coherent pinned input (GPU L2 may cache it)
//   G  as E, every input line read 10 times (the reuse factor of a 10 x 10 vocabulary node)
//   H  as F, read 10 times
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define IN_BYTES (96 * 1024)
#define OUT_INTS 1024

__global__ void k_empty() {}

__global__ void k_flag(volatile unsigned *flag, unsigned ticket)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = ticket; }
}

// 96 one-wave workgroups, each sums its 1 KB slice `reps` times and writes 1/96 of the output; the last block to
// finish publishes the flag (the pattern of the matcher's last-arriving block)
__global__ __launch_bounds__(64) void k_work(const uint4 *__restrict__ in, int *__restrict__ out, unsigned *counter,
                                             volatile unsigned *flag, unsigned ticket, int reps)
{
    const int lane = threadIdx.x, wg = blockIdx.x;
    unsigned s = 0;
    for (int r = 0; r < reps; r++) {
        const uint4 v = in[wg * 64 + ((lane + r) & 63)];
        s += v.x ^ v.y ^ v.z ^ v.w;
    }
    for (int i = lane; i < OUT_INTS / 96; i += 64) out[wg * (OUT_INTS / 96) + i] = (int)s + i;
    __threadfence_system();
    if (lane == 0) {
        const unsigned prev = atomicAdd(counter, 1u);
        if (prev == gridDim.x - 1) {
            *counter = 0;
            if (flag) { __threadfence_system(); *flag = ticket; }
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void report(const char *name, std::vector<double> &t)
{
    std::sort(t.begin(), t.end());
    printf("%-64s median %7.1f us   p10 %7.1f   p90 %7.1f\n", name, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10]);
}

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *h_flag; int *h_out_p; unsigned char *h_in_c, *h_in_nc, *h_in_plain; int *h_out_plain;
    CHECK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_out_p, OUT_INTS * 4, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_in_c, IN_BYTES, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_in_nc, IN_BYTES, hipHostMallocNonCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&h_in_plain, IN_BYTES, hipHostMallocDefault));
    CHECK(hipHostMalloc((void **)&h_out_plain, OUT_INTS * 4, hipHostMallocDefault));
    unsigned char *d_in; int *d_out; unsigned *d_counter;
    CHECK(hipMalloc((void **)&d_in, IN_BYTES));
    CHECK(hipMalloc((void **)&d_out, OUT_INTS * 4));
    CHECK(hipMalloc((void **)&d_counter, 4));
    CHECK(hipMemset(d_counter, 0, 4));
    std::vector<unsigned char> src(IN_BYTES);
    for (int i = 0; i < IN_BYTES; i++) src[i] = (unsigned char)(i * 7 + 3);
    *h_flag = 0;
    const int N = 2000, WARM = 50;
    unsigned ticket = 0;
    std::vector<double> t;
    auto poll = [&](unsigned tk) { while (*(volatile unsigned *)h_flag != tk) { } };

    t.clear();
    for (int i = 0; i < N + WARM; i++) {
        const double a = now_us();
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        CHECK(hipStreamSynchronize(st));
        if (i >= WARM) t.push_back(now_us() - a);
    }
    report("A empty kernel + hipStreamSynchronize", t);

    t.clear();
    for (int i = 0; i < N + WARM; i++) {
        const double a = now_us();
        ticket++;
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, h_flag, ticket);
        poll(ticket);
        if (i >= WARM) t.push_back(now_us() - a);
    }
    CHECK(hipStreamSynchronize(st));
    report("B empty kernel + flag poll in coherent pinned memory", t);

    t.clear();
    for (int i = 0; i < N + WARM; i++) {
        const double a = now_us();
        memcpy(h_in_plain, src.data(), IN_BYTES);
        CHECK(hipMemcpyAsync(d_in, h_in_plain, IN_BYTES, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_work, dim3(96), dim3(64), 0, st, (const uint4 *)d_in, d_out, d_counter, (volatile unsigned *)nullptr, 0u, 1);
        CHECK(hipMemcpyAsync(h_out_plain, d_out, OUT_INTS * 4, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        if (i >= WARM) t.push_back(now_us() - a);
    }
    report("C memcpy + H2D copy + kernel + D2H copy + sync", t);

    t.clear();
    for (int i = 0; i < N + WARM; i++) {
        const double a = now_us();
        ticket++;
        memcpy(h_in_plain, src.data(), IN_BYTES);
        CHECK(hipMemcpyAsync(d_in, h_in_plain, IN_BYTES, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_work, dim3(96), dim3(64), 0, st, (const uint4 *)d_in, h_out_p, d_counter, h_flag, ticket, 1);
        poll(ticket);
        if (i >= WARM) t.push_back(now_us() - a);
    }
    CHECK(hipStreamSynchronize(st));
    report("D memcpy + H2D copy + kernel writing pinned + flag poll", t);

    struct V { const char *name; unsigned char *in; int reps; bool sync; };
    const V vs[] = { { "E memcpy + kernel reading coherent pinned + flag poll", h_in_c, 1, false },
                     { "F memcpy + kernel reading non-coherent pinned + flag poll", h_in_nc, 1, false },
                     { "G as E, every line read 10 times", h_in_c, 10, false },
                     { "H as F, every line read 10 times", h_in_nc, 10, false },
                     { "I as E, completion by hipStreamSynchronize", h_in_c, 1, true } };
    for (const V &v : vs) {
        t.clear();
        for (int i = 0; i < N + WARM; i++) {
            const double a = now_us();
            ticket++;
            memcpy(v.in, src.data(), IN_BYTES);
            hipLaunchKernelGGL(k_work, dim3(96), dim3(64), 0, st, (const uint4 *)v.in, h_out_p, d_counter, h_flag, ticket, v.reps);
            if (v.sync) CHECK(hipStreamSynchronize(st)); else poll(ticket);
            if (i >= WARM) t.push_back(now_us() - a);
        }
        CHECK(hipStreamSynchronize(st));
        report(v.name, t);
    }
    // the host memcpy alone
    t.clear();
    for (int i = 0; i < N + WARM; i++) {
        const double a = now_us();
        memcpy(h_in_c, src.data(), IN_BYTES);
        if (i >= WARM) t.push_back(now_us() - a);
    }
    report("  (host memcpy of 96 KB into pinned memory alone)", t);
    return 0;
}
