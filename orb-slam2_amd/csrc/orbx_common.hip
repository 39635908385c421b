// orbx_common.hip — error reporting, device query, profiling hooks, host Hamming.
#include "orbx_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void orbx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *orbx_last_error(void) { return g_err; }

extern "C" int orbx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// "<pci bus id> <uuid hex> <name>" of device `device`: what an N-rank launch prints per rank so that a reader can see that every
// rank held its own card (bench.py config.rank_devices)
extern "C" int orbx_device_identity(int device, char *buf, int cap)
{
    if (!buf || cap < 16) { orbx_set_error("identity buffer too small"); return ORBX_E_INVALID; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { orbx_set_error("device %d out of range (%d visible)", device, n); return ORBX_E_NO_DEVICE; }
    char bus[32] = "?";
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) strcpy(bus, "?");
    hipUUID u;
    char hex[33] = "";
    if (hipDeviceGetUuid(&u, device) == hipSuccess)
        for (int i = 0; i < 16; i++) snprintf(hex + 2 * i, 3, "%02x", (unsigned char)u.bytes[i]);
    hipDeviceProp_t p;
    const char *name = hipGetDeviceProperties(&p, device) == hipSuccess ? p.gcnArchName : "?";
    snprintf(buf, cap, "%s %s %s", bus, hex[0] ? hex : "?", name);
    return ORBX_OK;
}

// ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1733-1749): popcount(a XOR b), 256 bits
extern "C" int orbx_hamming(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

// ---- per-launch HIP-event timing on the launch stream (bench.py's roofline leg).
// Event records are not free (each is a barrier packet on the queue), so back-to-back launches on one
// stream share the boundary event: the end event of launch i is the begin event of launch i+1.  A
// launch's time therefore runs from the completion of its predecessor to its own completion.

static hipEvent_t prof_new_event(orbx_extractor *e)
{
    hipEvent_t ev = nullptr;
    if (!e->prof_pool.empty()) { ev = e->prof_pool.back(); e->prof_pool.pop_back(); return ev; }
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

void orbx_prof_begin(orbx_extractor *e, int stage, hipStream_t s)
{
    if (!e->prof) return;
    if (!((e->prof_mask >> stage) & 1u)) { e->prof_chain = false; return; }    // stage not selected: its launches break the event chain
    ProfEvent ev;
    ev.stage = stage;
    ev.b = nullptr;
    if (!e->prof_ev.empty() && e->prof_ev.back().b && e->prof_last_stream == s && e->prof_chain) {
        ev.a = e->prof_ev.back().b; // shared boundary
        ev.owns_a = false;
    } else {
        ev.a = prof_new_event(e);
        if (!ev.a) return;
        ev.owns_a = true;
        hipEventRecord(ev.a, s);
    }
    e->prof_ev.push_back(ev);
}

void orbx_prof_end(orbx_extractor *e, hipStream_t s)
{
    if (!e->prof || e->prof_ev.empty() || e->prof_ev.back().b) return;
    hipEvent_t b = prof_new_event(e);
    if (!b) return;
    hipEventRecord(b, s);
    e->prof_ev.back().b = b;
    e->prof_last_stream = s;
    e->prof_chain = true;
}

extern "C" int orbx_profile_enable(orbx_extractor *e, int enable)
{
    if (!e) { orbx_set_error("null extractor"); return ORBX_E_INVALID; }
    e->prof = enable != 0;
    e->prof_chain = false;
    return ORBX_OK;
}

extern "C" int orbx_profile_stages(orbx_extractor *e, unsigned stage_mask)
{
    if (!e) { orbx_set_error("null extractor"); return ORBX_E_INVALID; }
    e->prof_mask = stage_mask;
    e->prof_chain = false;
    return ORBX_OK;
}

extern "C" int orbx_profile_read(orbx_extractor *e, float *ms, int *launches, int reset)
{
    if (!e) { orbx_set_error("null extractor"); return ORBX_E_INVALID; }
    ORBX_HIP(hipSetDevice(e->device));
    ORBX_HIP(hipDeviceSynchronize());
    for (auto &ev : e->prof_ev) {
        float t = 0;
        if (ev.a && ev.b && hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) { e->prof_ms[ev.stage] += t; e->prof_n[ev.stage]++; }
    }
    for (auto &ev : e->prof_ev) { // events go back to the pool (a shared boundary is owned by the earlier launch as its b)
        if (ev.owns_a && ev.a) e->prof_pool.push_back(ev.a);
        if (ev.b) e->prof_pool.push_back(ev.b);
    }
    e->prof_ev.clear();
    e->prof_chain = false;
    for (int i = 0; i < ORBX_STAGE_COUNT; i++) {
        if (ms) ms[i] = e->prof_ms[i];
        if (launches) launches[i] = e->prof_n[i];
        if (reset) { e->prof_ms[i] = 0; e->prof_n[i] = 0; }
    }
    return ORBX_OK;
}
