// Optional in-library kernel timing with HIP events on the launching stream (bench.py's live roofline
// measurement): when enabled, every GEMM launch is bracketed by two events and its algorithmic FLOPs recorded.
#include "common.h"
#include "avllm_internal.h"
#include <vector>

namespace {
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;
    std::vector<double> flops;
    size_t used = 0;
} g_prof;
}

bool av_prof_enabled() { return g_prof.on && g_prof.used + 2 <= g_prof.ev.size(); }
void av_prof_before(hipStream_t st) { (void)hipEventRecord(g_prof.ev[g_prof.used], st); }
void av_prof_after(hipStream_t st, double flops) {
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
    g_prof.flops.push_back(flops);
    g_prof.used += 2;
}

extern "C" int avllm_profile_begin(int32_t max_launches) {
    AV_CHECK_ARG(max_launches > 0, "profile_begin: max_launches");
    while (g_prof.ev.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        // device-scope release: a default event makes every record a system-scope fence (cache write-back) between two kernels, ~5 us
        // of idle GPU per record and 818 bracketed launches per step
        AV_HIP(hipEventCreateWithFlags(&e, hipEventReleaseToDevice));
        g_prof.ev.push_back(e);
    }
    g_prof.used = 0;
    g_prof.flops.clear();
    g_prof.on = true;
    return AV_OK;
}

// pause / resume between steps without dropping what has been recorded: every bracketed launch costs ~6 us of idle GPU (two event
// records are two barrier packets), so the bench brackets a sample of its timed steps, not all of them
extern "C" int avllm_profile_enable(int32_t on) { g_prof.on = on != 0; return AV_OK; }

// out[0] = total GEMM milliseconds, out[1] = total algorithmic FLOPs, out[2] = launches, out[3] = launches dropped
extern "C" int avllm_profile_end(double* out) {
    AV_CHECK_ARG(out, "profile_end: null");
    g_prof.on = false;
    double ms = 0, fl = 0;
    const size_t n = g_prof.used / 2;
    if (n) AV_HIP(hipEventSynchronize(g_prof.ev[g_prof.used - 1]));
    for (size_t i = 0; i < n; ++i) {
        float t = 0.f;
        AV_HIP(hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        ms += t; fl += g_prof.flops[i];
    }
    out[0] = ms; out[1] = fl; out[2] = (double)n; out[3] = 0;
    return AV_OK;
}
