// C ABI of libtargetdiff_hip.so, part 1 of 5: error string, ABI / build tags and the per-class kernel timers.
// See include/targetdiff_hip.h for the contract of every entry point and the reference seam it replaces.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "td_device.h"
#include "td_internal.h"
#include "td_api.h"

using namespace tdapi;

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void td_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *td_last_error(void) { return g_err; }
extern "C" int td_abi_version(void) { return TD_ABI_VERSION; }
#ifndef TD_BUILD_TAG
#define TD_BUILD_TAG "untagged"
#endif
extern "C" const char *td_build_tag(void) { return TD_BUILD_TAG; }

// ------------------------------------------------------------------------------------------ kernel timers
// Optional per-kernel-class HIP-event timers (bench.py's roofline leg): events are recorded on the launch
// stream around the selected classes; td_profile_end synchronises and sums hipEventElapsedTime.
namespace tdapi {
Profiler g_prof;
}  // namespace tdapi

extern "C" int td_profile_begin(uint32_t class_mask) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int c = 0; c < PC_COUNT; ++c) {
        for (hipEvent_t e : g_prof.ev[c]) g_prof.pool.push_back(e);
        g_prof.ev[c].clear();
    }
    g_prof.mask = class_mask;
    return TD_OK;
}

extern "C" int td_profile_end(float *ms_out, int32_t *count_out, int32_t num_classes) {
    g_prof.mask = 0;
    TD_CHECK_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int c = 0; c < PC_COUNT; ++c) {
        float total = 0.f;
        int n = 0;
        for (size_t i = 0; i + 1 < g_prof.ev[c].size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g_prof.ev[c][i], g_prof.ev[c][i + 1]) == hipSuccess) { total += ms; ++n; }
        }
        if (c < num_classes) {
            if (ms_out) ms_out[c] = total;
            if (count_out) count_out[c] = n;
        }
        for (hipEvent_t e : g_prof.ev[c]) g_prof.pool.push_back(e);
        g_prof.ev[c].clear();
    }
    return TD_OK;
}
