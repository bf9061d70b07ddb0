// Error plumbing and version string of libmspl_hip.so.
#include <stdarg.h>

#include <atomic>

#include "common.hpp"

namespace mspl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Launch-shape preference: 0 = one pass at a time (many small workgroups, shortest launch), 1 = several independent passes share
// the chip (fewer, longer workgroups: another pass fills the ramp and tail, so steady-state efficiency wins).  Read by the
// launchers at launch time, i.e. at hipGraph capture; never changes results.
std::atomic<int> g_throughput_mode{0};

}  // namespace mspl

extern "C" int mspl_set_throughput_mode(int32_t on) {
    const int prev = mspl::g_throughput_mode.exchange(on ? 1 : 0);
    return prev;
}

extern "C" const char* mspl_version(void) { return "mspl_hip 0.1 (gfx950)"; }

extern "C" size_t mspl_last_error(char* buf, size_t cap) {
    const size_t n = strlen(mspl::g_err);
    if (buf && cap) {
        const size_t m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, mspl::g_err, m);
        buf[m] = 0;
    }
    return n;
}
