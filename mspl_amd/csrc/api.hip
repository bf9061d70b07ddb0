// Error plumbing and version string of libmspl_hip.so.
#include <stdarg.h>

#include "common.hpp"

namespace mspl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace mspl

extern "C" const char* mspl_version(void) { return "mspl_hip 0.3 (gfx950)"; }

extern "C" int mspl_abi_version(void) { return 3; }

extern "C" size_t mspl_last_error(char* buf, size_t cap) {
    const size_t n = strlen(mspl::g_err);
    if (buf && cap) {
        const size_t m = n < cap - 1 ? n : cap - 1;
        memcpy(buf, mspl::g_err, m);
        buf[m] = 0;
    }
    return n;
}
