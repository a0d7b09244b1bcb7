// Error plumbing and misc entry points of libneat_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace neat {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return NEAT_OK;
    return set_error(NEAT_ERR_HIP, "%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
}

}  // namespace neat

extern "C" const char* neat_version(void) { return "neat_hip 0.1 (gfx950)"; }
extern "C" const char* neat_last_error(void) { return neat::g_err; }
