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

int ensure_dynamic_lds(const void* fn, int bytes, unsigned long long* done, const char* what) {
    int dev = 0;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return NEAT_OK;
    rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
    if (rc) return rc;
    __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);       // racing first calls set the same attribute twice: harmless
    return NEAT_OK;
}

}  // namespace neat

extern "C" const char* neat_version(void) { return "neat_hip 0.1 (gfx950)"; }
extern "C" const char* neat_last_error(void) { return neat::g_err; }
