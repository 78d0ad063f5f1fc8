// Version / error-string entry points of the C ABI.
#include "common.h"
#include <string.h>

namespace mr {
static thread_local char g_last_err[256] = "";
void set_last_hip_error(const char* msg) {
    strncpy(g_last_err, msg ? msg : "", sizeof(g_last_err) - 1);
    g_last_err[sizeof(g_last_err) - 1] = 0;
}
}  // namespace mr

extern "C" int mr_version(void) { return 100; }

extern "C" const char* mr_strerror(int code) {
    switch (code) {
        case MR_OK: return "ok";
        case MR_EINVAL: return "invalid argument";
        case MR_EALIGN: return "pointer or leading dimension misaligned";
        case MR_ELAUNCH: return "HIP launch/runtime error";
        case MR_EWS: return "workspace too small";
        case MR_EUNSUPPORTED: return "unsupported shape or mode";
        default: return "unknown error";
    }
}

extern "C" const char* mr_last_hip_error(void) { return mr::g_last_err; }
