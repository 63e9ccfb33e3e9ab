// Error plumbing and trivial queries of libwsu (see include/wsu.h).
#include "wsu_device.h"
#include <cstdarg>
#include <cstdio>

namespace {
thread_local char g_err[512] = "";
}

void wsu_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int wsu_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        wsu_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return WSU_ERR_HIP;
    }
    return WSU_OK;
}

extern "C" {
int wsu_version(void) { return WSU_VERSION; }
const char* wsu_last_error(void) { return g_err; }
int wsu_act_elem_size(int mode) { return mode == WSU_MODE_BF16 ? 2 : (mode == WSU_MODE_F16F8 ? 3 : (mode == WSU_MODE_F32 || mode == WSU_MODE_BF16X3 || mode == WSU_MODE_BF16X3S || mode == WSU_MODE_F16F8X ? 4 : 0)); }
}
