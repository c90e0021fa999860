// error.cpp -- last-error string and build info of libvfmreg_hip.so
#include "common.h"

#include <string.h>

static thread_local char g_err[512] = "";

char* vfm_err_buf() { return g_err; }

int vfm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

VFM_EXPORT const char* vfm_last_error(void) { return g_err; }
VFM_EXPORT const char* vfm_build_info(void) { return "vfmreg-hip gfx950 r1"; }
