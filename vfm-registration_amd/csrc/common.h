// common.h -- shared host-side helpers for the libvfmreg_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vfmreg.h"
#include "config.h"

#define VFM_EXPORT extern "C" __attribute__((visibility("default")))

// thread-local last-error string (the only global state of the library)
char *vfm_err_buf();
int vfm_fail(int code, const char *fmt, ...);

#define VFM_CHECK_ARG(cond, ...)                                  \
    do {                                                          \
        if (!(cond)) return vfm_fail(VFM_EINVAL, __VA_ARGS__);    \
    } while (0)

#define VFM_CHECK_LAUNCH(what)                                                             \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) return vfm_fail(VFM_EHIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

#define VFM_CHECK_HIP(call)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) return vfm_fail(VFM_EHIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

static inline size_t vfm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve consecutive 256-byte aligned regions out of a caller-provided workspace
struct VfmCarver {
    unsigned char *base;
    size_t off;
    explicit VfmCarver(void *p) : base(static_cast<unsigned char *>(p)), off(0) {}
    template <typename T>
    T *take(size_t count) {
        off = vfm_align_up(off, 256);
        T *r = reinterpret_cast<T *>(base ? base + off : nullptr);
        off += count * sizeof(T);
        return r;
    }
    size_t used() const { return vfm_align_up(off, 256); }
};
