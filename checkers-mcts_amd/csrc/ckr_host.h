// ckr_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/ckr.h"

namespace ckr {

char* last_error_buf();                 // thread-local, 512 bytes (ckr_rules.hip)
int   fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define CKR_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::ckr::fail(_e == hipErrorOutOfMemory ? CKR_ERR_OOM : CKR_ERR_HIP,     \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                               __FILE__, __LINE__);                                       \
    } while (0)

inline int require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(CKR_ERR_NO_DEVICE, "no HIP device visible: libckr has no CPU fallback");
    }
    return CKR_OK;
}

}  // namespace ckr
