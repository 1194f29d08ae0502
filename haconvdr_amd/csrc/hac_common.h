// Shared host-side helpers for the haconvdr C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/haconvdr.h"

namespace hac {

typedef unsigned long long u64;
typedef unsigned int u32;

// thread-local error slot behind hac_last_error()
std::string &last_error_slot();
int fail(int code, const char *fmt, ...);

#define HAC_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return hac::fail(_e == hipErrorOutOfMemory ? HAC_ERR_OOM : HAC_ERR_HIP,             \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                             __LINE__);                                                         \
    } while (0)

#define HAC_TRY(expr)                 \
    do {                              \
        int _rc = (expr);             \
        if (_rc != HAC_OK) return _rc; \
    } while (0)

// RAII device guard: the library never leaves the caller's current device changed.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// device buffer that only ever grows
struct GrowBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return HAC_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e != hipSuccess) return fail(HAC_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = want;
        // A NEW buffer starts out as zeros, whatever the allocator recycled (fresh pages, which is what a first search sees, are
        // zero; recycled ones are not).  This is hygiene, not a guarantee: a buffer that is REUSED rather than regrown still holds
        // what earlier calls left in it, so no kernel may read beyond a device-side count without masking (what fixed the hang a
        // soak run found in the device-decided fallback: gather_rows_kernel zeroes the rows behind the count, padded queries'
        // thresholds are NaN).  Growth is rare; the null-stream memset + wait keeps the fill ahead of every stream, which also
        // means that a call that has to GROW a workspace synchronizes once (include/haconvdr.h: warm up at the largest shape
        // before capturing a *_device call into a graph).
        if (hipMemset(p, 0, want) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(p);
            p = nullptr;
            cap = 0;
            return fail(HAC_ERR_HIP, "hipMemset(%zu) of a new workspace failed", want);
        }
        return HAC_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace hac
