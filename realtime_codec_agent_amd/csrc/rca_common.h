// Shared helpers for the HIP side of the C ABI (include/rca.h).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rca.h"

namespace rca {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define RCA_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return rca::fail(RCA_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define RCA_LAUNCH_CHECK()                                                                         \
    do {                                                                                           \
        hipError_t _e = hipGetLastError();                                                         \
        if (_e != hipSuccess)                                                                      \
            return rca::fail(RCA_ERR_HIP, "%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
    } while (0)

inline const rca_tensor_t* find_tensor(const rca_tensor_t* ts, int n, const std::string& name) {
    for (int i = 0; i < n; ++i)
        if (ts[i].name && name == ts[i].name) return &ts[i];
    return nullptr;
}

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return RCA_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(RCA_ERR_HIP, "hipMalloc(%zu) -> %s", want, hipGetErrorString(e));
        cap = want;
        return RCA_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace rca
