// common.h - shared host-side helpers of libsoslam_ba (error plumbing, device buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "soslam_ba.h"

namespace soslam {

void set_last_error(const char* fmt, ...);

#define SOSLAM_HIP_CHECK(expr)                                                                      \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ::soslam::set_last_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SOSLAM_ERR_HIP;                                                                  \
        }                                                                                           \
    } while (0)

#define SOSLAM_CHECK(expr)                   \
    do {                                     \
        int _s = (expr);                     \
        if (_s != SOSLAM_OK) return _s;      \
    } while (0)

// Device allocation owned by a handle.  No copies, freed in the destructor.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0) count = 1;
        SOSLAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        return SOSLAM_OK;
    }
    int upload(const std::vector<T>& h, hipStream_t s)
    {
        SOSLAM_CHECK(alloc(h.size()));
        if (!h.empty()) SOSLAM_HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
        return SOSLAM_OK;
    }
    int zero(hipStream_t s)
    {
        if (n) SOSLAM_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
        return SOSLAM_OK;
    }
};

inline unsigned div_up(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

}  // namespace soslam
