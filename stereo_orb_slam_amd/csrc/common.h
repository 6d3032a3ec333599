// common.h - shared host-side helpers of libsoslam_ba (error plumbing, device buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
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

// Device allocation owned by a handle.  No copies, freed in the destructor.  alloc() keeps an existing allocation that
// is large enough: a handle that is given one window after another (the reference's per-frame / sliding-window
// schedule) stops calling hipMalloc once it has seen its largest window.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;     // elements in use
    size_t cap = 0;   // elements allocated
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        cap = 0;
    }
    int alloc(size_t count)
    {
        if (p && cap >= count) { n = count; return SOSLAM_OK; }
        release();
        size_t want = count + count / 4;   // head room: consecutive windows differ by a few percent
        if (want == 0) want = 1;
        SOSLAM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T)));
        // SOSLAM_POISON_ALLOC=1 (development): fresh floating-point allocations are filled with 0xFF bytes (NaN), so that
        // any read of memory the library did not write shows up in the tests instead of depending on what the allocator
        // happened to hand out.  Index buffers are zeroed: a poisoned index would fault the GPU instead of failing a test.
        static const bool poison = std::getenv("SOSLAM_POISON_ALLOC") != nullptr;
        if (poison) {
            SOSLAM_HIP_CHECK(hipMemset(p, std::is_floating_point<T>::value ? 0xFF : 0x00, want * sizeof(T)));
            SOSLAM_HIP_CHECK(hipDeviceSynchronize());   // the fill runs on the null stream; the handles' streams do not wait for it
        }
        n = count;
        cap = want;
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
    void swap(DevBuf& o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
    }
};

inline unsigned div_up(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

}  // namespace soslam
