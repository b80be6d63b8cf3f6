// Shared helpers of libevdnerf.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/evdnerf.h"

namespace evd {

// thread-local last-error message behind evd_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

#define EVD_HIP(call)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return evd::fail(EVD_E_HIP, "%s failed: %s (%s:%d)", #call,           \
                                               hipGetErrorString(e_), __FILE__, __LINE__);          \
    } while (0)

#define EVD_REQUIRE(cond, ...)                                                  \
    do {                                                                        \
        if (!(cond)) return evd::fail(EVD_E_INVALID, __VA_ARGS__);              \
    } while (0)

#define EVD_LAUNCH_CHECK() EVD_HIP(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Raise a kernel's dynamic-LDS limit once per device (the attribute is per device; one process may drive several).
// `done` is the caller's function-local bitmask of devices already configured.
#define EVD_SET_MAX_LDS(kernel, bytes)                                                                             \
    do {                                                                                                           \
        static std::atomic<unsigned long long> done_{0};                                                           \
        int dev_ = 0;                                                                                              \
        EVD_HIP(hipGetDevice(&dev_));                                                                              \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                       \
        if (!(done_.load(std::memory_order_relaxed) & bit_)) {                                                     \
            EVD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            done_.fetch_or(bit_, std::memory_order_relaxed);                                                       \
        }                                                                                                          \
    } while (0)

// developer switches read from the environment once per process
inline bool env_flag(const char* name) {
    const char* e = getenv(name);
    return e && e[0] && e[0] != '0';
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// activations: reference networks/nerf.py:31-33, networks/pdrf/voxnerf.py:28-30
__device__ __forceinline__ float act(int code, float x) {
    switch (code) {
    case EVD_ACT_RELU: return fmaxf(x, 0.f);
    case EVD_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case EVD_ACT_EXP: return expf(x);
    case EVD_ACT_SIGMOID1: return 1.002f / (expf(-x) + 1.f) - 0.001f;
    case EVD_ACT_SOFTPLUS: {
        float y = x - 1.f;
        return y > 20.f ? y : log1pf(expf(y));
    }
    case EVD_ACT_TANH: return tanhf(x);
    default: return x;
    }
}

// torch.linspace(start, end, steps)[i] (ATen RangeFactories: symmetric fill)
__host__ __device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// simple device buffer owned by a handle
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        bytes = n;
        EVD_HIP(hipMalloc(&p, n ? n : 1));
        return EVD_OK;
    }
    int upload(const void* host, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        EVD_HIP(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
        return EVD_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Side stream + join event of a handle's backward entry (the wgrad launches overlap the dgrad chain; nerf_train.h BwdPlan::side).
// PER-HANDLE state, created on first use and destroyed with the handle; the library keeps no process-wide table.  The mutex guards the
// CREATION only.  A backward entry is NOT re-entrant per handle: the one side stream and the one join event are shared by every call on
// that handle, so two host threads running evd_*_mlp_backward on the SAME handle at once would interleave their fork / join records.
// Concurrent calls on DIFFERENT handles (one handle per device / per model, the intended use) are independent.
struct SideStream {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    int get(hipStream_t* s, hipEvent_t* e) {
        std::lock_guard<std::mutex> lock(mu);
        if (!stream) {
            hipStream_t ns = nullptr;
            hipEvent_t ne = nullptr;
            EVD_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
            hipError_t er = hipEventCreateWithFlags(&ne, hipEventDisableTiming);
            if (er != hipSuccess) {
                (void)hipStreamDestroy(ns);
                return fail(EVD_E_HIP, "hipEventCreateWithFlags failed: %s", hipGetErrorString(er));
            }
            ev = ne;
            stream = ns;
        }
        *s = stream;
        *e = ev;
        return EVD_OK;
    }
    void release() {
        std::lock_guard<std::mutex> lock(mu);
        if (ev) (void)hipEventDestroy(ev);
        if (stream) (void)hipStreamDestroy(stream);
        ev = nullptr;
        stream = nullptr;
    }
};
// number of persistent wgrad workgroups next to the dgrad chain (EVD_BWD_OVERLAP = 0: no side stream; N: that many, at most `cap`)
inline int bwd_overlap_blocks(int cap) {
    static const int want = [] { const char* e = getenv("EVD_BWD_OVERLAP"); return e ? atoi(e) : 192; }();
    return want <= 0 ? 0 : (want > cap ? cap : want);
}

// TEST HOOK (tests/test_gpu_dist.py), defined once in evd_api.hip: EVD_TEST_SIDE_SPIN_US=N makes every wgrad launch on a handle's side
// stream start N microseconds late (a spin kernel in front of it), so that a missing join of the side stream into the caller's stream --
// the edge the early gradient all-reduce of dist.GradReducer.attach relies on -- shows as wrong sums instead of passing by luck of
// timing.  0 / unset: one predictable branch per backward entry, no launch.  evd_debug_side_spin_count() reports how many spin kernels were
// launched (the test asserts that the hook fired); EVD_TEST_SKIP_SIDE_JOIN=1 is the test's NEGATIVE control: the entry returns without
// joining the side stream, and the delayed launches must then show as wrong gradients.
int test_side_spin(hipStream_t side);          // EVD_OK, or the launch error
bool test_skip_side_join();

}  // namespace evd
