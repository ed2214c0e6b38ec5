// device_scope.hpp -- HIP-side helpers of the C-ABI entry points: device selection, launch geometry, error mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "errors.hpp"

namespace env2048 {
namespace host {

// Store policy of the write-dominated kernels (chance outcomes, one-hot encoding, unpacking): an output larger than the 256 MiB
// Infinity Cache is a stream nobody on the chip reads back soon -- non-temporal stores hold 6.1 - 6.6 TB/s there where plain
// stores drop to 3.9 (profiles/r04_gg_*); a smaller one stays in the cache and plain stores are the faster ones.
// ENV2048_EXP_STREAM_STORES=1 / 0 forces either (A/B measurements).
inline bool stream_output(uint64_t bytes)
{
    static const int forced = [] {
        const char *e = getenv("ENV2048_EXP_STREAM_STORES");
        return e ? (e[0] == '1' ? 1 : 0) : -1;
    }();
    return forced >= 0 ? forced == 1 : bytes > (256ull << 20);
}

// Write-through (sc1) stores for the 16-byte outputs of the kernels that write boards / observations (everything but the
// step kernel, which has its own size rule): nothing is left dirty in the XCD L2s for the end-of-kernel write-back.
// ENV2048_EXP_THROUGH=0 turns them off (A/B measurements).
inline bool wide_stores_through()
{
    static const bool on = [] {
        const char *e = getenv("ENV2048_EXP_THROUGH");
        return !(e && e[0] == '0');
    }();
    return on;
}

inline int hip_fail(hipError_t e, const char *what)
{
    return fail(ENV2048_ERR_HIP_BASE - (int)e, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
}

// Select `device` for the duration of one entry point and put the caller's device back after.  The device
// count is read once per process (it cannot change under a running HIP runtime); per call only the calling
// thread's current device is queried, and nothing is switched when it already is the requested one.
inline int visible_devices(hipError_t *err)
{
    static hipError_t cached_err = hipSuccess;
    static const int cached = [] {
        int count = 0;
        cached_err = hipGetDeviceCount(&count);
        return cached_err == hipSuccess ? count : 0;
    }();
    *err = cached_err;
    return cached;
}

struct DeviceScope {
    int prev = -1;
    int rc = ENV2048_OK;
    explicit DeviceScope(int device)
    {
        hipError_t e = hipSuccess;
        const int count = visible_devices(&e);
        if (count <= 0) {
            rc = fail(ENV2048_ERR_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
            return;
        }
        if (device < 0 || device >= count) {
            rc = fail(ENV2048_ERR_DEVICE, "device %d out of range [0, %d)", device, count);
            return;
        }
        e = hipGetDevice(&prev);
        if (e != hipSuccess) {
            rc = hip_fail(e, "hipGetDevice");
            prev = -1;
            return;
        }
        if (prev != device) {
            e = hipSetDevice(device);
            if (e != hipSuccess) {
                rc = hip_fail(e, "hipSetDevice");
                prev = -1;
            }
        } else {
            prev = -1;
        }
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }
inline dim3 rollout_grid(uint64_t n, int bpl) { return dim3((unsigned)((n + block_boards(bpl) - 1) / block_boards(bpl))); }

inline int launched(const char *what)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ENV2048_OK : hip_fail(e, what);
}

}  // namespace host
}  // namespace env2048
