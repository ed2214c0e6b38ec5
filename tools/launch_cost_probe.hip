// tools/launch_cost_probe.hip -- host-side cost of enqueuing one kernel launch on gfx950 / ROCm 7.2, by API form:
// hipLaunchKernelGGL (what the library uses), hipModuleLaunchKernel with a kernelParams array, and with a pre-packed argument
// buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER); one stream and two alternating streams.  The kernel has the step kernel's
// argument shape (15 arguments, 96 bytes) and does nothing, so the figures are pure enqueue cost.
// hipcc --offload-arch=gfx950 -O2 tools/launch_cost_probe.hip -o tools/launch_cost_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void shaped_kernel(uint8_t *boards, const uint8_t *actions, uint64_t n, uint64_t env_base, uint32_t flags, uint32_t seed_lo,
                              uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi, int32_t *reward, uint8_t *done, uint8_t *legal, void *obs,
                              uint8_t *max_exp, unsigned long long *counter)
{
    if (n == 0xFFFFFFFFFFFFFFFFull) boards[0] = (uint8_t)(flags + seed_lo + seed_hi + step_lo + step_hi + env_base);
}

struct __attribute__((packed, aligned(8))) Packed {
    uint8_t *boards; const uint8_t *actions; uint64_t n, env_base; uint32_t flags, seed_lo, seed_hi, step_lo, step_hi; uint32_t pad;
    int32_t *reward; uint8_t *done; uint8_t *legal; void *obs; uint8_t *max_exp; unsigned long long *counter;
};

int main()
{
    uint8_t *buf;
    CHECK(hipMalloc(&buf, 4096));
    hipStream_t s[2];
    CHECK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipFunction_t fn;
    CHECK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void *>(shaped_kernel)));
    const int K = 4000;
    for (int streams = 1; streams <= 2; ++streams) {
        for (int form = 0; form < 3; ++form) {
            double best = 1e30;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < K; ++k) {
                    hipStream_t st = s[streams == 2 ? (k & 1) : 0];
                    uint64_t n = 1, base = 0;
                    uint32_t fl = 1, sl = 2, sh = 3, tl = (uint32_t)k, th = 0;
                    int32_t *rw = (int32_t *)buf; uint8_t *dn = buf, *lg = nullptr, *mx = nullptr; void *ob = nullptr;
                    unsigned long long *ct = nullptr; const uint8_t *ac = buf;
                    if (form == 0) {
                        hipLaunchKernelGGL(shaped_kernel, dim3(1), dim3(64), 0, st, buf, ac, n, base, fl, sl, sh, tl, th, rw, dn, lg, ob, mx, ct);
                    } else if (form == 1) {
                        void *params[] = {&buf, &ac, &n, &base, &fl, &sl, &sh, &tl, &th, &rw, &dn, &lg, &ob, &mx, &ct};
                        CHECK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, params, nullptr));
                    } else {
                        Packed p{buf, ac, n, base, fl, sl, sh, tl, th, 0, rw, dn, lg, ob, mx, ct};
                        size_t size = sizeof(p);
                        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
                        CHECK(hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, nullptr, extra));
                    }
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
                CHECK(hipDeviceSynchronize());
                CHECK(hipGetLastError());
                if (us < best) best = us;
            }
            printf("%d stream(s), %-40s %6.2f us per launch (host side, best of 5 x %d)\n", streams,
                   form == 0 ? "hipLaunchKernelGGL" : form == 1 ? "hipModuleLaunchKernel(kernelParams)" : "hipModuleLaunchKernel(packed buffer)", best, K);
        }
    }
    return 0;
}
