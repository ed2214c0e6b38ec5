// tools/anyorder_probe.hip -- does hipExtAnyOrderLaunch let a kernel start before its predecessor in the SAME stream has
// finished on gfx950?  (hip_ext.h says the flag is "not supported on AMD GFX9xx boards".)  Kernel A spins ~300 us and stamps
// its end; kernel B stamps its start.  B.start < A.end  <=>  the barrier bit was cleared.
// hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long *out, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void stamp_kernel(unsigned long long *out)
{
    if (threadIdx.x == 0) out[2] = __builtin_amdgcn_s_memrealtime();
}

int main()
{
    unsigned long long *d, h[3];
    CHECK(hipMalloc(&d, 24));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(d, 0, 24));
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, 30000ull);  // 100 MHz ticks: 300 us
            if (mode == 0)
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, d);
            else
                hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
            printf("%s: A ran %.1f us; B started %.1f us after A started (%s A ended)\n", mode ? "any-order " : "in-order  ",
                   (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0, h[2] < h[1] ? "BEFORE" : "after");
        }
    }
    return 0;
}
