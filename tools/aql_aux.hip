// aql_aux.hip -- helper kernels of tools/aql_probe.cpp (a bare gfx950 code object, loaded through ROCr).
#include <hip/hip_runtime.h>
#include <stdint.h>

// Which XCD does each workgroup run on?  out[block] = XCC_ID | HW_ID << 8  (one store per workgroup).
extern "C" __global__ void xcc_probe_kernel(uint32_t *out)
{
    uint32_t xcc_id, hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc_id & 0xfu) | (hw_id << 8);
}

// Last packet of a chain: publish `value` to a word another engine waits on (a hipStreamWaitValue64 of a HIP stream).  The
// packet before it released the chain's results at system scope; this one only stores.
extern "C" __global__ void store_u64_kernel(uint64_t *word, uint64_t value)
{
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
