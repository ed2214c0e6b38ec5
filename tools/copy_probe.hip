// tools/copy_probe.hip -- a plain streaming copy, the measuring stick bench.py prices the HBM-cold leg against beside the
// 8 TB/s spec peak ("the measured device-copy ceiling on the box, reported alongside").  16-byte vectors;
// the caller picks the shape (grid-stride with four loads in flight per lane, or one vector per lane; temporal or
// non-temporal accesses) and bench.py reports the best of a small sweep.  Measurement tool only: built into
// tools/libcopy_probe.so by __graft_entry__.build(), loaded by bench.py through ctypes, never by the product.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int BLOCK = 256;
constexpr int UNROLL = 4;
typedef uint32_t vec4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ vec4 ld(const vec4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(vec4 v, vec4 *p)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <bool NT>
__global__ __launch_bounds__(BLOCK) void copy_strided(const vec4 *__restrict__ src, vec4 *__restrict__ dst, uint64_t vectors)
{
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < vectors; i += UNROLL * stride) {
        vec4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) st<NT>(v[u], dst + i + u * stride);
    }
    for (; i < vectors; i += stride) st<NT>(ld<NT>(src + i), dst + i);
}

// every workgroup copies one contiguous tile of BLOCK * UNROLL vectors (16 KiB)
template <bool NT>
__global__ __launch_bounds__(BLOCK) void copy_tiles(const vec4 *__restrict__ src, vec4 *__restrict__ dst, uint64_t vectors)
{
    const uint64_t base = (uint64_t)blockIdx.x * BLOCK * UNROLL + threadIdx.x;
    vec4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
        if (base + u * BLOCK < vectors) v[u] = ld<NT>(src + base + u * BLOCK);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
        if (base + u * BLOCK < vectors) st<NT>(v[u], dst + base + u * BLOCK);
}

}  // namespace

// bytes a multiple of 16, src / dst 16-byte aligned device pointers.  workgroups > 0: grid-stride copy with that many
// workgroups; workgroups == 0: one 16 KiB tile per workgroup.  nontemporal != 0: non-temporal loads and stores.
// Returns the hipError_t of the launch.
extern "C" __attribute__((visibility("default"))) int copy_probe(void *stream, const void *src, void *dst, uint64_t bytes, uint32_t workgroups,
                                                                 int nontemporal)
{
    if (bytes % 16 || ((uintptr_t)src | (uintptr_t)dst) % 16) return (int)hipErrorInvalidValue;
    const uint64_t vectors = bytes / 16;
    const vec4 *s = (const vec4 *)src;
    vec4 *d = (vec4 *)dst;
    hipStream_t q = (hipStream_t)stream;
    if (workgroups) {
        if (nontemporal)
            hipLaunchKernelGGL(copy_strided<true>, dim3(workgroups), dim3(BLOCK), 0, q, s, d, vectors);
        else
            hipLaunchKernelGGL(copy_strided<false>, dim3(workgroups), dim3(BLOCK), 0, q, s, d, vectors);
    } else {
        const unsigned tiles = (unsigned)((vectors + BLOCK * UNROLL - 1) / (BLOCK * UNROLL));
        if (nontemporal)
            hipLaunchKernelGGL(copy_tiles<true>, dim3(tiles), dim3(BLOCK), 0, q, s, d, vectors);
        else
            hipLaunchKernelGGL(copy_tiles<false>, dim3(tiles), dim3(BLOCK), 0, q, s, d, vectors);
    }
    return (int)hipGetLastError();
}
