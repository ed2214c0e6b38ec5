// philox_strip.hpp -- the step words of a wave tile through a wave-private LDS strip; LDS-only barrier (device code).
// [step-kernel-source-begin]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "board_ops.hpp"
#include "tiling.hpp"

namespace env2048 {

// Compute the step words of the wave tile starting at global environment env0 into `strip`
// (strip[k] = word of environment (env0 & ~3) + k).  Every lane of the wave must call this.
template <int BPL = BOARDS_PER_LANE>
__device__ __forceinline__ void fill_word_strip(uint32_t *strip, const RngKey &key, uint64_t env0, uint32_t lane)
{
    const uint64_t group0 = env0 >> 2;
    constexpr uint32_t GROUPS = tile_boards(BPL) / 4;  // Philox blocks an aligned tile needs
    uint32_t w0, w1, w2, w3;
    env_block(key, group0 + lane, STREAM_STEP, w0, w1, w2, w3);
    if (GROUPS == 64u) {
        reinterpret_cast<uint4 *>(strip)[lane] = make_uint4(w0, w1, w2, w3);
        if (env0 & 3u) {  // wave-uniform: the tile straddles a 65th block
            env_block(key, group0 + 64u, STREAM_STEP, w0, w1, w2, w3);
            if (lane == 0u) reinterpret_cast<uint4 *>(strip)[64] = make_uint4(w0, w1, w2, w3);
        }
    } else if (lane <= GROUPS) {  // spare lanes: the straddling block costs nothing extra
        reinterpret_cast<uint4 *>(strip)[lane] = make_uint4(w0, w1, w2, w3);
    }
    // The strip is private to this wave and a wave's LDS operations execute in order, so no
    // s_barrier is needed -- only a compiler-level fence so the cross-lane reads stay behind the writes.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// s_waitcnt vmcnt(0) only (expcnt / lgkmcnt left at their maxima): gfx9 encoding.
#define ENV2048_WAIT_ALL_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e.
// wait for the board loads that are deliberately still in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace env2048
// [step-kernel-source-end]
