// board_io.hpp -- how boards and observations travel between HBM and registers (device code, gfx950).
//
// Data layout in HBM: `boards` is an array of 16-byte records (one global_load_dwordx4 /
// global_store_dwordx4 per board, a wave moves 1 KiB contiguous); actions/done/legal are byte
// arrays, reward int32, obs 16 floats (or halves) per board.
// [step-kernel-source-begin]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "board_ops.hpp"
#include "tiling.hpp"

namespace env2048 {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ Board load_board(const uint8_t *boards, uint64_t i)
{
    const uint4 v = reinterpret_cast<const uint4 *>(boards)[i];
    return Board{v.x, v.y, v.z, v.w};
}

// Agent-scope board loads for the launches of a direct-dispatch chain (step_block<..., RESIDENT>): buffer_load_dwordx4 ... sc1
// goes past the CU's vector L1 to the XCD's L2.  (16-byte loads have no agent-scope form the compiler tracks other than the
// buffer intrinsic with cache-policy bit 4 = sc1 on gfx94x / gfx950; `nt` global loads measured 6 % slower on the
// write-through path and are not an architectural L1 bypass.)  The descriptor covers one wave's tile.
struct TileBuffer {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ TileBuffer make_tile_buffer(uint8_t *boards, uint64_t tile0, uint64_t tile_boards_here)
{
    uint8_t *base = boards + 16u * tile0;  // (wave-uniform; the read-first-lanes make that provable to the compiler)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(16u * tile_boards_here));
    return TileBuffer{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ Board load_board_agent(const TileBuffer &tile, uint32_t board_in_tile)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(tile.rsrc, (int)(16u * board_in_tile), 0, /*sc1*/ 16);
    return Board{v.x, v.y, v.z, v.w};
}

// Write-through (sc1) stores: the line is not left dirty in the XCD's L2, so the end-of-kernel L2
// write-back -- which sits on the critical path of back-to-back step launches -- has nothing to do
// (measured: -6 % per launch at 2^20 boards for the board stores alone, -12 % with reward/done).
// Only WIDE sc1 stores are cheap: narrow ones (dword / byte per lane) are one fabric write each.
// gfx9 hazard: a VMEM store of more than 64 bits must not be followed directly by a VALU write to its
// data VGPRs (the data is read one cycle after issue).  hipcc inserts that wait state for stores it
// emits itself but cannot see inside an asm statement, so it is part of the string (s_nop 1 = two
// wait states).  Without it boards were corrupted -- only under memory back-pressure (2^20 boards).
__device__ __forceinline__ void store_x4_through(void *ptr, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void store_x1_through(void *ptr, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}

__device__ __forceinline__ void store_board(uint8_t *boards, uint64_t i, const Board &bd)
{
    reinterpret_cast<uint4 *>(boards)[i] = make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
}

// Board store of the step kernel: write-through for launches small enough that the end-of-kernel L2
// write-back matters, ordinary write-back stores for HBM-bound sizes (where sc1 measured 5 % slower).
__device__ __forceinline__ void store_board(uint8_t *boards, uint64_t i, const Board &bd, bool through)
{
    if (through)
        store_x4_through(reinterpret_cast<uint4 *>(boards) + i, u32x4{bd.r0, bd.r1, bd.r2, bd.r3});
    else
        store_board(boards, i, bd);
}

// log2(tile)/16 per cell [ref: reinforce/game/core.py:346-377 encode_observation]; exact in f32.
__device__ __forceinline__ void store_obs(float *obs, uint64_t i, const Board &bd)
{
    float4 *o = reinterpret_cast<float4 *>(obs) + 4 * i;
    const uint32_t rows[4] = {bd.r0, bd.r1, bd.r2, bd.r3};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t x = rows[r];
        o[r] = make_float4((float)(x & 0xFFu) * 0.0625f, (float)((x >> 8) & 0xFFu) * 0.0625f,
                           (float)((x >> 16) & 0xFFu) * 0.0625f, (float)(x >> 24) * 0.0625f);
    }
}

// Coalesced f32 observation of 64 consecutive boards held one per lane: every lane drops its 16
// packed bytes into a wave-private 1 KiB LDS strip, reads back ONE dword (row l&3 of board l>>2 of
// each quarter) and converts it, so that each of the four store instructions writes 1 KiB of
// contiguous floats instead of 64 float4 that are 64 B apart.  `skip` lanes publish a sentinel (no
// real row is 0xFFFFFFFF) and nothing is stored for their boards (they are rebuilt later by the
// reset pass, possibly by another wave).  first = index of lane 0's board; boards >= n are skipped.
__device__ __forceinline__ void store_obs_coalesced(float *obs, uint32_t *strip, uint64_t first, uint64_t n, uint32_t lane,
                                                    const Board &bd, bool skip, bool through = false)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // earlier reads of the strip are done
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<uint4 *>(strip)[lane] = skip ? make_uint4(~0u, ~0u, ~0u, ~0u) : make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float4 *out = reinterpret_cast<float4 *>(obs) + 4 * first;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t x = strip[64u * k + lane];
        const uint64_t board = first + 16u * k + (lane >> 2);
        if (x != ~0u && board < n) {
            const float4 v = make_float4((float)(x & 0xFFu) * 0.0625f, (float)((x >> 8) & 0xFFu) * 0.0625f,
                                         (float)((x >> 16) & 0xFFu) * 0.0625f, (float)(x >> 24) * 0.0625f);
            if (through)
                store_x4_through(out + 64u * k + lane, u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z),
                                                              __float_as_uint(v.w)});
            else
                out[64u * k + lane] = v;
        }
    }
}

// 16-bit observations (ENV2048_FLAG_OBS_F16 / _BF16): the same values -- e/16 with e <= 31 has five
// significant bits, exact in both formats -- at half the bytes, which is what bounds the trainer-facing path.

__device__ __forceinline__ uint32_t obs_pair16(uint32_t e_lo, uint32_t e_hi, bool bf16)
{
    const float a = (float)e_lo * 0.0625f, b = (float)e_hi * 0.0625f;
    if (bf16) return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);  // exact: low 16 bits are zero
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));             // exact: no rounding happens
}
__device__ __forceinline__ uint2 obs_row16(uint32_t x, bool bf16)
{
    return make_uint2(obs_pair16(x & 0xFFu, (x >> 8) & 0xFFu, bf16), obs_pair16((x >> 16) & 0xFFu, x >> 24, bf16));
}

__device__ __forceinline__ void store_obs16(void *obs, uint64_t i, const Board &bd, bool bf16)
{
    uint4 *o = reinterpret_cast<uint4 *>(obs) + 2 * i;
    const uint2 r0 = obs_row16(bd.r0, bf16), r1 = obs_row16(bd.r1, bf16), r2 = obs_row16(bd.r2, bf16), r3 = obs_row16(bd.r3, bf16);
    o[0] = make_uint4(r0.x, r0.y, r1.x, r1.y);
    o[1] = make_uint4(r2.x, r2.y, r3.x, r3.y);
}

// The coalesced form for 16-bit observations: a board is 32 B, so a lane reads back TWO row dwords of the
// strip (rows 2l, 2l+1 of each half) and each of the two store instructions writes 1 KiB contiguous.
__device__ __forceinline__ void store_obs16_coalesced(void *obs, uint32_t *strip, uint64_t first, uint64_t n, uint32_t lane,
                                                      const Board &bd, bool skip, bool through, bool bf16)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<uint4 *>(strip)[lane] = skip ? make_uint4(~0u, ~0u, ~0u, ~0u) : make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint4 *out = reinterpret_cast<uint4 *>(obs) + 2 * first;
#pragma unroll
    for (uint32_t k = 0; k < 2u; ++k) {
        const uint2 x = reinterpret_cast<const uint2 *>(strip)[64u * k + lane];
        const uint64_t board = first + 32u * k + (lane >> 1);
        if (x.x != ~0u && board < n) {
            const uint2 lo = obs_row16(x.x, bf16), hi = obs_row16(x.y, bf16);
            if (through)
                store_x4_through(out + 64u * k + lane, u32x4{lo.x, lo.y, hi.x, hi.y});
            else
                out[64u * k + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

// Observation of one board / of a wave's 64 boards in the format the flags ask for.
__device__ __forceinline__ void store_obs_any(void *obs, uint64_t i, const Board &bd, uint32_t fmt)
{
    if (fmt == OBS_F32)
        store_obs(reinterpret_cast<float *>(obs), i, bd);
    else
        store_obs16(obs, i, bd, fmt == OBS_BF16);
}
__device__ __forceinline__ void store_obs_any_coalesced(void *obs, uint32_t *strip, uint64_t first, uint64_t n, uint32_t lane,
                                                        const Board &bd, bool skip, bool through, uint32_t fmt)
{
    if (fmt == OBS_F32)
        store_obs_coalesced(reinterpret_cast<float *>(obs), strip, first, n, lane, bd, skip, through);
    else
        store_obs16_coalesced(obs, strip, first, n, lane, bd, skip, through, fmt == OBS_BF16);
}

}  // namespace env2048
// [step-kernel-source-end]
