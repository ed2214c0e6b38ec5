// env2048_kernels.hip -- HIP kernels (gfx950) and the C-ABI of include/env2048.h.
//
// Data layout in HBM: `boards` is an array of 16-byte records (one global_load_dwordx4 /
// global_store_dwordx4 per board, a wave moves 1 KiB contiguous); actions/done/legal are byte
// arrays, reward int32, obs 16 floats per board.  All board arithmetic happens in VGPRs
// (board_ops.hpp); LDS only carries the direction-selector table, the redistributed Philox words
// and the reset queue.  No MFMA -- this path is integer/indexing work bounded by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "env2048.h"
#include "board_ops.hpp"

using namespace env2048;

namespace {

#ifndef ENV2048_BLOCK
#define ENV2048_BLOCK 512
#endif
constexpr int BLOCK = ENV2048_BLOCK;

// [step-kernel-source-begin]  (bench.py hashes the text between these markers + board_ops.hpp to tell whether the committed
// rocprofv3 counters in profiles/step_pmc.json still describe this step kernel)
__device__ __forceinline__ Board load_board(const uint8_t *boards, uint64_t i)
{
    uint4 v = reinterpret_cast<const uint4 *>(boards)[i];
    return Board{v.x, v.y, v.z, v.w};
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
constexpr uint32_t OBS_F32 = 0u, OBS_BF16 = 2u;  // flags bits 2..3 (1 = f16)
__host__ __device__ __forceinline__ uint32_t obs_format(uint32_t flags) { return (flags >> 2) & 3u; }

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

// ------------------------------------------------------------------------------------------
// Hot kernels: step and rollout.
//
// Work decomposition (large tiling; the step kernel also exists with ONE board per lane for small launches, see
// tile_boards() below): a wave owns a TILE of 256 consecutive boards and each lane handles four of
// them, board j of lane l being tile board 64*j + l -- so every board load/store is one fully
// coalesced 1 KiB wave access (global_load/store_dwordx4) and reward/done/action accesses are
// dense too.  A block is 8 waves = 2048 boards.
//
// RNG: one Philox4x32-10 block yields the step words of FOUR neighbouring environments
// (env >> 2 selects the block, env & 3 the word), so a wave needs 64 blocks (65 when its first
// environment index is not a multiple of 4) for its 256 boards: lane l computes block
// (env0 >> 2) + l and the words are redistributed through a wave-private LDS strip
// (ds_write_b128 + 4 x ds_read_b32), which costs no VALU work.
//
// Direction handling: the v_perm_b32 selector records of board_ops.hpp live in LDS; a lane
// fetches the record of its action with two ds_read_b128.
//
// Auto-reset: terminal boards are rare (~0.7 % of steps in random play), so instead of every
// wave paying a second Philox block whenever ANY of its lanes finished a game, finished boards
// are queued in LDS and the LAST wave of the block to finish rebuilds all of them (no second barrier).
// ------------------------------------------------------------------------------------------
constexpr int WAVES_PER_BLOCK = BLOCK / 64;
#ifndef ENV2048_BOARDS_PER_LANE
#define ENV2048_BOARDS_PER_LANE 4
#endif
constexpr int BOARDS_PER_LANE = ENV2048_BOARDS_PER_LANE;  // the large-launch tiling (and the rollout kernels')
// Tile geometry as a function of the boards each lane handles.  The step kernel exists in two tilings: BPL = 4 (one
// Philox block feeds four boards of a lane's neighbourhood, constants and selector fetches amortise: best from 2^19
// boards up, where VALU issue is the limit) and BPL = 1 (a wave's dependent chain is one board long and four times
// as many CUs get work: 4.2 vs 5.2 us per step at 4 096 ... 262 144 boards, where latency is the limit).
constexpr int tile_boards(int bpl) { return 64 * bpl; }                                  // per wave
constexpr int block_boards(int bpl) { return tile_boards(bpl) * WAVES_PER_BLOCK; }       // per block
constexpr int word_strip(int bpl)                                                        // uint32 words of a wave's LDS strip:
{                                                                                        // Philox blocks (+1 straddling, padded to
    return (tile_boards(bpl) + 4 + 4 + tile_boards(bpl) / 4 + 3) / 4 * 4;                // 16 B) + a byte per board, >= 256 words
}                                                                                        // where the obs transposition uses it
constexpr int word_strip_words(int bpl) { return word_strip(bpl) < 256 ? 256 : word_strip(bpl); }
constexpr uint64_t SMALL_TILING_MAX_BOARDS = 1ull << 18;  // launches up to this size use the BPL = 1 tiling

__constant__ uint32_t DIR_TABLE[32] = {ENV2048_DIR_SELECTOR_VALUES};
// plane-space transitions of the multi-step kernels (board_ops.hpp, transition_oriented): 16 selector records
// [orientation][action] followed by 4 rank-mask records [action], 16 bytes each
__constant__ uint32_t ORIENT_TABLE[80] = {ENV2048_ORIENT_SELECTOR_VALUES, ENV2048_ORIENT_RANK_MASK_VALUES};

constexpr uint32_t INTERNAL_FLAG_WIDE_OUTPUTS = 0x80000000u;  // reward 16-B / done 4-B aligned: transposed wide stores
constexpr uint32_t INTERNAL_FLAG_WRITE_THROUGH = 0x40000000u;  // sc1 stores (host decides by launch size)

struct StepArgs {
    uint64_t n;
    uint64_t env_base;
    uint8_t *boards;
    const uint8_t *actions;
    int32_t *reward;
    uint8_t *done;
    uint8_t *legal;
    void *obs;  // f32, or f16 / bf16 with ENV2048_FLAG_OBS_F16 / _BF16
    RngKey key;
    uint32_t flags;
    uint8_t *max_exp;  // log2(max tile) of the board after move + spawn, BEFORE an auto-reset replaces it
};

__device__ __forceinline__ DirSelectors load_selectors(const uint4 *dir_lds, uint32_t action)
{
    const uint4 f = dir_lds[2u * action], i = dir_lds[2u * action + 1u];
    return DirSelectors{f.x, f.y, f.z, f.w, i.x, i.y, i.z, i.w};
}

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

// Rebuild the queued (terminal) boards of this block; executed by `nthreads` threads with ids `t`.
template <bool EXTRAS>
__device__ __forceinline__ void reset_queued(const StepArgs &a, uint64_t block0, const uint16_t *reset_slot,
                                             uint32_t n_reset, uint32_t t, uint32_t nthreads, bool through)
{
    for (uint32_t q = t; q < n_reset; q += nthreads) {
        const uint64_t i = block0 + reset_slot[q];
        uint32_t w0, w1, w2, w3;
        env_block(a.key, a.env_base + i, STREAM_RESET, w0, w1, w2, w3);
        const Board fresh = fresh_board(w0, w1, w2, w3);
        store_board(a.boards, i, fresh, through);  // same policy as the step's own stores: no dirty L2 lines left
        if (EXTRAS) {
            if (a.legal) a.legal[i] = (uint8_t)legal_mask(fresh);
            if (a.obs) store_obs_any(a.obs, i, fresh, obs_format(a.flags));
        }
    }
}

// The step kernel.  Order of events in a wave -- it matters, because vmcnt counts loads AND stores in
// issue order, so a wait for a load that sits behind a store also waits for that store's ack:
//   1. issue the selector-table load (8 lanes) and the board/action loads of the FIRST half of the tile
//   2. 64 (65) Philox blocks while those are in flight; words -> wave-private LDS strip -> registers
//   3. issue the loads of the SECOND half (every wave has queued its first half by now, so the chip-wide
//      load burst is served board-major); selector table -> LDS; LDS-only workgroup barrier
//   4. wait for the first half (exact vmcnt: nothing but loads has been issued), compute it, results in
//      registers; wait for the second half; store the first half; compute + store the second half
//   5. queue terminal boards; the last wave of the block to finish rebuilds them (no second barrier)
// EXTRAS: legal and/or obs requested.  FULL: every board of the block exists (no bounds checks);
// RANDOM: actions come from the step words instead of memory.  Build-time knobs (defaults measured
// best): ENV2048_BLOCK threads per block, ENV2048_BOARDS_PER_LANE; diagnostic builds
// ENV2048_EXP_ABLATE (bit 0 no game logic, bit 1 no Philox, bit 2 no reward/done stores, bit 3 terminal
// boards queued but not rebuilt) and ENV2048_EXP_STAMPS (five s_memrealtime stamps + HW_ID per wave into
// the obs buffer) exist for tools/microbench only.
template <bool EXTRAS, bool FULL, bool RANDOM, int BOARDS_PER_LANE>
__device__ __forceinline__ void step_block(const StepArgs &a, uint4 *dir_lds, uint32_t (*word_lds)[word_strip_words(BOARDS_PER_LANE)],
                                           uint32_t *queue_state, uint16_t *reset_slot)
{
    constexpr int TILE_BOARDS = tile_boards(BOARDS_PER_LANE), BLOCK_BOARDS = block_boards(BOARDS_PER_LANE);
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t block0 = (uint64_t)blockIdx.x * BLOCK_BOARDS;  // first board of this block (local index)
    const uint64_t tile0 = block0 + (uint64_t)wave * TILE_BOARDS;
    const uint64_t env0 = a.env_base + tile0;
    const bool auto_reset = (a.flags & ENV2048_FLAG_AUTO_RESET) != 0u;
    const bool through = (a.flags & INTERNAL_FLAG_WRITE_THROUGH) != 0u;
    // the LDS transposition of reward/done only pays together with write-through stores (measured)
    const bool wide_outputs = through && FULL && BOARDS_PER_LANE >= 2 && (a.flags & INTERNAL_FLAG_WIDE_OUTPUTS) != 0u;

#ifdef ENV2048_EXP_STAMPS  // diagnostic build: wave timeline stamps into the (otherwise unused) obs buffer
    uint64_t stamp0 = __builtin_amdgcn_s_memrealtime(), stamp1 = 0, stamp2 = 0, stamp3 = 0, stamp4 = 0;
#endif
    // 1. loads of the FIRST half of the tile (boards 0,1) -- and the selector table
    uint4 table_row = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 8u) table_row = reinterpret_cast<const uint4 *>(DIR_TABLE)[tid];
    Board bd[BOARDS_PER_LANE];
    uint32_t act[BOARDS_PER_LANE];
    constexpr int HALF = BOARDS_PER_LANE / 2 > 0 ? BOARDS_PER_LANE / 2 : 1;
    auto issue_loads = [&](int j) {
        const uint64_t i = tile0 + 64u * j + lane;
        bd[j] = Board{0u, 0u, 0u, 0u};
        act[j] = 0u;
        if (FULL || i < a.n) {
            bd[j] = load_board(a.boards, i);
            if (!RANDOM) act[j] = a.actions[i];
        }
    };
#pragma unroll
    for (int j = 0; j < HALF; ++j) issue_loads(j);

    // 2. step words (pure ALU + LDS) ...
    uint32_t *strip = word_lds[wave];
    const uint32_t strip_off = (uint32_t)env0 & 3u;
    uint32_t word[BOARDS_PER_LANE];
#if defined(ENV2048_EXP_ABLATE) && (ENV2048_EXP_ABLATE & 2)  // timing-only build: no Philox
#pragma unroll
    for (int j = 0; j < BOARDS_PER_LANE; ++j) word[j] = (lane * 2654435761u) ^ (a.key.step_lo * 40503u + j);
#else
    fill_word_strip<BOARDS_PER_LANE>(strip, a.key, env0, lane);
#pragma unroll
    for (int j = 0; j < BOARDS_PER_LANE; ++j) word[j] = strip[strip_off + 64u * j + lane];
#endif

    // ... and only now the loads of the SECOND half: by now every wave of the chip has queued its
    // first half, so the memory system delivers the tile board-major and the first half is home
    // after half of the load burst instead of all of it.
#pragma unroll
    for (int j = HALF; j < BOARDS_PER_LANE; ++j) issue_loads(j);

    // 3. selector table (its load was issued first, so it has long arrived) + bookkeeping words
    if (tid < 8u) dir_lds[tid] = table_row;
    if (tid == 0u) {
        *queue_state = 0u;  // bits 0-15: queued terminal boards, bits 16+: waves of the block that have finished
    }
    lds_barrier();
    // Nothing that touches a loaded value may be scheduled above the barrier: hipcc otherwise hoists the
    // selector address of board 0 (action << 5) there, the vmcnt wait goes with it, and every wave of the
    // block then waits at the barrier for the slowest wave's boards.
    __builtin_amdgcn_sched_barrier(0);

    // 4. first half home (exact count: vmcnt counts in issue order and no store has been issued
    //    yet; in the bounds-checked path loads may be skipped, so wait for everything there)
    if (FULL && BOARDS_PER_LANE >= 2)
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((RANDOM ? 1 : 2) * (BOARDS_PER_LANE - HALF)));  // vmcnt(#second-half loads)
    else
        ENV2048_WAIT_ALL_VMEM();
#ifdef ENV2048_EXP_STAMPS
    stamp1 = __builtin_amdgcn_s_memrealtime();
#endif
    // fetch the selector records of all boards now: one LDS round trip instead of one per board
    DirSelectors sel[BOARDS_PER_LANE];
#pragma unroll
    for (int j = 0; j < HALF; ++j) sel[j] = load_selectors(dir_lds, (RANDOM ? word[j] : act[j]) & 3u);

    uint32_t reward[BOARDS_PER_LANE];
    bool done[BOARDS_PER_LANE];
    auto compute = [&](int j) {
        bool changed;
#if defined(ENV2048_EXP_ABLATE) && (ENV2048_EXP_ABLATE & 1)  // timing-only build: no game logic
        reward[j] = sel[j].fA + word[j];
        changed = false;
        done[j] = (bd[j].r0 & 0x10000000u) != 0u;
        bd[j].r1 ^= sel[j].iD;
#else
        transition(bd[j], sel[j], word[j], reward[j], changed, done[j]);
#endif
    };
    auto emit = [&](int j) {
        const uint64_t i = tile0 + 64u * j + lane;
        if (EXTRAS && (a.flags & ENV2048_FLAG_REWARD_LOG2_F32))  // wave-uniform: the reward array then holds float32
            reward[j] = __float_as_uint(normalized_reward(reward[j]));
        if (FULL || i < a.n) {
            if (EXTRAS && a.max_exp) a.max_exp[i] = (uint8_t)max_exponent(bd[j]);  // of a finished game: its last board
            const bool pending_reset = auto_reset && done[j];
            if (pending_reset) {
                const uint32_t slot = atomicAdd(queue_state, 1u) & 0xFFFFu;
                reset_slot[slot] = (uint16_t)(wave * TILE_BOARDS + 64u * j + lane);
            } else {
                store_board(a.boards, i, bd[j], through);
                if (EXTRAS && a.legal) a.legal[i] = (uint8_t)legal_mask(bd[j]);
                if (EXTRAS && !FULL && a.obs) store_obs_any(a.obs, i, bd[j], obs_format(a.flags));
            }
#if defined(ENV2048_EXP_ABLATE) && (ENV2048_EXP_ABLATE & 4)  // timing-only build: no reward/done stores
            asm volatile("" ::"v"(reward[j]), "v"((uint32_t)done[j]));
#else
            if (!wide_outputs) {  // narrow write-once outputs: streaming (nt) stores, not sc1 (see store_x4_through)
                if (a.reward) __builtin_nontemporal_store((int32_t)reward[j], &a.reward[i]);
                __builtin_nontemporal_store((uint8_t)(done[j] ? 1 : 0), &a.done[i]);
            }
#endif
        }
        if (EXTRAS && FULL && a.obs)  // (the step words were copied to registers: the strip is free)
            store_obs_any_coalesced(a.obs, strip, tile0 + 64u * j, a.n, lane, bd[j], auto_reset && done[j], through,
                                    obs_format(a.flags));
    };

    // first half: compute only -- its stores wait until the second half's loads are home, so that
    // no vmcnt wait ever has a store (whose ack takes a memory round trip) in front of it
#pragma unroll
    for (int j = 0; j < HALF; ++j) compute(j);
#ifdef ENV2048_EXP_STAMPS
    asm volatile("" ::"v"(bd[0].r0), "v"(bd[HALF - 1].r3), "v"(reward[0]), "v"(reward[HALF - 1]));
    stamp3 = __builtin_amdgcn_s_memrealtime();
#endif
    ENV2048_WAIT_ALL_VMEM();
#ifdef ENV2048_EXP_STAMPS
    stamp4 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int j = HALF; j < BOARDS_PER_LANE; ++j) sel[j] = load_selectors(dir_lds, (RANDOM ? word[j] : act[j]) & 3u);
#pragma unroll
    for (int j = 0; j < HALF; ++j) emit(j);
#pragma unroll
    for (int j = HALF; j < BOARDS_PER_LANE; ++j) {
        compute(j);
        emit(j);
    }
    // 5a. No second barrier: waves retire as they finish.  Each announces itself with ONE LDS atomic on the
    // word that also counts the queue entries (its own entries were queued before, and a wave's LDS
    // operations execute in order), so the LAST wave of the block learns from the returned value both that
    // it is last and how many boards are queued.  Issued before the output transposition below, whose
    // LDS round trips hide its latency.
    uint32_t state_at_arrival = 0u;
    if (auto_reset) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0u) state_at_arrival = atomicAdd(queue_state, 0x10000u);
    }
    // Wide write-once outputs: transpose the tile's 256 rewards / done bytes through the (now free) LDS
    // strip so that lane l owns boards 4l..4l+3 and writes one dwordx4 (+ one dword) write-through.
#if !(defined(ENV2048_EXP_ABLATE) && (ENV2048_EXP_ABLATE & 4))
    if (wide_outputs) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint8_t *done_bytes = reinterpret_cast<uint8_t *>(strip + TILE_BOARDS + 8);
#pragma unroll
        for (int j = 0; j < BOARDS_PER_LANE; ++j) {
            strip[64u * j + lane] = reward[j];
            done_bytes[64u * j + lane] = done[j] ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (BOARDS_PER_LANE == 4 || lane < (uint32_t)TILE_BOARDS / 4u) {  // (with two boards per lane the first 32 lanes carry the tile)
            const uint4 r4 = reinterpret_cast<const uint4 *>(strip)[lane];
            const uint32_t d4 = reinterpret_cast<const uint32_t *>(done_bytes)[lane];
            if (a.reward) store_x4_through(a.reward + tile0 + 4u * lane, u32x4{r4.x, r4.y, r4.z, r4.w});
            store_x1_through(a.done + tile0 + 4u * lane, d4);
        }
    }
#endif
#ifdef ENV2048_EXP_STAMPS
    stamp2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0u && a.obs) {
        uint64_t *out = reinterpret_cast<uint64_t *>(a.obs) + 8 * ((uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave);
        out[0] = stamp0;
        out[1] = stamp1;
        out[2] = stamp3;
        out[3] = stamp4;
        out[4] = stamp2;
        uint32_t hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        uint32_t xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        out[5] = ((uint64_t)xcc_id << 32) | hw_id;
    }
#endif

    // 5b. the last wave of the block rebuilds the queued terminal boards
    if (!auto_reset) return;
    state_at_arrival = __builtin_amdgcn_readfirstlane(state_at_arrival);
    if ((state_at_arrival >> 16) != WAVES_PER_BLOCK - 1u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    const uint32_t n_reset = state_at_arrival & 0xFFFFu;
#if defined(ENV2048_EXP_ABLATE) && (ENV2048_EXP_ABLATE & 8)  // timing-only build: terminal boards are not rebuilt
    (void)n_reset;
#else
    reset_queued<EXTRAS>(a, block0, reset_slot, n_reset, lane, 64u, through);
#endif
}

// Scalar kernel arguments in the order the kernel needs them: the first 13 dwords (everything up to the
// Philox key) are preloaded into SGPRs by the command processor (built with
// -mllvm -amdgpu-kernarg-preload-count=16; gfx950 has 14 free user SGPRs), so a wave issues its board
// loads without first waiting for an s_load of the argument block; the output pointers arrive while the
// Philox blocks are computed.
template <bool EXTRAS, bool RANDOM, int BPL>
__global__ __launch_bounds__(BLOCK) void step_kernel(uint8_t *boards, const uint8_t *actions, uint64_t n, uint64_t env_base,
                                                     uint32_t flags, uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo,
                                                     uint32_t step_hi, int32_t *reward, uint8_t *done, uint8_t *legal, void *obs,
                                                     uint8_t *max_exp, unsigned long long *step_counter)
{
    // Step index relative to a counter in device memory (records with `step_counter`): what makes a captured hipGraph of
    // step launches replayable -- the graph's launches carry the offsets 0 .. K-1, a one-thread env2048_advance_counter
    // node at its end adds K, and every replay continues the Philox stream instead of repeating it.  One scalar load per
    // wave (wave-uniform), issued with the board loads.
    if (step_counter) {
        const unsigned long long s = __hip_atomic_load(step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                                     (((unsigned long long)step_hi << 32) | step_lo);
        step_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)s);
        step_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(s >> 32));
    }
    const StepArgs a{n, env_base, boards, actions, reward, done, legal, obs, RngKey{seed_lo, seed_hi, step_lo, step_hi}, flags, max_exp};
    __shared__ uint4 dir_lds[8];
    __shared__ __attribute__((aligned(16))) uint32_t word_lds[WAVES_PER_BLOCK][word_strip_words(BPL)];
    __shared__ uint32_t queue_state;
    __shared__ uint16_t reset_slot[block_boards(BPL)];
    const uint64_t block0 = (uint64_t)blockIdx.x * block_boards(BPL);
    if (block0 + block_boards(BPL) <= a.n)  // block-uniform: every block but the last takes the unchecked path
        step_block<EXTRAS, true, RANDOM, BPL>(a, dir_lds, word_lds, &queue_state, reset_slot);
    else
        step_block<EXTRAS, false, RANDOM, BPL>(a, dir_lds, word_lds, &queue_state, reset_slot);
}

// Several mutually INDEPENDENT step launches as ONE launch (env2048_step_many groups them on the host): blockIdx.y picks the
// record, blockIdx.x the block inside it -- the dispatcher hands out x fastest, so the records stream through the chip one
// after the other with no kernel boundary between them: the stores of record k drain while record k+1 computes and the
// loads of record k+2 are already in flight.  That is what a dependent chain of launches cannot have (every launch there is
// load-everything -> compute -> store-everything -> boundary), and it is what brings batches that do NOT live in the
// Infinity Cache from 0.45 to the memory-only floor of the access pattern.  The record table travels in the kernel
// argument block (96 B per record, read with scalar loads at a wave-uniform offset); the per-block code is step_block,
// unchanged, so results are bit-identical to the separate launches.
struct FusedRecord {
    uint8_t *boards;
    const uint8_t *actions;
    uint64_t n, env_base;
    uint32_t flags, seed_lo, seed_hi, step_lo, step_hi, blocks;
    int32_t *reward;
    uint8_t *done;
    uint8_t *legal;
    void *obs;
    uint8_t *max_exp;
};
constexpr int MAX_FUSED_RECORDS = 32;  // 32 x 96 B = 3 KiB of the 4 KiB argument block
struct FusedTable {
    FusedRecord rec[MAX_FUSED_RECORDS];
};

template <bool EXTRAS, bool RANDOM, int BPL>
__global__ __launch_bounds__(BLOCK) void step_many_kernel(const FusedTable table)
{
    const FusedRecord &r = table.rec[blockIdx.y];
    if (blockIdx.x >= r.blocks) return;  // the records of a group may differ in size: surplus blocks retire at once
    const StepArgs a{r.n, r.env_base, r.boards, r.actions, r.reward, r.done, r.legal, r.obs, RngKey{r.seed_lo, r.seed_hi, r.step_lo, r.step_hi},
                     r.flags, r.max_exp};
    __shared__ uint4 dir_lds[8];
    __shared__ __attribute__((aligned(16))) uint32_t word_lds[WAVES_PER_BLOCK][word_strip_words(BPL)];
    __shared__ uint32_t queue_state;
    __shared__ uint16_t reset_slot[block_boards(BPL)];
    const uint64_t block0 = (uint64_t)blockIdx.x * block_boards(BPL);
    if (block0 + block_boards(BPL) <= a.n)
        step_block<EXTRAS, true, RANDOM, BPL>(a, dir_lds, word_lds, &queue_state, reset_slot);
    else
        step_block<EXTRAS, false, RANDOM, BPL>(a, dir_lds, word_lds, &queue_state, reset_slot);
}

__global__ void advance_counter_kernel(unsigned long long *counter, unsigned long long delta) { *counter += delta; }
// [step-kernel-source-end]

struct ResetArgs {
    uint64_t n;
    uint64_t env_base;
    uint8_t *boards;
    uint8_t *done;
    RngKey key;
};

__global__ __launch_bounds__(BLOCK) void reset_kernel(ResetArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n) return;
    uint32_t w0, w1, w2, w3;
    env_block(a.key, a.env_base + i, STREAM_RESET, w0, w1, w2, w3);
    store_board(a.boards, i, fresh_board(w0, w1, w2, w3));
    if (a.done) a.done[i] = 0;
}

struct RolloutArgs {
    uint64_t n;
    uint64_t env_base;
    uint8_t *boards;
    int64_t *return_sum;
    uint32_t *episodes;
    uint32_t *valid_moves;
    uint32_t seed_lo, seed_hi;
    uint64_t step0;
    uint32_t T;
    uint32_t flags;
    // STATS only: running (length, return) of the game each environment is in, carried across calls (may be NULL), and
    // the 40-word accumulator block of env2048_rollout_episodes
    uint32_t *ep_length;
    int64_t *ep_return;
    unsigned long long *stats;
    // REPLAY only: pre-staged action rows (read) and per-step reward / done rows (written), all [T, row_stride]
    const uint8_t *action_rows;
    int32_t *reward_rows;
    uint8_t *done_rows;
    uint64_t row_stride;
};

// T steps with the boards resident in registers; same tile layout and RNG words as step_kernel.
// STATS: per-EPISODE statistics (what evaluate_games reports, reinforce/training/self_play.py:503-521): every lane keeps
// the running length/return of its four games; a finished game (rare: 0.7 % of steps) updates block-wide accumulators in
// LDS (count, sums, sums of squares, min/max return, histogram of the final max tile), which the block adds to the
// caller's 40-word block with one global atomic per word when it retires.
// REPLAY: the action of step t comes from row t of a pre-staged [T, row_stride] byte array instead of the step word, and
// reward / done of every step are written as rows (6 B per board-step instead of the 38 B of a step launch): T
// env2048_step calls with staged actions in ONE launch, boards in registers.  Row t+1's action bytes are requested
// before step t is computed, so their latency hides behind a whole step of arithmetic.
// BPL: boards per lane, as for step_kernel -- 4 for large batches (Philox blocks shared, VALU-bound), 1 up to 2^18
// environments (four times as many waves: a batch that small leaves most SIMDs idle otherwise).
template <bool STATS, bool REPLAY = false, int BPL = ENV2048_BOARDS_PER_LANE>
__global__ __launch_bounds__(BLOCK) void rollout_kernel(RolloutArgs a)
{
    constexpr int BOARDS_PER_LANE = BPL, TILE_BOARDS = tile_boards(BPL), WORD_STRIP = word_strip_words(BPL);  // (shadow the globals)
    __shared__ uint4 orient_lds[20];  // [4 * orientation + action] selectors, then [16 + action] rank masks
    __shared__ __attribute__((aligned(16))) uint32_t word_lds[WAVES_PER_BLOCK][WORD_STRIP];
    __shared__ unsigned long long stats_lds[STATS ? ENV2048_ROLLOUT_STATS_WORDS : 1];
    // REPLAY, wide rows: per wave 256 reward words + 256 done bytes + 256 action bytes, to turn the lane's four boards
    // (64j + l) into four ADJACENT ones (4l .. 4l+3) so that a row of the tile moves as one dwordx4 / dword per lane
    __shared__ __attribute__((aligned(16))) uint32_t rows_lds[REPLAY ? WAVES_PER_BLOCK : 1][REPLAY ? TILE_BOARDS + 2 * (TILE_BOARDS / 4) : 1];
    if (STATS) {
        if (threadIdx.x < ENV2048_ROLLOUT_STATS_WORDS) stats_lds[threadIdx.x] = threadIdx.x == 4u ? ~0ull : 0ull;  // [4] = min
    }

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 20u) orient_lds[tid] = reinterpret_cast<const uint4 *>(ORIENT_TABLE)[tid];
    __syncthreads();

    const uint64_t tile0 = ((uint64_t)blockIdx.x * WAVES_PER_BLOCK + wave) * TILE_BOARDS;
    const uint64_t env0 = a.env_base + tile0;
    const bool auto_reset = (a.flags & ENV2048_FLAG_AUTO_RESET) != 0u;
    uint32_t *strip = word_lds[wave];
    const uint32_t strip_off = (uint32_t)env0 & 3u;

    // The boards live in the PLANE SPACE of their last move for the whole kernel (transition_oriented: one byte network per
    // step instead of two); ori16[j] = 16 * orientation of bd[j] = the LDS offset of that orientation's selector rows / 4.
    Board bd[BOARDS_PER_LANE];
    uint32_t ori16[BOARDS_PER_LANE];
    bool valid[BOARDS_PER_LANE];
    int64_t ret[BOARDS_PER_LANE];
    uint32_t eps[BOARDS_PER_LANE], moves[BOARDS_PER_LANE];
    uint32_t cur_len[BOARDS_PER_LANE];
    unsigned long long cur_ret[BOARDS_PER_LANE];
#pragma unroll
    for (int j = 0; j < BOARDS_PER_LANE; ++j) {
        const uint64_t i = tile0 + 64u * j + lane;
        valid[j] = i < a.n;
        bd[j] = valid[j] ? load_board(a.boards, i) : Board{0u, 0u, 0u, 0u};
        ori16[j] = 16u * ORIENT_BOARD;
        ret[j] = 0;
        eps[j] = 0u;
        moves[j] = 0u;
        cur_len[j] = (STATS && valid[j] && a.ep_length) ? a.ep_length[i] : 0u;
        cur_ret[j] = (STATS && valid[j] && a.ep_return) ? (unsigned long long)a.ep_return[i] : 0ull;
    }
    // wide rows: the whole tile exists, four boards per lane, and the host found every row 4-/16-byte aligned
    const bool wide_rows = REPLAY && BOARDS_PER_LANE == 4 && tile0 + TILE_BOARDS <= a.n && (a.flags & INTERNAL_FLAG_WIDE_OUTPUTS) != 0u;
    uint32_t *rew_lds = rows_lds[REPLAY ? wave : 0];
    uint8_t *done_lds = reinterpret_cast<uint8_t *>(rew_lds + (REPLAY ? TILE_BOARDS : 0));
    uint8_t *act_lds = done_lds + (REPLAY ? TILE_BOARDS : 0);
    uint32_t next_act[BOARDS_PER_LANE];
    uint32_t next_act4 = 0u;  // wide: the action bytes of boards 4l .. 4l+3 of the next row
    if (REPLAY) {
        if (wide_rows) {
            next_act4 = reinterpret_cast<const uint32_t *>(a.action_rows + tile0)[lane];
        } else {
#pragma unroll
            for (int j = 0; j < BOARDS_PER_LANE; ++j) next_act[j] = (valid[j] && a.T) ? a.action_rows[tile0 + 64u * j + lane] : 0u;
        }
    }

    for (uint32_t t = 0; t < a.T; ++t) {
        uint32_t act[BOARDS_PER_LANE];
        if (REPLAY) {
            const bool more = t + 1u < a.T;  // uniform
            if (wide_rows) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // last step's reads of the action bytes are done
                __builtin_amdgcn_wave_barrier();
                reinterpret_cast<uint32_t *>(act_lds)[lane] = next_act4;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < BOARDS_PER_LANE; ++j) act[j] = act_lds[64u * j + lane];
                if (more) next_act4 = reinterpret_cast<const uint32_t *>(a.action_rows + (uint64_t)(t + 1u) * a.row_stride + tile0)[lane];
            } else {
#pragma unroll
                for (int j = 0; j < BOARDS_PER_LANE; ++j) {
                    act[j] = next_act[j];
                    if (more && valid[j]) next_act[j] = a.action_rows[(uint64_t)(t + 1u) * a.row_stride + tile0 + 64u * j + lane];
                }
            }
        }
        const uint64_t step = a.step0 + t;
        const RngKey key{a.seed_lo, a.seed_hi, (uint32_t)step, (uint32_t)(step >> 32)};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // last step's strip reads precede this rewrite
        __builtin_amdgcn_wave_barrier();
        fill_word_strip<BPL>(strip, key, env0, lane);
        uint32_t pending = 0u;  // bit j: my board j finished a game this step
#pragma unroll
        for (int j = 0; j < BOARDS_PER_LANE; ++j) {
            const uint32_t word = strip[strip_off + 64u * j + lane];
            uint32_t reward;
            bool changed, done;
            {
                const uint32_t act16 = ((REPLAY ? act[j] : word) & 3u) << 4;  // 16 * action: byte offset of its rank-mask row
                const uint4 sel = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(orient_lds) + (4u * ori16[j] + act16));
                const uint4 msk = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(orient_lds + 16) + act16);
                transition_oriented(bd[j], OrientStep{sel.x, sel.y, sel.z, sel.w, msk.x, msk.y, msk.z, msk.w}, word, reward, changed, done);
                ori16[j] = act16;
            }
            ret[j] += reward;
            eps[j] += done ? 1u : 0u;
            moves[j] += changed ? 1u : 0u;
            pending |= (done ? 1u : 0u) << j;
            if (REPLAY && wide_rows) {
                rew_lds[64u * j + lane] = reward;
                done_lds[64u * j + lane] = done ? 1 : 0;
            } else if (REPLAY && valid[j]) {
                const uint64_t at = (uint64_t)t * a.row_stride + tile0 + 64u * j + lane;
                if (a.reward_rows) __builtin_nontemporal_store((int32_t)reward, &a.reward_rows[at]);
                if (a.done_rows) __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), &a.done_rows[at]);
            }
            if (STATS && cur_len[j] != ENV2048_EPISODE_OVER) {
                cur_len[j] += 1u;
                cur_ret[j] += reward;
                if (done && valid[j]) {  // the game of this board slot ended with this move: bd[j] is its last board
                    const unsigned long long len = cur_len[j], sum = cur_ret[j];
                    atomicAdd(&stats_lds[0], 1ull);
                    atomicAdd(&stats_lds[1], len);
                    atomicAdd(&stats_lds[2], sum);
                    atomicAdd(&stats_lds[3], sum * sum);
                    atomicMin(&stats_lds[4], sum);
                    atomicMax(&stats_lds[5], sum);
                    atomicAdd(&stats_lds[6], len * len);
                    atomicAdd(&stats_lds[8u + (max_exponent(bd[j]) & 31u)], 1ull);
                    // with auto-reset the slot starts its next game; without, the finished game stays on the board and is
                    // counted ONCE: the slot is marked over (also across calls, through ep_length)
                    cur_len[j] = auto_reset ? 0u : ENV2048_EPISODE_OVER;
                    cur_ret[j] = auto_reset ? 0ull : cur_ret[j];
                }
            }
        }
        if (REPLAY && wide_rows) {  // row t of the tile: one dwordx4 of rewards and one dword of done bytes per lane
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t row0 = (uint64_t)t * a.row_stride + tile0;
            if (a.reward_rows) {
                const u32x4 r4 = reinterpret_cast<const u32x4 *>(rew_lds)[lane];
                __builtin_nontemporal_store(r4, reinterpret_cast<u32x4 *>(a.reward_rows + row0) + lane);
            }
            if (a.done_rows) __builtin_nontemporal_store(reinterpret_cast<const uint32_t *>(done_lds)[lane], reinterpret_cast<uint32_t *>(a.done_rows + row0) + lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next step rewrites these LDS rows
            __builtin_amdgcn_wave_barrier();
        }
        if (auto_reset) {
            // one reset block per finished board; usually at most one lane of the wave has one
            while (pending) {
                const uint32_t j = (uint32_t)__ffs((int)pending) - 1u;
                pending &= pending - 1u;
                uint32_t w0, w1, w2, w3;
                env_block(key, env0 + 64u * j + lane, STREAM_RESET, w0, w1, w2, w3);
                const Board fresh = fresh_board(w0, w1, w2, w3);
#pragma unroll
                for (int jj = 0; jj < BOARDS_PER_LANE; ++jj)
                    if (j == (uint32_t)jj) {
                        bd[jj] = fresh;  // rows: orientation ORIENT_BOARD
                        ori16[jj] = 16u * ORIENT_BOARD;
                    }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < BOARDS_PER_LANE; ++j) {
        if (!valid[j]) continue;
        const uint64_t i = tile0 + 64u * j + lane;
        {  // back to rows: one pass from the board's orientation to ORIENT_BOARD
            const uint4 sel = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(orient_lds) + (4u * ori16[j] + 16u * ORIENT_BOARD));
            bd[j] = orient_network(bd[j], sel.x, sel.y, sel.z, sel.w);
        }
        store_board(a.boards, i, bd[j]);
        if (a.return_sum) a.return_sum[i] = ret[j];
        if (a.episodes) a.episodes[i] = eps[j];
        if (a.valid_moves) a.valid_moves[i] = moves[j];
        if (STATS && a.ep_length) a.ep_length[i] = cur_len[j];
        if (STATS && a.ep_return) a.ep_return[i] = (int64_t)cur_ret[j];
    }
    if (STATS) {
        __syncthreads();  // every wave of the block has finished its games' updates
        if (tid < ENV2048_ROLLOUT_STATS_WORDS && tid != 7u) {
            const unsigned long long v = stats_lds[tid];
            if (tid == 4u) {
                if (v != ~0ull) atomicMin(&a.stats[4], v);
            } else if (tid == 5u) {
                atomicMax(&a.stats[5], v);
            } else if (v) {
                atomicAdd(&a.stats[tid], v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Episode segmentation of recorded step rows: what reinforce/training/self_play.py:359-392 does
// with per-game Python lists.  Rows are the [rows, row_stride] arrays the step kernel wrote (done, reward, max_exp),
// possibly a ring: logical row r lives in physical row (first_row + r) % ring_rows.  One lane per environment walks
// its column (a wave reads 64 adjacent environments of a row: coalesced).
// ------------------------------------------------------------------------------------------
struct EpisodeRows {
    uint64_t n, row_stride;
    uint32_t rows, first_row, ring_rows, flags;
    const uint8_t *done;
    const void *reward;  // int32 scores, or float32 with ENV2048_EPISODES_REWARD_F32; may be NULL
    const uint8_t *max_exp;  // may be NULL
};

constexpr uint32_t EPISODE_ROWS_IN_FLIGHT = 8;  // rows whose loads are issued together (the walk itself is sequential)

__global__ __launch_bounds__(256) void count_episodes_kernel(EpisodeRows a, uint32_t *counts)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const bool frozen = (a.flags & ENV2048_EPISODES_FROZEN) != 0u;
    uint32_t count = 0u, open = 0u, phys = a.first_row;
    bool stop = false;
    for (uint32_t r0 = 0; r0 < a.rows && !stop; r0 += EPISODE_ROWS_IN_FLIGHT) {
        uint8_t d[EPISODE_ROWS_IN_FLIGHT];
        const uint32_t batch = min(EPISODE_ROWS_IN_FLIGHT, a.rows - r0);
#pragma unroll
        for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
            d[u] = 0;
            if (u < batch) {
                d[u] = a.done[(uint64_t)phys * a.row_stride + i];
                phys = phys + 1u == a.ring_rows ? 0u : phys + 1u;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
            if (u < batch && !stop) {
                ++open;
                if (d[u]) {
                    ++count;
                    open = 0u;
                    stop = frozen;  // a finished game stops recording (self_play.py:359,368)
                }
            }
        }
    }
    counts[i] = count + (open ? 1u : 0u);
}

__global__ __launch_bounds__(256) void segment_episodes_kernel(EpisodeRows a, const uint64_t *offsets, env2048_episode *episodes,
                                                               int32_t *episode_of)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const bool frozen = (a.flags & ENV2048_EPISODES_FROZEN) != 0u;
    const bool reward_f32 = (a.flags & ENV2048_EPISODES_REWARD_F32) != 0u;
    uint64_t slot = offsets[i];
    uint32_t start = 0u, length = 0u, best = 0u, phys = a.first_row, r = 0u;
    double total = 0.0;
    bool stop = false;
    auto emit = [&](bool terminated) {
        env2048_episode e;
        e.env = (uint32_t)i;
        e.start = start;
        e.length = length;
        e.terminated = terminated ? 1 : 0;
        e.max_exp = (uint8_t)best;
        e.reserved = 0;
        e.total_reward = total;
        episodes[slot++] = e;
    };
    while (r < a.rows && !stop) {
        uint8_t d[EPISODE_ROWS_IN_FLIGHT], m[EPISODE_ROWS_IN_FLIGHT];
        double w[EPISODE_ROWS_IN_FLIGHT];
        uint64_t at[EPISODE_ROWS_IN_FLIGHT];
        const uint32_t batch = min(EPISODE_ROWS_IN_FLIGHT, a.rows - r);
#pragma unroll
        for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
            d[u] = m[u] = 0;
            w[u] = 0.0;
            at[u] = 0;
            if (u < batch) {
                at[u] = (uint64_t)phys * a.row_stride + i;
                phys = phys + 1u == a.ring_rows ? 0u : phys + 1u;
                d[u] = a.done[at[u]];
                if (a.reward) w[u] = reward_f32 ? (double)static_cast<const float *>(a.reward)[at[u]] : (double)static_cast<const int32_t *>(a.reward)[at[u]];
                if (a.max_exp) m[u] = a.max_exp[at[u]];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
            if (u < batch) {
                if (stop) {  // frozen: rows after the end of the game belong to no episode
                    if (episode_of) episode_of[at[u]] = -1;
                } else {
                    total += w[u];
                    best = max(best, (uint32_t)m[u]);
                    ++length;
                    if (episode_of) episode_of[at[u]] = (int32_t)slot;
                    if (d[u]) {
                        emit(true);
                        start = r + u + 1u;
                        length = 0u;
                        best = 0u;
                        total = 0.0;
                        stop = frozen;
                    }
                }
            }
        }
        r += batch;
    }
    if (length) emit(false);
    if (episode_of)
        for (; r < a.rows; ++r) {
            episode_of[(uint64_t)phys * a.row_stride + i] = -1;
            phys = phys + 1u == a.ring_rows ? 0u : phys + 1u;
        }
}

// reinforce/game step on int32 value boards, everything of env.py:79-133 in one launch (one board
// per lane; a compatibility path for the JAX-shaped API, not the throughput path).
__device__ __forceinline__ uint32_t pack4(int4 v)
{
    auto e = [](int x) -> uint32_t { return x > 0 ? (uint32_t)(31 - __clz(x)) : 0u; };
    return e(v.x) | (e(v.y) << 8) | (e(v.z) << 16) | (e(v.w) << 24);
}
__device__ __forceinline__ int4 unpack4(uint32_t r)
{
    auto v = [](uint32_t e) -> int { return e ? (int)(1u << (e & 31u)) : 0; };
    return make_int4(v(r & 0xFFu), v((r >> 8) & 0xFFu), v((r >> 16) & 0xFFu), v(r >> 24));
}

// `*_in` -> `*_out` (the same pointers for the in-place entry point): a finished game copies through.
__global__ __launch_bounds__(BLOCK) void step_state_i32_kernel(uint64_t n, uint64_t env_base, const int32_t *boards_in,
                                                               int32_t *boards_out, const int32_t *actions,
                                                               const int32_t *step_count_in, int32_t *step_count_out,
                                                               const uint8_t *done_in, uint8_t *done_out,
                                                               const float *total_in, float *total_out, float *reward_out,
                                                               int32_t *max_tile, RngKey key, const uint2 *keys, float *obs_out,
                                                               uint32_t *legal4_out, int64_t mirror)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    // `mirror` != 0: every output element is ALSO written `mirror` bytes further on -- the caller's pinned-host copy of its
    // result block (a few hundred bytes at the reference's batch sizes), so that reading the results back needs no copy
    auto put = [mirror](auto *p, auto v) {
        *p = v;
        if (mirror) *reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + mirror) = v;
    };
    const int4 *rows_in = reinterpret_cast<const int4 *>(boards_in) + 4 * i;
    int4 *rows_out = reinterpret_cast<int4 *>(boards_out) + 4 * i;
    const int4 v0 = rows_in[0], v1 = rows_in[1], v2 = rows_in[2], v3 = rows_in[3];
    Board bd{pack4(v0), pack4(v1), pack4(v2), pack4(v3)};
    const bool was_done = done_in[i] != 0;
    const bool in_place = boards_in == boards_out && done_in == done_out && step_count_in == step_count_out && total_in == total_out;
    // keyed form: environment i draws from ITS OWN key, as environment 0 at step 0 of the stream seeded with
    // (keys[i][0] << 32) | keys[i][1] -- row i of the batch equals the single-environment call with keys[i]
    const uint64_t env = keys ? 0u : env_base + i;
    if (keys) {
        const uint2 k = keys[i];
        key = RngKey{k.y, k.x, 0u, 0u};
    }
    uint32_t w[4];
    env_block(key, env >> 2, STREAM_STEP, w[0], w[1], w[2], w[3]);
    const uint32_t sel = (uint32_t)env & 3u;
    const uint32_t word = sel == 0u ? w[0] : sel == 1u ? w[1] : sel == 2u ? w[2] : w[3];
    float reward = 0.0f;
    int32_t steps = step_count_in ? step_count_in[i] : 0;
    float total = total_in ? total_in[i] : 0.0f;
    uint8_t done = 1;
    uint32_t legal_bits = 0u;
    if (!was_done) {
        uint32_t score;
        bool changed;
        afterstate(bd, (uint32_t)actions[i] & 3u, score, changed);
        spawn_tile_word(bd, word, changed);
        reward = (float)score;  // exact: a step's score is < 2^24
        put(rows_out + 0, unpack4(bd.r0));
        put(rows_out + 1, unpack4(bd.r1));
        put(rows_out + 2, unpack4(bd.r2));
        put(rows_out + 3, unpack4(bd.r3));
        legal_bits = legal_mask(bd);
        done = legal_bits == 0u ? 1 : 0;
        steps += 1;
        total += reward;
    } else if (legal4_out) {
        legal_bits = legal_mask(bd);  // (a frozen game: the mask of the board as it stands)
    }
    // what the caller asks about the NEW state right after a move (reinforce/training/self_play.py:316-317), in the same
    // launch: observation and (N, 4) legal bytes of the board this step leaves behind -- a finished game's unchanged board
    if (obs_out) {
        store_obs(obs_out, i, bd);
        if (mirror) store_obs(reinterpret_cast<float *>(reinterpret_cast<char *>(obs_out) + mirror), i, bd);
    }
    if (legal4_out) put(legal4_out + i, (legal_bits & 1u) | ((legal_bits & 2u) << 7) | ((legal_bits & 4u) << 14) | ((legal_bits & 8u) << 21));
    if (was_done && !in_place) {  // the values as they came in (not re-derived from the packed form)
        put(rows_out + 0, v0);
        put(rows_out + 1, v1);
        put(rows_out + 2, v2);
        put(rows_out + 3, v3);
    }
    if (!was_done || !in_place) {
        put(done_out + i, done);
        if (step_count_out) put(step_count_out + i, steps);
        if (total_out) put(total_out + i, total);
    }
    put(reward_out + i, reward);
    if (max_tile) {
        const uint32_t rws[4] = {bd.r0, bd.r1, bd.r2, bd.r3};
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m = max(m, rws[r] & 0xFFu);
            m = max(m, (rws[r] >> 8) & 0xFFu);
            m = max(m, (rws[r] >> 16) & 0xFFu);
            m = max(m, rws[r] >> 24);
        }
        put(max_tile + i, m ? (int32_t)(1u << (m & 31u)) : 0);
    }
}

// batched_reset with one PRNG key per environment, straight to int32 value boards: board i = the fresh board of
// environment 0 at step 0 of the stream seeded with keys[i] (= create_initial_board(keys[i])).
// step_count / done / total (each may be NULL): the bookkeeping fields of a fresh GameState, written as zeros.
__global__ __launch_bounds__(BLOCK) void reset_keyed_i32_kernel(uint64_t n, const uint2 *keys, int32_t *boards, float *obs_out,
                                                                uint32_t *legal4_out, int32_t *step_count, uint8_t *done, float *total,
                                                                int64_t mirror)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    auto put = [mirror](auto *p, auto v) {  // (see step_state_i32_kernel)
        *p = v;
        if (mirror) *reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + mirror) = v;
    };
    const uint2 k = keys[i];
    uint32_t w0, w1, w2, w3;
    env_block(RngKey{k.y, k.x, 0u, 0u}, 0u, STREAM_RESET, w0, w1, w2, w3);
    const Board bd = fresh_board(w0, w1, w2, w3);
    int4 *rows = reinterpret_cast<int4 *>(boards) + 4 * i;
    put(rows + 0, unpack4(bd.r0));
    put(rows + 1, unpack4(bd.r1));
    put(rows + 2, unpack4(bd.r2));
    put(rows + 3, unpack4(bd.r3));
    if (obs_out) {
        store_obs(obs_out, i, bd);
        if (mirror) store_obs(reinterpret_cast<float *>(reinterpret_cast<char *>(obs_out) + mirror), i, bd);
    }
    if (legal4_out) {
        const uint32_t bits = legal_mask(bd);
        put(legal4_out + i, (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21));
    }
    if (step_count) put(step_count + i, (int32_t)0);
    if (done) put(done + i, (uint8_t)0);
    if (total) put(total + i, 0.0f);
}

// What the reinforce/game callers ask about a batch of int32 value boards, in one launch: the f32
// observation, the legal-action mask as four 0/1 BYTES per board (a torch/NumPy bool array) and the
// JAX-flavoured done flag ("no action changes the board").
__global__ __launch_bounds__(BLOCK) void status_i32_kernel(uint64_t n, const int32_t *boards, float *obs, uint32_t *legal4,
                                                           uint8_t *done)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int4 *rows = reinterpret_cast<const int4 *>(boards) + 4 * i;
    const Board bd{pack4(rows[0]), pack4(rows[1]), pack4(rows[2]), pack4(rows[3])};
    if (obs) store_obs(obs, i, bd);
    if (legal4 || done) {
        const uint32_t bits = legal_mask(bd);
        if (legal4) legal4[i] = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
        if (done) done[i] = bits == 0u ? 1 : 0;
    }
}

__global__ __launch_bounds__(BLOCK) void afterstate_kernel(uint64_t n, const uint8_t *boards_in,
                                                           const uint8_t *actions, uint8_t *boards_out,
                                                           int32_t *reward, uint8_t *changed_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board bd = load_board(boards_in, i);
    uint32_t score;
    bool changed;
    afterstate(bd, (uint32_t)actions[i] & 3u, score, changed);
    store_board(boards_out, i, bd);
    if (reward) reward[i] = (int32_t)score;
    if (changed_out) changed_out[i] = changed ? 1 : 0;
}

// Afterstates of ALL four actions of every board (the expansion step of a search): the byte transpose is
// computed once, up/down slide the row dwords as planes, left/right the transposed ones, and the two
// directions of an axis differ only in the order the planes are handed to slide_merge_planes.
// out_boards[4*i + a], out_reward[4*i + a]; out_legal[i] bit a = action a changes the board.
__global__ __launch_bounds__(BLOCK) void afterstates4_kernel(uint64_t n, const uint8_t *boards, uint8_t *out_boards,
                                                             int32_t *out_reward, uint8_t *out_legal)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Board bd = load_board(boards, i);
    uint32_t c0, c1, c2, c3;
    transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, c0, c1, c2, c3);
    uint32_t legal = 0u;
    int4 rewards;
    {  // 0 = left: planes are the columns, wall side first
        uint32_t a = c0, b = c1, c = c2, d = c3, changed;
        rewards.x = (int32_t)slide_merge_planes(a, b, c, d, changed);
        Board out;
        transpose4x4(a, b, c, d, out.r0, out.r1, out.r2, out.r3);
        store_board(out_boards, 4 * i + 0, out);
        legal |= changed != 0u ? 1u : 0u;
    }
    {  // 1 = up: planes are the rows
        uint32_t a = bd.r0, b = bd.r1, c = bd.r2, d = bd.r3, changed;
        rewards.y = (int32_t)slide_merge_planes(a, b, c, d, changed);
        store_board(out_boards, 4 * i + 1, Board{a, b, c, d});
        legal |= changed != 0u ? 2u : 0u;
    }
    {  // 2 = right: columns, last column is the wall
        uint32_t a = c3, b = c2, c = c1, d = c0, changed;
        rewards.z = (int32_t)slide_merge_planes(a, b, c, d, changed);
        Board out;
        transpose4x4(d, c, b, a, out.r0, out.r1, out.r2, out.r3);
        store_board(out_boards, 4 * i + 2, out);
        legal |= changed != 0u ? 4u : 0u;
    }
    {  // 3 = down: rows, last row is the wall
        uint32_t a = bd.r3, b = bd.r2, c = bd.r1, d = bd.r0, changed;
        rewards.w = (int32_t)slide_merge_planes(a, b, c, d, changed);
        store_board(out_boards, 4 * i + 3, Board{d, c, b, a});
        legal |= changed != 0u ? 8u : 0u;
    }
    if (out_reward) reinterpret_cast<int4 *>(out_reward)[i] = rewards;
    if (out_legal) out_legal[i] = (uint8_t)legal;
}

// One full expectimax ply of the TRUE environment in one launch: for every board the afterstates of all four actions
// [ref: twentyfortyeight/core/gameboard.py:105-129 latent_state] and, for every legal action, all (empty cell, tile)
// chance outcomes of its afterstate with their probabilities, in the reference's enumeration order
// [ref: gameboard.py:132-171 after_state] -- 4 x 32 = 128 successor slots per board -- plus, optionally, the status byte of
// every successor (legal-action mask, terminal flag: gamemove.py:45-83, gameboard.py:328-348).  The 4n afterstates never
// go to HBM.  Work split: a wave owns 64 boards; first each lane computes the four afterstates of ONE board (one byte
// transpose shared by the four slides) and parks them, their empty-cell masks and spawn probabilities in the wave's LDS
// strip; then the wave walks its 64 boards in 128 iterations, lane l producing successor (action 2h + l/32, slot l%32) of
// board it/2 (h = it%2), so that every store instruction of the wave writes 1 KiB (boards), 256 B (probabilities) or 64 B
// (status) of CONTIGUOUS output.  HBM-bound: 16 B in, 128 x (16 + 4 [+ 1]) B out per board.
struct PlyAfter {  // one (board, action) entry of the LDS strip: 32 bytes
    uint32_t r0, r1, r2, r3;
    uint32_t empties;  // cell mask of the afterstate's empty cells; 0 = action illegal (nothing to enumerate)
    uint32_t n_empty;
    float p2, p4;  // P(tile 2) / n_empty, P(tile 4) / n_empty
};

constexpr int PLY_BLOCK = 256;

__global__ __launch_bounds__(PLY_BLOCK) void expectimax_ply_kernel(uint64_t n, const uint8_t *boards, uint8_t *succ_boards,
                                                                   float *succ_prob, uint8_t *succ_status, uint8_t *succ_count,
                                                                   int32_t *after_reward, uint8_t *legal_out)
{
    __shared__ __attribute__((aligned(16))) PlyAfter strip_lds[PLY_BLOCK / 64][64 * 4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t board0 = ((uint64_t)blockIdx.x * (PLY_BLOCK / 64) + wave) * 64u;  // first board of this wave
    if (board0 >= n) return;  // wave-uniform
    PlyAfter *strip = strip_lds[wave];

    // phase 1: one board per lane -> four afterstates into LDS
    const uint64_t mine = board0 + lane;
    if (mine < n) {
        const Board bd = load_board(boards, mine);
        uint32_t c0, c1, c2, c3;
        transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, c0, c1, c2, c3);
        Board after[4];
        int4 rewards;
        uint32_t legal = 0u, changed;
        {  // 0 = left: planes are the columns, wall side first
            uint32_t a = c0, b = c1, c = c2, d = c3;
            rewards.x = (int32_t)slide_merge_planes(a, b, c, d, changed);
            transpose4x4(a, b, c, d, after[0].r0, after[0].r1, after[0].r2, after[0].r3);
            legal |= changed != 0u ? 1u : 0u;
        }
        {  // 1 = up: planes are the rows
            uint32_t a = bd.r0, b = bd.r1, c = bd.r2, d = bd.r3;
            rewards.y = (int32_t)slide_merge_planes(a, b, c, d, changed);
            after[1] = Board{a, b, c, d};
            legal |= changed != 0u ? 2u : 0u;
        }
        {  // 2 = right: columns, last column is the wall
            uint32_t a = c3, b = c2, c = c1, d = c0;
            rewards.z = (int32_t)slide_merge_planes(a, b, c, d, changed);
            transpose4x4(d, c, b, a, after[2].r0, after[2].r1, after[2].r2, after[2].r3);
            legal |= changed != 0u ? 4u : 0u;
        }
        {  // 3 = down: rows, last row is the wall
            uint32_t a = bd.r3, b = bd.r2, c = bd.r1, d = bd.r0;
            rewards.w = (int32_t)slide_merge_planes(a, b, c, d, changed);
            after[3] = Board{d, c, b, a};
            legal |= changed != 0u ? 8u : 0u;
        }
        uint32_t counts = 0u;
#pragma unroll
        for (uint32_t act = 0; act < 4u; ++act) {
            const bool is_legal = ((legal >> act) & 1u) != 0u;
            const uint32_t empt = is_legal ? empty_mask(after[act]) : 0u;  // a legal move always leaves an empty cell
            const uint32_t n_empty = ENV2048_POPC(empt);
            const float cells = (float)(n_empty ? n_empty : 1u);
            strip[4u * lane + act] = PlyAfter{after[act].r0, after[act].r1, after[act].r2, after[act].r3, empt, n_empty, 0.9f / cells, 0.1f / cells};
            counts |= (2u * n_empty) << (8u * act);
        }
        if (after_reward) reinterpret_cast<int4 *>(after_reward)[mine] = rewards;
        if (legal_out) legal_out[mine] = (uint8_t)legal;
        if (succ_count) reinterpret_cast<uint32_t *>(succ_count)[mine] = counts;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the strip is wave-private: LDS ops of a wave are ordered
    __builtin_amdgcn_wave_barrier();

    // phase 2: 128 successor slots per board, 64 per iteration
    const uint32_t boards_here = (uint32_t)min((uint64_t)64, n - board0);
    const uint32_t slot = lane & 31u, k = slot >> 1;
    const bool four = (slot & 1u) != 0u;
    uint4 *out_b = reinterpret_cast<uint4 *>(succ_boards) + board0 * 128u + lane;
    float *out_p = succ_prob + board0 * 128u + lane;
    uint8_t *out_s = succ_status ? succ_status + board0 * 128u + lane : nullptr;
    for (uint32_t it = 0; it < 2u * boards_here; ++it) {
        const PlyAfter a = strip[4u * (it >> 1) + 2u * (it & 1u) + (lane >> 5)];
        Board bd{0u, 0u, 0u, 0u};
        float prob = 0.0f;
        uint32_t status = 0u;
        if (k < a.n_empty) {
            bd = Board{a.r0, a.r1, a.r2, a.r3};
            place_tile(bd, kth_empty(a.empties, k), four);
            prob = four ? a.p4 : a.p2;
            if (out_s) {
                const uint32_t legal = legal_mask(bd);
                status = legal | (legal == 0u ? 16u : 0u);  // a non-empty board is terminal iff no action is legal
            }
        }
        // write-once output streams: write-through (sc1) stores leave nothing dirty in the XCD L2s -- measured 121 vs 151 us
        // per 2^18 boards (ordinary stores) and 130 us (non-temporal); ENV2048_PLY_STORE=0/1 rebuilds those variants
#ifndef ENV2048_PLY_STORE
#define ENV2048_PLY_STORE 2
#endif
#if ENV2048_PLY_STORE == 1
        __builtin_nontemporal_store(bd.r0, &reinterpret_cast<uint32_t *>(out_b + 64u * it)[0]);
        __builtin_nontemporal_store(bd.r1, &reinterpret_cast<uint32_t *>(out_b + 64u * it)[1]);
        __builtin_nontemporal_store(bd.r2, &reinterpret_cast<uint32_t *>(out_b + 64u * it)[2]);
        __builtin_nontemporal_store(bd.r3, &reinterpret_cast<uint32_t *>(out_b + 64u * it)[3]);
        __builtin_nontemporal_store(prob, out_p + 64u * it);
        if (out_s) __builtin_nontemporal_store((uint8_t)status, out_s + 64u * it);
#elif ENV2048_PLY_STORE == 2
        store_x4_through(out_b + 64u * it, u32x4{bd.r0, bd.r1, bd.r2, bd.r3});
        store_x1_through(out_p + 64u * it, __float_as_uint(prob));
        if (out_s) out_s[64u * it] = (uint8_t)status;
#else
        out_b[64u * it] = make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
        out_p[64u * it] = prob;
        if (out_s) out_s[64u * it] = (uint8_t)status;
#endif
    }
}

// Depth-2 expectimax of the TRUE environment, the whole search tree of a board in one launch, nothing of it in HBM
// (VERDICT r2 "next" #8; the frontier of expectimax_ply_kernel kept on chip):
//     Q2(s, a) = r(s, a) + sum over chance outcomes o of afterstate(s, a):  P(o) * max over legal a' of r(o, a')        (legal a)
//     V2(s)    = max over legal a of Q2(s, a)   (0 for a terminal board);  best(s) = the smallest such a
// with r and the afterstates of latent_state [ref: twentyfortyeight/core/gameboard.py:105-129], the outcomes and
// P(o) = TILE_SPAWN_PROBS[value] / num_empty of after_state [ref: gameboard.py:132-171] and legality as in
// legal_actions_mask [ref: twentyfortyeight/core/gamemove.py:45-83] (== "the slide changes the board").  A successor without
// a legal move (game over) contributes 0.  Up to 4 x 16 x 2 = 128 successors per board, 4 slides each: 512 slide/merge
// evaluations per board, all in registers.
// Work split as in the ply kernel: a wave owns 64 boards; phase 1, one board per lane -> its four afterstates into a
// wave-private LDS strip; phase 2, 128 iterations in which lane l evaluates successor (action 2h + l/32, slot l%32) of
// board it/2 and the 32 lanes of a half-wave add up their best rewards separately for the tile-2 and the tile-4 outcomes
// (four xor-shuffles: the sums are INTEGERS, so the result does not depend on the order); phase 3, one board per lane
// again: Q2 = r + (0.9 * S2 + 0.1 * S4) / n_empty as an exact fraction, arg-max, outputs.  Pure VALU work: 16 B in, 89 B out per board.
struct D2After {  // one (board, action) entry of the LDS strip: 24 bytes
    uint32_t r0, r1, r2, r3;  // the afterstate
    uint32_t empties;         // cell mask of the afterstate's empty cells; 0 = action illegal
    int32_t reward;
};

// best reward over the legal moves of one board (0 if it has none): one byte transpose shared by the four slides
__device__ __forceinline__ int32_t best_move_reward(const Board &bd)
{
    // max over the four moves of the merge score.  Opposite directions score the same: after compaction a line is a sequence
    // of runs of equal tiles, and a run of L tiles of value v gives floor(L / 2) merges worth 2v each whether the greedy pairing
    // starts at its left or at its right end (gameboard.py:56-67 pairs from the wall outward; runs do not interact).  A move
    // that changes nothing merges nothing (score 0), so legality never has to be looked at: two slides instead of four.
    uint32_t c0, c1, c2, c3, changed;
    transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, c0, c1, c2, c3);
    const int32_t horizontal = (int32_t)slide_merge_planes(c0, c1, c2, c3, changed);  // left == right
    uint32_t a = bd.r0, b = bd.r1, c = bd.r2, d = bd.r3;
    const int32_t vertical = (int32_t)slide_merge_planes(a, b, c, d, changed);  // up == down
    return max(horizontal, vertical);  // all-zero = no merge anywhere: 0 either way
}

constexpr int D2_BLOCK = 256;

__global__ __launch_bounds__(D2_BLOCK) void expectimax_depth2_kernel(uint64_t n, const uint8_t *boards, int32_t *terms, float *q,
                                                                      uint8_t *best_action, float *value, double *value64)
{
    __shared__ __attribute__((aligned(16))) D2After strip_lds[D2_BLOCK / 64][64 * 4];
    __shared__ int32_t sums_lds[D2_BLOCK / 64][64 * 4][2];   // per (board, action): S2, S4
    __shared__ uint16_t first_lds[D2_BLOCK / 64][64 * 4 + 4];  // exclusive prefix sum of the successor counts (+ the total)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t board0 = ((uint64_t)blockIdx.x * (D2_BLOCK / 64) + wave) * 64u;
    if (board0 >= n) return;  // wave-uniform
    D2After *strip = strip_lds[wave];
    int32_t(*sums)[2] = sums_lds[wave];
    uint16_t *first = first_lds[wave];

    // phase 1: one board per lane -> four afterstates
    const uint64_t mine = board0 + lane;
    uint32_t mine_count = 0u, c0 = 0u, c1 = 0u, c2 = 0u;  // successors of my board, and of its first three actions
    if (mine < n) {
        const Board bd = load_board(boards, mine);
        uint32_t t0, t1, t2, t3;
        transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, t0, t1, t2, t3);
        Board after[4];
        int32_t rewards[4];
        uint32_t legal = 0u, changed;
        {
            uint32_t a = t0, b = t1, c = t2, d = t3;
            rewards[0] = (int32_t)slide_merge_planes(a, b, c, d, changed);
            transpose4x4(a, b, c, d, after[0].r0, after[0].r1, after[0].r2, after[0].r3);
            legal |= changed != 0u ? 1u : 0u;
        }
        {
            uint32_t a = bd.r0, b = bd.r1, c = bd.r2, d = bd.r3;
            rewards[1] = (int32_t)slide_merge_planes(a, b, c, d, changed);
            after[1] = Board{a, b, c, d};
            legal |= changed != 0u ? 2u : 0u;
        }
        {
            uint32_t a = t3, b = t2, c = t1, d = t0;
            rewards[2] = (int32_t)slide_merge_planes(a, b, c, d, changed);
            transpose4x4(d, c, b, a, after[2].r0, after[2].r1, after[2].r2, after[2].r3);
            legal |= changed != 0u ? 4u : 0u;
        }
        {
            uint32_t a = bd.r3, b = bd.r2, c = bd.r1, d = bd.r0;
            rewards[3] = (int32_t)slide_merge_planes(a, b, c, d, changed);
            after[3] = Board{d, c, b, a};
            legal |= changed != 0u ? 8u : 0u;
        }
        uint32_t count[4];
#pragma unroll
        for (uint32_t act = 0; act < 4u; ++act) {
            const bool is_legal = ((legal >> act) & 1u) != 0u;
            const uint32_t empt = is_legal ? empty_mask(after[act]) : 0u;  // a legal move always leaves an empty cell
            strip[4u * lane + act] = D2After{after[act].r0, after[act].r1, after[act].r2, after[act].r3, empt, is_legal ? rewards[act] : 0};
            sums[4u * lane + act][0] = 0;
            sums[4u * lane + act][1] = 0;
            count[act] = 2u * ENV2048_POPC(empt);  // successors of this afterstate: a 2 and a 4 in every empty cell
        }
        mine_count = count[0] + count[1] + count[2] + count[3];
        c0 = count[0];
        c1 = count[1];
        c2 = count[2];
    }
    // successors are WORK ITEMS numbered densely over the wave's 256 (board, action) entries: entry e owns items
    // [first[e], first[e + 1]).  On mid-game boards 34 of the 128 slots per board hold a successor, so walking the slots
    // would leave three quarters of the lanes idle.  Exclusive scan of the per-board counts over the wave (six shuffles).
    uint32_t scan = mine_count;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = __shfl_up(scan, d);
        scan += lane >= d ? up : 0u;
    }
    const uint32_t total = __shfl(scan, 63);
    const uint32_t before = scan - mine_count;
    first[4u * lane + 0u] = (uint16_t)before;
    first[4u * lane + 1u] = (uint16_t)(before + c0);
    first[4u * lane + 2u] = (uint16_t)(before + c0 + c1);
    first[4u * lane + 3u] = (uint16_t)(before + c0 + c1 + c2);
    if (lane == 63u) first[256] = (uint16_t)total;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // strip, sums and first are wave-private: LDS ops of a wave are ordered
    __builtin_amdgcn_wave_barrier();

    // phase 2: 64 successors per iteration
    for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
        const uint32_t w = w0 + lane;
        if (w < total) {
            uint32_t lo = 0u, hi = 256u;  // the LAST entry whose first item is <= w (entries without successors are skipped by that)
#pragma unroll
            for (int step = 0; step < 8; ++step) {
                const uint32_t mid = (lo + hi) >> 1;
                const bool right = first[mid] <= w;
                lo = right ? mid : lo;
                hi = right ? hi : mid;
            }
            const D2After a = strip[lo];
            const uint32_t slot = w - first[lo];
            Board bd{a.r0, a.r1, a.r2, a.r3};
            place_tile(bd, kth_empty(a.empties, slot >> 1), (slot & 1u) != 0u);
            atomicAdd(&sums[lo][slot & 1u], best_move_reward(bd));  // integer sums: the order of the adds does not matter
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // phase 3: one board per lane again
    if (mine >= n) return;
    // Q2 = r + (0.9 S2 + 0.1 S4) / n = X / D with the INTEGERS X = 10 n r + 9 S2 + S4 and D = 10 n: the arg-max compares the
    // fractions exactly (cross-multiplied in 64 bits; a tie keeps the smaller action) and the float outputs are ONE correctly
    // rounded double division each -- so device and oracle agree bit for bit, independent of fma contraction or summation order
    int64_t best_x = 0, best_d = 1;
    uint32_t best_a = 255u;
    float q4[4];
    int4 t[4];
#pragma unroll
    for (uint32_t act = 0; act < 4u; ++act) {
        const D2After a = strip[4u * lane + act];
        const int32_t s2 = sums[4u * lane + act][0], s4 = sums[4u * lane + act][1], n_empty = (int32_t)ENV2048_POPC(a.empties);
        t[act] = make_int4(a.reward, s2, s4, n_empty);
        if (a.empties != 0u) {
            const int64_t d = 10 * (int64_t)n_empty, x = d * (int64_t)a.reward + 9 * (int64_t)s2 + (int64_t)s4;
            q4[act] = (float)((double)x / (double)d);
            if (best_a == 255u || x * best_d > best_x * d) {
                best_x = x;
                best_d = d;
                best_a = act;
            }
        } else {
            q4[act] = -__builtin_inff();
        }
    }
    const double best_q = (double)best_x / (double)best_d;
    if (terms) {
        int4 *out = reinterpret_cast<int4 *>(terms) + 4 * mine;
        out[0] = t[0];
        out[1] = t[1];
        out[2] = t[2];
        out[3] = t[3];
    }
    if (q) reinterpret_cast<float4 *>(q)[mine] = make_float4(q4[0], q4[1], q4[2], q4[3]);
    if (best_action) best_action[mine] = (uint8_t)best_a;
    if (value) value[mine] = (float)best_q;
    if (value64) value64[mine] = best_q;  // (0 / 1 = 0.0 for a terminal board)
}

// ---- depth-d expectimax (d >= 3): the tree is kept on the device as one dense FRONTIER per level -------------------------
// expand: the children of a board are the (empty cell, tile) outcomes of its LEGAL afterstates, in the order (action, cell in
// row-major rank, tile 2 then 4) [ref: gameboard.py:105-171 latent_state + after_state composed]; a count pass, an exclusive
// prefix sum (host side: one torch.cumsum) and a write pass build the next level densely -- on mid-game boards 34 of the 128
// possible children exist.  The deepest frontier is valued by the depth-2 kernel; the back-up kernel folds a level's child
// values into its boards' values: Q(s, a) = r + (0.9 * S2 + 0.1 * S4) / num_empty with S2 / S4 summed in child order, every
// operation a separate correctly rounded double operation (compiled with fp contract off), V = the first maximum over the legal actions.
__device__ __forceinline__ void four_afterstates(const Board &bd, Board after[4], int32_t rewards[4], uint32_t &legal)
{
    uint32_t c0, c1, c2, c3, changed;
    transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, c0, c1, c2, c3);
    legal = 0u;
    {
        uint32_t a = c0, b = c1, c = c2, d = c3;
        rewards[0] = (int32_t)slide_merge_planes(a, b, c, d, changed);
        transpose4x4(a, b, c, d, after[0].r0, after[0].r1, after[0].r2, after[0].r3);
        legal |= changed != 0u ? 1u : 0u;
    }
    {
        uint32_t a = bd.r0, b = bd.r1, c = bd.r2, d = bd.r3;
        rewards[1] = (int32_t)slide_merge_planes(a, b, c, d, changed);
        after[1] = Board{a, b, c, d};
        legal |= changed != 0u ? 2u : 0u;
    }
    {
        uint32_t a = c3, b = c2, c = c1, d = c0;
        rewards[2] = (int32_t)slide_merge_planes(a, b, c, d, changed);
        transpose4x4(d, c, b, a, after[2].r0, after[2].r1, after[2].r2, after[2].r3);
        legal |= changed != 0u ? 4u : 0u;
    }
    {
        uint32_t a = bd.r3, b = bd.r2, c = bd.r1, d = bd.r0;
        rewards[3] = (int32_t)slide_merge_planes(a, b, c, d, changed);
        after[3] = Board{d, c, b, a};
        legal |= changed != 0u ? 8u : 0u;
    }
}

__global__ __launch_bounds__(BLOCK) void expand_count_kernel(uint64_t n, const uint8_t *boards, uint32_t *counts)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board after[4];
    int32_t rewards[4];
    uint32_t legal;
    four_afterstates(load_board(boards, i), after, rewards, legal);
    uint32_t count = 0u;
#pragma unroll
    for (uint32_t act = 0; act < 4u; ++act)
        if ((legal >> act) & 1u) count += 2u * ENV2048_POPC(empty_mask(after[act]));
    counts[i] = count;
}

__global__ __launch_bounds__(BLOCK) void expand_kernel(uint64_t n, const uint8_t *boards, const uint64_t *offsets, uint8_t *children,
                                                       int32_t *reward, uint8_t *n_empty)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board after[4];
    int32_t rewards[4];
    uint32_t legal;
    four_afterstates(load_board(boards, i), after, rewards, legal);
    uint64_t at = offsets[i];
#pragma unroll
    for (uint32_t act = 0; act < 4u; ++act) {
        const bool ok = ((legal >> act) & 1u) != 0u;
        const uint32_t empt = ok ? empty_mask(after[act]) : 0u;  // a legal move always leaves an empty cell
        const uint32_t ne = ENV2048_POPC(empt);
        reward[4 * i + act] = ok ? rewards[act] : 0;
        n_empty[4 * i + act] = (uint8_t)ne;  // 0 = illegal action
        for (uint32_t k = 0; k < ne; ++k) {
            const uint32_t bit = kth_empty(empt, k);
            Board two = after[act], four = after[act];
            place_tile(two, bit, false);
            place_tile(four, bit, true);
            store_board(children, at, two);
            store_board(children, at + 1u, four);
            at += 2u;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void expectimax_backup_kernel(uint64_t n, const int32_t *reward, const uint8_t *n_empty,
                                                                  const uint64_t *offsets, const double *child_value, double *value,
                                                                  double *q, uint8_t *best_action)
{
#pragma clang fp contract(off)  // plain operators below, created under contract(off): no fma for 0.9 * S2 + 0.1 * S4 (HIP's __dmul_rn /
    // __dadd_rn are inline functions built under the default mode -- inlined, their multiply and add DO fuse)
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t at = offsets[i];
    double best_q = 0.0;
    uint32_t best_a = 255u;
#pragma unroll
    for (uint32_t act = 0; act < 4u; ++act) {
        const uint32_t ne = n_empty[4 * i + act];
        double qa = -__builtin_inf();
        if (ne) {
            double s2 = 0.0, s4 = 0.0;
            for (uint32_t k = 0; k < ne; ++k) {
                s2 = s2 + child_value[at];
                s4 = s4 + child_value[at + 1u];
                at += 2u;
            }
            const double t2 = 0.9 * s2, t4 = 0.1 * s4;
            const double mix = t2 + t4;
            const double mean = mix / (double)ne;
            qa = (double)reward[4 * i + act] + mean;
            if (best_a == 255u || qa > best_q) {
                best_q = qa;
                best_a = act;
            }
        }
        if (q) q[4 * i + act] = qa;
    }
    value[i] = best_a == 255u ? 0.0 : best_q;
    if (best_action) best_action[i] = (uint8_t)best_a;
}

__global__ __launch_bounds__(BLOCK) void spawn_kernel(uint64_t n, uint8_t *boards, const uint8_t *spawn_k,
                                                      const uint8_t *spawn_e, uint8_t *placed)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board bd = load_board(boards, i);
    const uint32_t empt = empty_mask(bd);
    const uint32_t k = spawn_k[i];
    const bool ok = k < (uint32_t)__popc(empt);
    const uint32_t bit = ok ? kth_empty(empt, k) : 0u;
    const uint32_t e = spawn_e[i];
    // general exponent (not only 1/2): OR the byte value into the selected (empty) cell
    uint32_t p0, p1, p2, p3;
    cell_planes(bit, p0, p1, p2, p3);
    bd.r0 |= p0 * e;
    bd.r1 |= p1 * e;
    bd.r2 |= p2 * e;
    bd.r3 |= p3 * e;
    store_board(boards, i, bd);
    if (placed) placed[i] = ok ? (uint8_t)cell_index(bit) : (uint8_t)0xFF;
}

// `count` sequential random spawns per board from the environment's step words (not a hot path:
// one board per lane, every lane computes its own Philox block).
__global__ __launch_bounds__(BLOCK) void spawn_random_kernel(uint64_t n, uint64_t env_base, uint8_t *boards,
                                                             uint32_t seed_lo, uint32_t seed_hi, uint64_t step_index,
                                                             uint32_t count)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t env = env_base + i;
    Board bd = load_board(boards, i);
    for (uint32_t t = 0; t < count; ++t) {
        const uint64_t step = step_index + t;
        const RngKey key{seed_lo, seed_hi, (uint32_t)step, (uint32_t)(step >> 32)};
        uint32_t w[4];
        env_block(key, env >> 2, STREAM_STEP, w[0], w[1], w[2], w[3]);
        const uint32_t sel = (uint32_t)env & 3u;
        const uint32_t word = sel == 0u ? w[0] : sel == 1u ? w[1] : sel == 2u ? w[2] : w[3];
        const bool has_room = count_empty(bd) != 0u;
        spawn_tile_word(bd, word, has_room);
    }
    store_board(boards, i, bd);
}

// The actions ENV2048_FLAG_RANDOM_ACTIONS makes a step take, materialised: action of environment env at (seed, step) = low two
// bits of its step word.  For recorders: a step played with these actions read from memory is the same step (same words for
// the spawn), and the action row then says what was played (recorded random steps had all-zero action rows).
__global__ __launch_bounds__(BLOCK) void random_actions_kernel(uint64_t n, uint64_t env_base, uint32_t seed_lo, uint32_t seed_hi,
                                                               uint64_t step_index, uint8_t *actions)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t env = env_base + i;
    const RngKey key{seed_lo, seed_hi, (uint32_t)step_index, (uint32_t)(step_index >> 32)};
    uint32_t w[4];
    env_block(key, env >> 2, STREAM_STEP, w[0], w[1], w[2], w[3]);
    const uint32_t sel = (uint32_t)env & 3u;
    const uint32_t word = sel == 0u ? w[0] : sel == 1u ? w[1] : sel == 2u ? w[2] : w[3];
    actions[i] = (uint8_t)(word & 3u);
}

// one thread per (board, outcome slot): 32 consecutive lanes write the 512 B of one board's outcomes
__global__ __launch_bounds__(BLOCK) void chance_outcomes_kernel(uint64_t n, const uint8_t *boards, uint8_t *out_boards,
                                                                float *out_prob, uint8_t *out_count)
{
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t i = t >> 5;
    if (i >= n) return;
    const uint32_t slot = (uint32_t)t & 31u;
    Board bd = load_board(boards, i);
    const uint32_t empt = empty_mask(bd);
    const uint32_t n_empty = ENV2048_POPC(empt);
    const uint32_t k = slot >> 1;
    float prob = 0.0f;
    if (n_empty == 0u) {
        prob = slot == 0u ? 1.0f : 0.0f;  // [ref gameboard.py:156-157] full board: itself, probability 1
        if (slot != 0u) bd = Board{0u, 0u, 0u, 0u};
    } else if (k < n_empty) {
        place_tile(bd, kth_empty(empt, k), (slot & 1u) != 0u);
        prob = ((slot & 1u) ? 0.1f : 0.9f) / (float)n_empty;
    } else {
        bd = Board{0u, 0u, 0u, 0u};
    }
    store_board(out_boards, t, bd);
    out_prob[t] = prob;
    if (out_count && slot == 0u) out_count[i] = (uint8_t)(n_empty ? 2u * n_empty : 1u);
}

// One block per 64 boards; a thread writes whole output DWORDs (coalesced 4-byte stores): byte g of a
// board's 16*width-byte record belongs to cell g / width and is 1 iff g % width equals that cell's
// exponent.  All index arithmetic is 32-bit on block-local offsets (16 * width is a multiple of 4).
__global__ __launch_bounds__(BLOCK) void onehot_kernel(uint64_t n, const uint8_t *boards, uint32_t width, uint32_t *out)
{
    __shared__ uint8_t cells[64 * 16];
    const uint64_t board0 = (uint64_t)blockIdx.x * 64;
    const uint32_t n_boards = (uint32_t)min((uint64_t)64, n - board0);
    for (uint32_t c = threadIdx.x; c < n_boards * 16u; c += BLOCK) cells[c] = boards[board0 * 16 + c];
    __syncthreads();
    const uint32_t dwords = n_boards * 4u * width;  // 16 * width bytes per board
    uint32_t *dst = out + board0 * 4u * width;
    for (uint32_t d = threadIdx.x; d < dwords; d += BLOCK) {
        const uint32_t g0 = 4u * d;
        uint32_t cell = g0 / width;
        uint32_t pos = g0 - cell * width;
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            word |= (pos == cells[cell & 1023u] ? 1u : 0u) << (8 * b);
            if (++pos == width) {
                pos = 0;
                ++cell;
            }
        }
        dst[d] = word;
    }
}

// Same output, 16 bytes per thread (1 KiB per wave store) for 16-byte aligned `out`: instead of testing every
// output byte, a thread walks the (at most 16/width + 2) cells that overlap its chunk and drops each cell's
// single 1 into place.  One division per chunk, done as a multiply (exact for block-local offsets < 2^18).
__global__ __launch_bounds__(BLOCK) void onehot_kernel16(uint64_t n, const uint8_t *boards, uint32_t width, uint32_t inv_width,
                                                         uint4 *out)
{
    __shared__ uint8_t cells[64 * 16];
    const uint64_t board0 = (uint64_t)blockIdx.x * 64;
    const uint32_t n_boards = (uint32_t)min((uint64_t)64, n - board0);
    for (uint32_t c = threadIdx.x; c < n_boards * 16u; c += BLOCK) cells[c] = boards[board0 * 16 + c];
    __syncthreads();
    const uint32_t chunks = n_boards * width;  // 16 * width bytes per board = width chunks of 16 bytes
    uint4 *dst = out + board0 * width;
    for (uint32_t q = threadIdx.x; q < chunks; q += BLOCK) {
        const uint32_t g0 = 16u * q;
        uint32_t cell = __umulhi(g0, inv_width);  // g0 / width
        uint32_t base = cell * width;             // first output byte of that cell
        uint64_t lo = 0, hi = 0;
        while (base < g0 + 16u) {
            const uint32_t e = cells[cell & 1023u];
            const uint32_t off = base + e - g0;  // position of the cell's 1 inside this chunk (wraps if before it)
            if (e < width && off < 16u) {
                if (off < 8u)
                    lo |= 1ull << (8u * off);
                else
                    hi |= 1ull << (8u * (off - 8u));
            }
            ++cell;
            base += width;
        }
        dst[q] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    }
}

__global__ __launch_bounds__(256) void status_kernel(uint64_t n, const uint8_t *boards, uint8_t *done,
                                                     uint8_t *legal, float *obs, uint8_t *max_exp,
                                                     uint8_t *n_empty)
{
    __shared__ __attribute__((aligned(16))) uint32_t strip_lds[4][256];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool valid = i < n;
    const Board bd = valid ? load_board(boards, i) : Board{0u, 0u, 0u, 0u};
    if (obs) store_obs_coalesced(obs, strip_lds[wave], i - lane, n, lane, bd, !valid);  // whole wave takes part
    if (!valid) return;
    const uint32_t empties = count_empty(bd);
    if (done) done[i] = (empties == 0u && no_equal_neighbours(bd)) ? 1 : 0;
    if (legal) legal[i] = (uint8_t)legal_mask(bd);
    if (n_empty) n_empty[i] = (uint8_t)empties;
    if (max_exp) {
        const uint32_t rows[4] = {bd.r0, bd.r1, bd.r2, bd.r3};
        uint32_t m = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m = max(m, rows[r] & 0xFFu);
            m = max(m, (rows[r] >> 8) & 0xFFu);
            m = max(m, (rows[r] >> 16) & 0xFFu);
            m = max(m, rows[r] >> 24);
        }
        max_exp[i] = (uint8_t)m;
    }
}

// values <-> exponents; one thread per 4 cells (16-byte value loads for int32, coalesced).
template <typename T>
__global__ __launch_bounds__(BLOCK) void pack_kernel(uint64_t n_quads, const T *values, uint32_t *packed)
{
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n_quads) return;
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t v = (uint64_t)values[4 * q + j];
        const uint32_t e = v ? (uint32_t)(63 - __clzll((long long)v)) : 0u;
        out |= (e & 0xFFu) << (8 * j);
    }
    packed[q] = out;
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void unpack_kernel(uint64_t n_quads, const uint32_t *packed, T *values)
{
    const uint64_t q = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n_quads) return;
    const uint32_t x = packed[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t e = (x >> (8 * j)) & 0xFFu;
        values[4 * q + j] = e ? (T)((uint64_t)1 << (e & 63u)) : (T)0;
    }
}

// ------------------------------------------------------------------------------------------
// Host side of the C-ABI
// ------------------------------------------------------------------------------------------
thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(ENV2048_ERR_HIP_BASE - (int)e, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
}

bool misaligned(const void *p, uintptr_t a) { return p && (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; }

// Select `device` for the duration of one entry point and put the caller's device back after.  The device
// count is read once per process (it cannot change under a running HIP runtime); per call only the calling
// thread's current device is queried, and nothing is switched when it already is the requested one.
int visible_devices(hipError_t *err)
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
};

int check_range(uint64_t n, uint64_t env_base)
{
    // env_hi is folded into counter word 3 as 2*env_hi + stream: keep env below 2^63.
    if (env_base + n < env_base || (env_base + n) >> 63)
        return fail(ENV2048_ERR_RANGE, "env_base + n = %llu + %llu exceeds the 2^63 environment index range",
                    (unsigned long long)env_base, (unsigned long long)n);
    return ENV2048_OK;
}

int check_count(uint64_t n)
{
    if (n > ((uint64_t)0x7FFFFFFF) * BLOCK)
        return fail(ENV2048_ERR_RANGE, "n = %llu exceeds the largest single launch (%llu boards)",
                    (unsigned long long)n, (unsigned long long)0x7FFFFFFF * BLOCK);
    return ENV2048_OK;
}

inline dim3 grid_for(uint64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }
inline dim3 rollout_grid(uint64_t n, int bpl) { return dim3((unsigned)((n + block_boards(bpl) - 1) / block_boards(bpl))); }
constexpr uint64_t ROLLOUT_SMALL_MAX_ENVS = 1ull << 18;  // up to here one board per lane (more waves) beats shared Philox blocks
                                                          // (512 steps: 0.44-0.87 ms vs 1.29-1.41 ms; at 2^20: 3.04 vs 2.64 ms)
inline bool rollout_small(uint64_t n, uint32_t flags)
{
    return (flags & ENV2048_FLAG_TILE_SMALL) ? true : (flags & ENV2048_FLAG_TILE_LARGE) ? false : n <= ROLLOUT_SMALL_MAX_ENVS;
}

int launched(const char *what)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ENV2048_OK : hip_fail(e, what);
}

constexpr uint32_t STEP_FLAGS = ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS | ENV2048_FLAG_OBS_F16 | ENV2048_FLAG_OBS_BF16 |
                                ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_STORES_WRITE_BACK | ENV2048_FLAG_REWARD_LOG2_F32 |
                                ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE | ENV2048_FLAG_SINGLE_CHAIN;
// Launches up to this many boards use write-through stores unless a STORES_* flag says otherwise: they win while
// boards + outputs of one launch stay within the 256 MiB Infinity Cache (measured 2^20..2^23 boards: -5..-13 %)
// and lose beyond it (2^24: +7 %).
constexpr uint64_t WRITE_THROUGH_MAX_BOARDS = 1ull << 23;

int validate_step(const env2048_step_args &a, const char *who)
{
    if (a.flags & ~STEP_FLAGS) return fail(ENV2048_ERR_INVALID_ARG, "%s: unknown flag bits 0x%x", who, a.flags);
    if ((a.flags & ENV2048_FLAG_OBS_F16) && (a.flags & ENV2048_FLAG_OBS_BF16))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: OBS_F16 and OBS_BF16 exclude each other", who);
    if ((a.flags & ENV2048_FLAG_STORES_WRITE_THROUGH) && (a.flags & ENV2048_FLAG_STORES_WRITE_BACK))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: STORES_WRITE_THROUGH and STORES_WRITE_BACK exclude each other", who);
    if ((a.flags & ENV2048_FLAG_TILE_SMALL) && (a.flags & ENV2048_FLAG_TILE_LARGE))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: TILE_SMALL and TILE_LARGE exclude each other", who);
    if (a.reserved) return fail(ENV2048_ERR_INVALID_ARG, "%s: reserved must be 0", who);
    if ((a.flags & ENV2048_FLAG_REWARD_LOG2_F32) && !a.reward)
        return fail(ENV2048_ERR_INVALID_ARG, "%s: REWARD_LOG2_F32 without a reward array", who);
    if (a.n == 0) return ENV2048_OK;
    if (!a.boards || !a.done) return fail(ENV2048_ERR_INVALID_ARG, "%s: boards and done are required", who);
    if (!a.actions && !(a.flags & ENV2048_FLAG_RANDOM_ACTIONS))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: actions is NULL without ENV2048_FLAG_RANDOM_ACTIONS", who);
    if (misaligned(a.boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "%s: boards must be 16-byte aligned", who);
    if (misaligned(a.reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "%s: reward must be 4-byte aligned", who);
    if (misaligned(a.obs, 16)) return fail(ENV2048_ERR_ALIGNMENT, "%s: obs must be 16-byte aligned", who);
    if (misaligned(a.step_counter, 8)) return fail(ENV2048_ERR_ALIGNMENT, "%s: step_counter must be 8-byte aligned", who);
    if (int rc = check_range(a.n, a.env_base)) return rc;
    return check_count(a.n);
}

// [step-kernel-source-begin]
// Enqueue one step launch (arguments already validated, device already selected).
void launch_step(const env2048_step_args &a, hipStream_t stream)
{
    uint32_t kernel_flags = a.flags & ~(ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_STORES_WRITE_BACK | ENV2048_FLAG_TILE_SMALL |
                                        ENV2048_FLAG_TILE_LARGE | ENV2048_FLAG_SINGLE_CHAIN);
    if (!misaligned(a.reward, 16) && !misaligned(a.done, 4)) kernel_flags |= INTERNAL_FLAG_WIDE_OUTPUTS;
    const bool through = (a.flags & ENV2048_FLAG_STORES_WRITE_THROUGH)  ? true
                         : (a.flags & ENV2048_FLAG_STORES_WRITE_BACK) ? false
                                                                      : a.n <= WRITE_THROUGH_MAX_BOARDS;
    if (through) kernel_flags |= INTERNAL_FLAG_WRITE_THROUGH;
#ifdef ENV2048_EXP_STAMPS
    const bool extras = false;
#else
    const bool extras = a.legal || a.obs || a.max_exp || (a.flags & ENV2048_FLAG_REWARD_LOG2_F32);
#endif
    const bool random_actions = (a.flags & ENV2048_FLAG_RANDOM_ACTIONS) != 0u;
    const bool small = (a.flags & ENV2048_FLAG_TILE_SMALL)   ? true
                       : (a.flags & ENV2048_FLAG_TILE_LARGE) ? false
                                                             : a.n <= SMALL_TILING_MAX_BOARDS;
    const dim3 grid((unsigned)((a.n + block_boards(small ? 1 : BOARDS_PER_LANE) - 1) / block_boards(small ? 1 : BOARDS_PER_LANE))), block(BLOCK);
    const uint32_t seed_lo = (uint32_t)a.seed, seed_hi = (uint32_t)(a.seed >> 32);
    const uint32_t step_lo = (uint32_t)a.step_index, step_hi = (uint32_t)(a.step_index >> 32);
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(a.step_counter);
#define ENV2048_LAUNCH_STEP(EXTRAS, RANDOM)                                                                                        \
    do {                                                                                                                           \
        if (small)                                                                                                                 \
            hipLaunchKernelGGL((step_kernel<EXTRAS, RANDOM, 1>), grid, block, 0, stream, a.boards, a.actions, a.n, a.env_base,     \
                               kernel_flags, seed_lo, seed_hi, step_lo, step_hi, a.reward, a.done, a.legal, a.obs, a.max_exp,      \
                               counter);                                                                                           \
        else                                                                                                                       \
            hipLaunchKernelGGL((step_kernel<EXTRAS, RANDOM, BOARDS_PER_LANE>), grid, block, 0, stream, a.boards, a.actions, a.n,   \
                               a.env_base, kernel_flags, seed_lo, seed_hi, step_lo, step_hi, a.reward, a.done, a.legal, a.obs,     \
                               a.max_exp, counter);                                                                                \
    } while (0)
    if (extras && random_actions)
        ENV2048_LAUNCH_STEP(true, true);
    else if (extras)
        ENV2048_LAUNCH_STEP(true, false);
    else if (random_actions)
        ENV2048_LAUNCH_STEP(false, true);
    else
        ENV2048_LAUNCH_STEP(false, false);
#undef ENV2048_LAUNCH_STEP
}

// ---- env2048_step_many: records that do not depend on each other become ONE launch ---------------------------------------
// The call's contract is "the records, in order".  Two records depend on each other iff one WRITES memory the other reads or
// writes (boards are read and written; actions are read; reward / done / legal / obs / max_exp are written); records that
// touch disjoint memory give the same bytes in any order, so a run of consecutive, pairwise independent records may execute
// concurrently -- as one step_many_kernel launch.  The policy is a function of the argument records alone (addresses, sizes,
// flags): no caller hint, no environment variable, no timing.
struct MemRange {
    uintptr_t lo, hi;
    uint32_t record;
    bool write;
};

inline int record_ranges(const env2048_step_args &a, uint32_t record, MemRange *out)
{
    int c = 0;
    auto add = [&](const void *p, uint64_t bytes, bool write) {
        if (p && bytes) out[c++] = MemRange{reinterpret_cast<uintptr_t>(p), reinterpret_cast<uintptr_t>(p) + bytes, record, write};
    };
    add(a.boards, 16 * a.n, true);
    if (!(a.flags & ENV2048_FLAG_RANDOM_ACTIONS)) add(a.actions, a.n, false);
    add(a.reward, 4 * a.n, true);
    add(a.done, a.n, true);
    add(a.legal, a.n, true);
    add(a.obs, a.n * 16u * (obs_format(a.flags) == OBS_F32 ? 4u : 2u), true);
    add(a.max_exp, a.n, true);
    return c;
}

// A set of byte ranges kept as sorted, pairwise disjoint intervals (overlapping or touching insertions are merged), so that
// "does [lo, hi) meet the set" is one binary search.  Fixed capacity: a group has at most MAX_FUSED_RECORDS x 7 ranges.
struct IntervalSet {
    static constexpr int CAP = MAX_FUSED_RECORDS * 7;
    uintptr_t lo[CAP], hi[CAP];
    int n = 0;
    int first_ending_after(uintptr_t x) const  // smallest i with hi[i] > x (hi is sorted because the intervals are disjoint)
    {
        int a = 0, b = n;
        while (a < b) {
            const int m = (a + b) / 2;
            if (hi[m] > x)
                b = m;
            else
                a = m + 1;
        }
        return a;
    }
    bool meets(uintptr_t l, uintptr_t h) const
    {
        const int i = first_ending_after(l);
        return i < n && lo[i] < h;
    }
    void insert(uintptr_t l, uintptr_t h)
    {
        int i = first_ending_after(l);
        int j = i;  // intervals [i, j) overlap [l, h): swallow them
        while (j < n && lo[j] < h) {
            l = lo[j] < l ? lo[j] : l;
            h = hi[j] > h ? hi[j] : h;
            ++j;
        }
        if (j == i) {  // nothing swallowed: open a slot
            for (int k = n; k > i; --k) {
                lo[k] = lo[k - 1];
                hi[k] = hi[k - 1];
            }
            ++n;
        } else if (j > i + 1) {  // several swallowed: close the gap
            const int drop = j - i - 1;
            for (int k = i + 1; k + drop < n; ++k) {
                lo[k] = lo[k + drop];
                hi[k] = hi[k + drop];
            }
            n -= drop;
        }
        lo[i] = l;
        hi[i] = h;
    }
};

struct StepVariant {
    bool extras, random_actions;
    bool operator==(const StepVariant &o) const { return extras == o.extras && random_actions == o.random_actions; }
};

inline StepVariant variant_of(const env2048_step_args &a)
{
    return StepVariant{a.legal || a.obs || a.max_exp || (a.flags & ENV2048_FLAG_REWARD_LOG2_F32) != 0u, (a.flags & ENV2048_FLAG_RANDOM_ACTIONS) != 0u};
}

// Tiling and store policy of a launch over `total` boards (one record, or all records of a fused group together).
inline bool plan_small_tiling(uint64_t total, uint32_t flags)
{
    return (flags & ENV2048_FLAG_TILE_SMALL) ? true : (flags & ENV2048_FLAG_TILE_LARGE) ? false : total <= SMALL_TILING_MAX_BOARDS;
}
inline bool plan_write_through(uint64_t total, uint32_t flags)
{
    return (flags & ENV2048_FLAG_STORES_WRITE_THROUGH) ? true : (flags & ENV2048_FLAG_STORES_WRITE_BACK) ? false : total <= WRITE_THROUGH_MAX_BOARDS;
}

// group[k] = index of the launch record k belongs to (consecutive records with the same index form one fused launch; a group
// of one is an ordinary env2048_step launch); returns the number of launches.  Greedy and order-preserving: a record joins the
// running group iff it is independent of every member, asks for the same kernel variant and tiling, does not read its step
// index from device memory, and the group stays within the argument-block table and within 2x of its smallest member's
// block count (surplus blocks of a ragged 2-D grid are cheap, not free).
uint64_t plan_groups(uint64_t count, const env2048_step_args *args, uint32_t *group)
{
    uint64_t launches = 0, k = 0;
    IntervalSet written, read;  // what the running group's members write / only read
    MemRange ranges[7];
    auto join = [&](const env2048_step_args &a, bool check) {
        const int c = record_ranges(a, 0, ranges);
        if (check)
            for (int i = 0; i < c; ++i)
                if (written.meets(ranges[i].lo, ranges[i].hi) || (ranges[i].write && read.meets(ranges[i].lo, ranges[i].hi))) return false;
        for (int i = 0; i < c; ++i) (ranges[i].write ? written : read).insert(ranges[i].lo, ranges[i].hi);
        return true;
    };
    while (k < count) {
        if (args[k].n == 0) {  // enqueues nothing; rides along with whatever comes next
            group[k++] = (uint32_t)launches;
            continue;
        }
        const StepVariant variant = variant_of(args[k]);
        const uint32_t tile_hint = args[k].flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE);
        uint64_t lo_n = args[k].n, hi_n = args[k].n;
        int members = 0;
        if (!args[k].step_counter) {
            written.n = read.n = 0;
            join(args[k], false);
            members = 1;
        }
        group[k++] = (uint32_t)launches;
        while (members && members < MAX_FUSED_RECORDS && k < count) {
            const env2048_step_args &a = args[k];
            if (a.n == 0 || a.step_counter || !(variant_of(a) == variant) ||
                (a.flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE)) != tile_hint)
                break;
            const uint64_t new_lo = a.n < lo_n ? a.n : lo_n, new_hi = a.n > hi_n ? a.n : hi_n;
            if (new_hi > 2 * new_lo && new_hi > (uint64_t)block_boards(1)) break;  // ragged beyond 2x (sizes within one block never are)
            if (!join(a, true)) break;  // it reads or writes what a member writes, or writes what a member reads: a new launch
            lo_n = new_lo;
            hi_n = new_hi;
            ++members;
            group[k++] = (uint32_t)launches;
        }
        ++launches;
    }
    return launches;
}

// The launches of an env2048_step_many call, in order: fn(first, last, live, total) for every group of records [first, last)
// that goes out together (live = records with n > 0, total = their boards).  Planning runs over windows of
// 2 x MAX_FUSED_RECORDS records so that its state stays on the stack; a group cut by the end of a window is planned again
// with the next window, so a long run of independent records is only ever cut where a full table ends.
template <typename Fn>
void for_each_launch(uint64_t count, const env2048_step_args *args, Fn fn)
{
    constexpr uint64_t WINDOW = 2 * MAX_FUSED_RECORDS;
    uint32_t group[WINDOW];
    for (uint64_t base = 0; base < count;) {
        const uint64_t window = count - base < WINDOW ? count - base : WINDOW;
        plan_groups(window, args + base, group);
        uint64_t take = window;
        if (base + window < count) {  // (a full window holds at least two groups: a group has at most MAX_FUSED_RECORDS members)
            while (take > 0 && group[take - 1] == group[window - 1]) --take;
        }
        for (uint64_t k = 0; k < take;) {
            uint64_t e = k + 1, live = 0, total = 0;
            while (e < take && group[e] == group[k]) ++e;
            for (uint64_t r = k; r < e; ++r) {
                live += args[base + r].n ? 1u : 0u;
                total += args[base + r].n;
            }
            fn(base + k, base + e, live, total);
            k = e;
        }
        base += take;
    }
}

// Enqueue records [first, last) -- pairwise independent, same variant -- as one step_many_kernel launch.
void launch_group(const env2048_step_args *args, uint64_t first, uint64_t last, hipStream_t stream)
{
    uint64_t total = 0;
    for (uint64_t k = first; k < last; ++k) total += args[k].n;
    while (first < last && args[first].n == 0) ++first;  // (an empty record may lead the group: it decides nothing)
    if (first == last) return;
    const bool small = plan_small_tiling(total, args[first].flags);
    const int bpl = small ? 1 : BOARDS_PER_LANE;
    FusedTable table;
    uint32_t max_blocks = 0, records = 0;
    for (uint64_t k = first; k < last; ++k) {
        const env2048_step_args &a = args[k];
        if (a.n == 0) continue;
        uint32_t kernel_flags = a.flags & ~(ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_STORES_WRITE_BACK | ENV2048_FLAG_TILE_SMALL |
                                            ENV2048_FLAG_TILE_LARGE | ENV2048_FLAG_SINGLE_CHAIN);
        if (!misaligned(a.reward, 16) && !misaligned(a.done, 4)) kernel_flags |= INTERNAL_FLAG_WIDE_OUTPUTS;
        if (plan_write_through(total, a.flags)) kernel_flags |= INTERNAL_FLAG_WRITE_THROUGH;
        const uint32_t blocks = (uint32_t)((a.n + block_boards(bpl) - 1) / block_boards(bpl));
        max_blocks = blocks > max_blocks ? blocks : max_blocks;
        table.rec[records++] = FusedRecord{a.boards, a.actions, a.n, a.env_base, kernel_flags, (uint32_t)a.seed, (uint32_t)(a.seed >> 32),
                                           (uint32_t)a.step_index, (uint32_t)(a.step_index >> 32), blocks, a.reward, a.done, a.legal, a.obs,
                                           a.max_exp};
    }
    const StepVariant v = variant_of(args[first]);
    const dim3 grid(max_blocks, records), block(BLOCK);
#define ENV2048_LAUNCH_MANY(EXTRAS, RANDOM)                                                                         \
    do {                                                                                                            \
        if (small)                                                                                                  \
            hipLaunchKernelGGL((step_many_kernel<EXTRAS, RANDOM, 1>), grid, block, 0, stream, table);               \
        else                                                                                                        \
            hipLaunchKernelGGL((step_many_kernel<EXTRAS, RANDOM, BOARDS_PER_LANE>), grid, block, 0, stream, table); \
    } while (0)
    if (v.extras && v.random_actions)
        ENV2048_LAUNCH_MANY(true, true);
    else if (v.extras)
        ENV2048_LAUNCH_MANY(true, false);
    else if (v.random_actions)
        ENV2048_LAUNCH_MANY(false, true);
    else
        ENV2048_LAUNCH_MANY(false, false);
#undef ENV2048_LAUNCH_MANY
}
// [step-kernel-source-end]

// ---- env2048_step_sequence: two half-batch chains -----------------------------------------------------------------------------
// K dependent steps of ONE batch are K dependent launches, and a 2^20-board launch is a single generation of workgroups: launch
// -> load phase -> arithmetic -> store drain -> kernel boundary, none of which overlaps with the next launch of the same chain
// (20-25 % above the memory floor, whatever the instruction count or occupancy).  But the environments of a batch do
// not depend on each other: the two HALVES of the batch are two independent chains, and run on two streams they slide against
// each other -- one half's load phase and store drain under the other half's arithmetic (measured: 6.5-6.7 us per 2^20-board
// step instead of 7.7-7.9).  One host thread cannot feed two chains (3.5 us per launch against a 6.5 us step), so the second
// chain is enqueued by a helper thread per device while the caller's thread enqueues the first; both chains are ordered behind
// everything already on the caller's stream (event fork) and the caller's stream waits for both (event join), so to the caller
// the call still looks like K launches on `stream`.  Results are those of the unsplit launches bit for bit (global environment
// indices in the Philox counters).
constexpr uint64_t SPLIT_MIN_BOARDS = 1ull << 20;  // below: a half (< 2^19 boards) is latency- / launch-bound on its own (2^19: 6.1 vs 5.5 us)
constexpr uint64_t SPLIT_MAX_BOARDS = 1ull << 23;  // above: one launch already runs at the memory floor (2^23: 43.7 vs 48.4 us, 2^24: 118 vs 98)
constexpr uint32_t SPLIT_MIN_STEPS = 32;           // two enqueuing threads cost 6-7 us of host time per step while the queues fill: K = 20
                                                   // blocks are enqueue-bound (+3 % .. -3 %), K = 50: -10 %, 100: -12 %, >= 1000: -15..17 %

struct DeviceLanes {  // per device, created on first use, never destroyed (a helper blocked in wait() must outlive static destructors)
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;  // re-recorded by every call (calls are serialised by call_mutex)
    std::mutex call_mutex;  // one split call at a time per device
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, job_done = false;
    std::atomic<bool> posted{false};  // set with has_job: what the helper polls while it is still warm
    hipError_t job_error = hipSuccess;
    hipError_t create_error = hipSuccess;
    pid_t owner = 0;  // the process that created the helper thread (threads do not survive a fork)
};

DeviceLanes *lanes_for(int device)
{
    static std::mutex table_mutex;
    static DeviceLanes *table[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(table_mutex);
    if (table[device] && table[device]->owner != getpid()) table[device] = nullptr;  // forked child: its helper does not exist
    if (!table[device]) {
        DeviceLanes *L = new DeviceLanes;
        L->owner = getpid();
        L->create_error = hipStreamCreateWithFlags(&L->side, hipStreamNonBlocking);  // (the caller has selected `device`)
        if (L->create_error == hipSuccess) L->create_error = hipEventCreateWithFlags(&L->fork, hipEventDisableTiming);
        if (L->create_error == hipSuccess) L->create_error = hipEventCreateWithFlags(&L->join, hipEventDisableTiming);
        if (L->create_error == hipSuccess) {
            std::thread([L, device] {
                (void)hipSetDevice(device);
                std::unique_lock<std::mutex> lk(L->m);
                for (;;) {
                    if (!L->has_job) {
                        // calls come in bursts (a stepping loop): after a job keep polling for ~300 us before going to sleep, so
                        // that the next call's second chain starts within a microsecond instead of a futex wake-up later
                        lk.unlock();
                        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
                        while (!L->posted.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
                        lk.lock();
                    }
                    L->cv.wait(lk, [L] { return L->has_job; });
                    L->posted.store(false, std::memory_order_relaxed);
                    std::function<void()> job = std::move(L->job);
                    L->has_job = false;
                    lk.unlock();
                    job();
                    const hipError_t e = hipGetLastError();
                    lk.lock();
                    L->job_error = e;
                    L->job_done = true;
                    L->cv.notify_all();
                }
            }).detach();
        }
        table[device] = L;
    }
    return table[device];
}

inline bool plan_two_chains(uint64_t n, uint32_t count, uint32_t flags)
{
#ifdef ENV2048_EXP_SPLIT_MIN_STEPS
    static const uint32_t min_steps = getenv("ENV2048_SPLIT_MIN_STEPS") ? (uint32_t)atoi(getenv("ENV2048_SPLIT_MIN_STEPS")) : SPLIT_MIN_STEPS;
    static const uint64_t max_boards = getenv("ENV2048_SPLIT_MAX_LOG2") ? 1ull << atoi(getenv("ENV2048_SPLIT_MAX_LOG2")) : SPLIT_MAX_BOARDS;
    return !(flags & ENV2048_FLAG_SINGLE_CHAIN) && n >= SPLIT_MIN_BOARDS && n <= max_boards && count >= min_steps;
#endif
    return !(flags & ENV2048_FLAG_SINGLE_CHAIN) && n >= SPLIT_MIN_BOARDS && n <= SPLIT_MAX_BOARDS && count >= SPLIT_MIN_STEPS;
}

}  // namespace

extern "C" {

int env2048_abi_version(void) { return ENV2048_ABI_VERSION; }

const char *env2048_last_error(void) { return g_error; }

int env2048_device_count(void)
{
    hipError_t e = hipSuccess;
    const int count = visible_devices(&e);
    if (e != hipSuccess) return fail(ENV2048_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return count;
}

int env2048_reset(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint8_t *done,
                  uint64_t seed, uint64_t step_index)
{
    if (n == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_reset: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_reset: boards must be 16-byte aligned");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    ResetArgs a{n, env_base, boards, done,
                RngKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)}};
    hipLaunchKernelGGL(reset_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_reset launch");
}

int env2048_step(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards,
                 const uint8_t *actions, int32_t *reward, uint8_t *done, uint8_t *legal, void *obs,
                 uint64_t seed, uint64_t step_index, uint32_t flags)
{
    const env2048_step_args a{n, env_base, boards, actions, reward, done, legal, obs, seed, step_index, flags, 0u, nullptr, nullptr};
    if (int rc = validate_step(a, "env2048_step")) return rc;
    if (n == 0) return ENV2048_OK;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    launch_step(a, (hipStream_t)stream);
    return launched("env2048_step launch");
}

int env2048_advance_counter(int device, void *stream, uint64_t *counter, uint64_t delta)
{
    if (!counter || misaligned(counter, 8)) return fail(ENV2048_ERR_INVALID_ARG, "env2048_advance_counter: counter is NULL or misaligned");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long *>(counter),
                       (unsigned long long)delta);
    return launched("env2048_advance_counter launch");
}

int env2048_step_many(int device, void *stream, uint64_t count, const env2048_step_args *args)
{
    if (count == 0) return ENV2048_OK;
    if (!args) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_many: args is NULL");
    for (uint64_t k = 0; k < count; ++k)  // all or nothing: a bad record enqueues no launch at all
        if (int rc = validate_step(args[k], "env2048_step_many")) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    for_each_launch(count, args, [&](uint64_t first, uint64_t last, uint64_t live, uint64_t) {
        if (live == 1) {
            for (uint64_t r = first; r < last; ++r)
                if (args[r].n) launch_step(args[r], (hipStream_t)stream);
        } else if (live > 1) {
            launch_group(args, first, last, (hipStream_t)stream);
        }
    });
    return launched("env2048_step_many launch");
}

int env2048_step_plan(uint64_t n, uint32_t flags, int32_t *boards_per_lane, int32_t *write_through)
{
    if (flags & ~STEP_FLAGS) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_plan: unknown flag bits 0x%x", flags);
    if (boards_per_lane) *boards_per_lane = plan_small_tiling(n, flags) ? 1 : BOARDS_PER_LANE;
    if (write_through) *write_through = plan_write_through(n, flags) ? 1 : 0;
    return ENV2048_OK;
}

int env2048_step_many_plan(uint64_t count, const env2048_step_args *args, uint32_t *launch_of, int32_t *boards_per_lane,
                           int32_t *write_through)
{
    if (count == 0) return 0;
    if (!args || !launch_of) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_many_plan: args and launch_of are required");
    for (uint64_t k = 0; k < count; ++k)
        if (int rc = validate_step(args[k], "env2048_step_many_plan")) return rc;
    uint32_t launches = 0;
    for_each_launch(count, args, [&](uint64_t first, uint64_t last, uint64_t live, uint64_t total) {
        for (uint64_t r = first; r < last; ++r) {
            launch_of[r] = launches;
            if (boards_per_lane) boards_per_lane[r] = plan_small_tiling(total, args[last - 1].flags) ? 1 : BOARDS_PER_LANE;
            if (write_through) write_through[r] = plan_write_through(total, args[r].flags) ? 1 : 0;
        }
        if (live) ++launches;
    });
    return (int)launches;
}

int env2048_step_sequence(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards,
                          const uint8_t *const *actions, int32_t *reward, uint8_t *done, uint8_t *legal, void *obs,
                          uint64_t row_stride, uint64_t seed, uint64_t step_index, uint32_t count, uint32_t flags)
{
    if (count == 0) return ENV2048_OK;
    if (!actions && !(flags & ENV2048_FLAG_RANDOM_ACTIONS))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_sequence: actions is NULL without ENV2048_FLAG_RANDOM_ACTIONS");
    if (step_index + count < step_index) return fail(ENV2048_ERR_RANGE, "env2048_step_sequence: step_index + count overflows");
    if (row_stride && row_stride < n)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_sequence: row_stride %llu is smaller than n = %llu (rows would overlap)",
                    (unsigned long long)row_stride, (unsigned long long)n);
    const uint32_t fmt = obs_format(flags);
    const uint64_t obs_bytes = fmt == OBS_F32 ? 4u : 2u;
    auto record = [&](uint32_t t) {
        const uint64_t off = (uint64_t)t * row_stride;  // in boards
        return env2048_step_args{n,
                                 env_base,
                                 boards,
                                 (flags & ENV2048_FLAG_RANDOM_ACTIONS) || !actions ? nullptr : actions[t],
                                 reward ? reward + off : nullptr,
                                 done ? done + off : nullptr,
                                 legal ? legal + off : nullptr,
                                 obs ? static_cast<uint8_t *>(obs) + 16u * obs_bytes * off : nullptr,
                                 seed,
                                 step_index + t,
                                 flags,
                                 0u,
                                 nullptr,
                                 nullptr};
    };
    for (uint32_t t = 0; t < count; ++t)
        if (int rc = validate_step(record(t), "env2048_step_sequence")) return rc;
    if (n == 0) return ENV2048_OK;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipStream_t main_stream = (hipStream_t)stream;
    bool two_chains = plan_two_chains(n, count, flags);
    if (two_chains) {  // a stream that is being captured into a graph takes the plain form (no cross-thread work inside a capture)
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(main_stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            two_chains = false;
        }
    }
    DeviceLanes *L = two_chains ? lanes_for(device) : nullptr;
    if (!L || L->create_error != hipSuccess) {
        for (uint32_t t = 0; t < count; ++t) launch_step(record(t), main_stream);
        return launched("env2048_step_sequence launch");
    }
    // the second half starts on a block boundary of the large tiling, so the first half consists of full blocks only
    const uint64_t h = (n / 2) / block_boards(BOARDS_PER_LANE) * block_boards(BOARDS_PER_LANE);
    auto half = [&](uint32_t t, int which) {
        env2048_step_args a = record(t);
        if (which == 0) {
            a.n = h;
        } else {
            a.n = n - h;
            a.env_base += h;
            a.boards += 16 * h;
            if (a.actions) a.actions += h;
            if (a.reward) a.reward += h;
            if (a.done) a.done += h;
            if (a.legal) a.legal += h;
            if (a.obs) a.obs = static_cast<uint8_t *>(a.obs) + 16u * obs_bytes * h;
        }
        return a;
    };
    std::lock_guard<std::mutex> call(L->call_mutex);
    hipError_t e = hipEventRecord(L->fork, main_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(L->side, L->fork, 0);
    if (e != hipSuccess) return hip_fail(e, "env2048_step_sequence: fork of the second chain");
    {
        std::lock_guard<std::mutex> lk(L->m);
        L->job = [&] {
            for (uint32_t t = 0; t < count; ++t) launch_step(half(t, 1), L->side);
        };
        L->has_job = true;
        L->job_done = false;
        L->posted.store(true, std::memory_order_release);
    }
    L->cv.notify_all();
    for (uint32_t t = 0; t < count; ++t) launch_step(half(t, 0), main_stream);
    const hipError_t mine = hipGetLastError();
    hipError_t theirs;
    {
        std::unique_lock<std::mutex> lk(L->m);
        L->cv.wait(lk, [L] { return L->job_done; });
        theirs = L->job_error;
    }
    e = hipEventRecord(L->join, L->side);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, L->join, 0);
    if (mine != hipSuccess) return hip_fail(mine, "env2048_step_sequence launch");
    if (theirs != hipSuccess) return hip_fail(theirs, "env2048_step_sequence launch (second chain)");
    if (e != hipSuccess) return hip_fail(e, "env2048_step_sequence: join of the second chain");
    return ENV2048_OK;
}

int env2048_step_sequence_plan(uint64_t n, uint32_t count, uint32_t flags, int32_t *chains, int32_t *boards_per_lane, int32_t *write_through)
{
    if (flags & ~STEP_FLAGS) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_sequence_plan: unknown flag bits 0x%x", flags);
    const bool two = plan_two_chains(n, count, flags);
    const uint64_t per_launch = two ? n - (n / 2) / block_boards(BOARDS_PER_LANE) * block_boards(BOARDS_PER_LANE) : n;
    if (chains) *chains = two ? 2 : 1;
    if (boards_per_lane) *boards_per_lane = plan_small_tiling(per_launch, flags) ? 1 : BOARDS_PER_LANE;
    if (write_through) *write_through = plan_write_through(per_launch, flags) ? 1 : 0;
    return ENV2048_OK;
}

static int step_state_i32_common(const char *who, int device, void *stream, uint64_t n, uint64_t env_base, const uint32_t *keys,
                                 const int32_t *boards_in, int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                 int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in,
                                 float *total_reward_out, float *reward, int32_t *max_tile, uint64_t seed, uint64_t step_index,
                                 float *obs_out = nullptr, uint8_t *legal4_out = nullptr, int64_t host_mirror_delta = 0)
{
    if (n == 0) return ENV2048_OK;
    if (misaligned(obs_out, 16) || misaligned(legal4_out, 4)) return fail(ENV2048_ERR_ALIGNMENT, "%s: obs must be 16-byte, legal4 4-byte aligned", who);
    if (host_mirror_delta & 15) return fail(ENV2048_ERR_ALIGNMENT, "%s: host_mirror_delta must be a multiple of 16 bytes", who);
    if (host_mirror_delta && boards_in == boards_out)
        return fail(ENV2048_ERR_INVALID_ARG, "%s: a host mirror goes with the out-of-place form (outputs in one block of their own)", who);
    if (!boards_in || !boards_out || !actions || !done_in || !done_out || !reward)
        return fail(ENV2048_ERR_INVALID_ARG, "%s: boards, actions, done and reward are required", who);
    if ((step_count_in == nullptr) != (step_count_out == nullptr) || (total_reward_in == nullptr) != (total_reward_out == nullptr))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: step_count / total_reward need both their in and out array", who);
    if (misaligned(boards_in, 16) || misaligned(boards_out, 16)) return fail(ENV2048_ERR_ALIGNMENT, "%s: boards must be 16-byte aligned", who);
    if (misaligned(actions, 4) || misaligned(step_count_in, 4) || misaligned(step_count_out, 4) || misaligned(total_reward_in, 4) ||
        misaligned(total_reward_out, 4) || misaligned(reward, 4) || misaligned(max_tile, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "%s: misaligned 4-byte array", who);
    if (misaligned(keys, 8)) return fail(ENV2048_ERR_ALIGNMENT, "%s: keys must be 8-byte aligned", who);
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)};
    hipLaunchKernelGGL(step_state_i32_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, env_base, boards_in, boards_out,
                       actions, step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward,
                       max_tile, key, reinterpret_cast<const uint2 *>(keys), obs_out, reinterpret_cast<uint32_t *>(legal4_out),
                       host_mirror_delta);
    return launched(who);
}

int env2048_step_state_i32_to(int device, void *stream, uint64_t n, uint64_t env_base, const int32_t *boards_in,
                              int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in, int32_t *step_count_out,
                              const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in, float *total_reward_out,
                              float *reward, int32_t *max_tile, uint64_t seed, uint64_t step_index)
{
    return step_state_i32_common("env2048_step_state_i32", device, stream, n, env_base, nullptr, boards_in, boards_out, actions,
                                 step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward, max_tile,
                                 seed, step_index);
}

int env2048_step_state_i32_keyed(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                 int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in, int32_t *step_count_out,
                                 const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in, float *total_reward_out,
                                 float *reward, int32_t *max_tile)
{
    if (n && !keys) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_state_i32_keyed: keys is NULL");
    return step_state_i32_common("env2048_step_state_i32_keyed", device, stream, n, 0, keys, boards_in, boards_out, actions,
                                 step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward, max_tile,
                                 0, 0);
}

int env2048_reset_state_i32_keyed(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards, int32_t *step_count,
                                  uint8_t *done, float *total_reward, float *obs, uint8_t *legal4, int64_t host_mirror_delta)
{
    if (n == 0) return ENV2048_OK;
    if (!keys || !boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_reset_keyed_i32: keys and boards are required");
    if (misaligned(keys, 8) || misaligned(boards, 16) || misaligned(obs, 16) || misaligned(legal4, 4) || misaligned(step_count, 4) ||
        misaligned(total_reward, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_reset_keyed_i32: keys must be 8-byte, boards / obs 16-byte, the other arrays 4-byte aligned");
    if (host_mirror_delta & 15) return fail(ENV2048_ERR_ALIGNMENT, "env2048_reset_state_i32_keyed: host_mirror_delta must be a multiple of 16 bytes");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(reset_keyed_i32_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, reinterpret_cast<const uint2 *>(keys),
                       boards, obs, reinterpret_cast<uint32_t *>(legal4), step_count, done, total_reward, host_mirror_delta);
    return launched("env2048_reset_keyed_i32 launch");
}

int env2048_reset_keyed_i32_status(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards, float *obs,
                                   uint8_t *legal4)
{
    return env2048_reset_state_i32_keyed(device, stream, n, keys, boards, nullptr, nullptr, nullptr, obs, legal4, 0);
}

int env2048_reset_keyed_i32(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards)
{
    return env2048_reset_keyed_i32_status(device, stream, n, keys, boards, nullptr, nullptr);
}

int env2048_step_state_i32_keyed_status(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                        int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                        int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in,
                                        float *total_reward_out, float *reward, int32_t *max_tile, float *obs, uint8_t *legal4)
{
    if (n && !keys) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_state_i32_keyed_status: keys is NULL");
    return step_state_i32_common("env2048_step_state_i32_keyed_status", device, stream, n, 0, keys, boards_in, boards_out, actions,
                                 step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward, max_tile,
                                 0, 0, obs, legal4);
}

int env2048_step_state_i32_keyed_mirrored(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                          int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                          int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in,
                                          float *total_reward_out, float *reward, int32_t *max_tile, float *obs, uint8_t *legal4,
                                          int64_t host_mirror_delta)
{
    if (n && !keys) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_state_i32_keyed_mirrored: keys is NULL");
    return step_state_i32_common("env2048_step_state_i32_keyed_mirrored", device, stream, n, 0, keys, boards_in, boards_out, actions,
                                 step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward, max_tile,
                                 0, 0, obs, legal4, host_mirror_delta);
}

int env2048_step_state_i32(int device, void *stream, uint64_t n, uint64_t env_base, int32_t *boards, const int32_t *actions,
                           int32_t *step_count, uint8_t *done_io, float *total_reward, float *reward, int32_t *max_tile,
                           uint64_t seed, uint64_t step_index)
{
    return env2048_step_state_i32_to(device, stream, n, env_base, boards, boards, actions, step_count, step_count, done_io, done_io,
                                     total_reward, total_reward, reward, max_tile, seed, step_index);
}

int env2048_status_i32(int device, void *stream, uint64_t n, const int32_t *boards, float *obs, uint8_t *legal4, uint8_t *done)
{
    if (n == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_status_i32: boards is NULL");
    if (misaligned(boards, 16) || misaligned(obs, 16) || misaligned(legal4, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_status_i32: boards/obs must be 16-byte, legal4 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(status_i32_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, obs,
                       reinterpret_cast<uint32_t *>(legal4), done);
    return launched("env2048_status_i32 launch");
}

static int check_episode_rows(const char *who, uint64_t n, uint64_t row_stride, uint32_t rows, uint32_t first_row, uint32_t ring_rows,
                              const uint8_t *done, const void *reward, uint32_t flags)
{
    if (flags & ~(ENV2048_EPISODES_FROZEN | ENV2048_EPISODES_REWARD_F32)) return fail(ENV2048_ERR_INVALID_ARG, "%s: unknown flag bits 0x%x", who, flags);
    if (!done) return fail(ENV2048_ERR_INVALID_ARG, "%s: done rows are required", who);
    if (row_stride < n) return fail(ENV2048_ERR_INVALID_ARG, "%s: row_stride %llu is smaller than n = %llu", who, (unsigned long long)row_stride, (unsigned long long)n);
    if (rows > ring_rows || (ring_rows && first_row >= ring_rows))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: rows %u / first_row %u do not fit a ring of %u rows", who, rows, first_row, ring_rows);
    if (misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "%s: reward must be 4-byte aligned", who);
    return check_count(n);
}

int env2048_count_episodes(int device, void *stream, uint64_t n, uint64_t row_stride, uint32_t rows, uint32_t first_row,
                           uint32_t ring_rows, const uint8_t *done, uint32_t flags, uint32_t *counts)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_episode_rows("env2048_count_episodes", n, row_stride, rows, first_row, ring_rows, done, nullptr, flags)) return rc;
    if (!counts || misaligned(counts, 4)) return fail(ENV2048_ERR_INVALID_ARG, "env2048_count_episodes: counts is NULL or misaligned");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const EpisodeRows a{n, row_stride, rows, first_row, ring_rows, flags, done, nullptr, nullptr};
    hipLaunchKernelGGL(count_episodes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, counts);
    return launched("env2048_count_episodes launch");
}

int env2048_segment_episodes(int device, void *stream, uint64_t n, uint64_t row_stride, uint32_t rows, uint32_t first_row,
                             uint32_t ring_rows, const uint8_t *done, const void *reward, const uint8_t *max_exp, uint32_t flags,
                             const uint64_t *offsets, env2048_episode *episodes, int32_t *episode_of)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_episode_rows("env2048_segment_episodes", n, row_stride, rows, first_row, ring_rows, done, reward, flags)) return rc;
    if (!offsets || !episodes) return fail(ENV2048_ERR_INVALID_ARG, "env2048_segment_episodes: offsets and episodes are required");
    if (misaligned(offsets, 8) || misaligned(episodes, 8) || misaligned(episode_of, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_segment_episodes: misaligned offsets / episodes / episode_of");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const EpisodeRows a{n, row_stride, rows, first_row, ring_rows, flags, done, reward, max_exp};
    hipLaunchKernelGGL(segment_episodes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, offsets,
                       episodes, episode_of);
    return launched("env2048_segment_episodes launch");
}

int env2048_rollout(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                    uint64_t step0, uint32_t T, int64_t *return_sum, uint32_t *episodes, uint32_t *valid_moves,
                    uint32_t flags)
{
    if (n == 0) return ENV2048_OK;
    if (flags & ~(ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS | ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_rollout: unknown flag bits 0x%x", flags);
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_rollout: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_rollout: boards must be 16-byte aligned");
    if (misaligned(return_sum, 8) || misaligned(episodes, 4) || misaligned(valid_moves, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_rollout: misaligned output array");
    if (step0 + T < step0) return fail(ENV2048_ERR_RANGE, "env2048_rollout: step0 + T overflows");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    RolloutArgs a{n, env_base, boards, return_sum, episodes, valid_moves, (uint32_t)seed, (uint32_t)(seed >> 32),
                  step0, T, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (rollout_small(n, flags))
        hipLaunchKernelGGL((rollout_kernel<false, false, 1>), rollout_grid(n, 1), dim3(BLOCK), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((rollout_kernel<false, false, BOARDS_PER_LANE>), rollout_grid(n, BOARDS_PER_LANE), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_rollout launch");
}

int env2048_replay(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions, int32_t *reward,
                   uint8_t *done, uint64_t row_stride, uint64_t seed, uint64_t step0, uint32_t T, uint32_t flags,
                   int64_t *return_sum, uint32_t *episodes, uint32_t *valid_moves)
{
    if (n == 0 || T == 0) return ENV2048_OK;
    if (flags & ~(ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_replay: unknown flag bits 0x%x", flags);
    const uint32_t tile_hint = flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE);
    flags &= ENV2048_FLAG_AUTO_RESET;
    if (!boards || !actions) return fail(ENV2048_ERR_INVALID_ARG, "env2048_replay: boards and actions are required");
    if (row_stride < n)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_replay: row_stride %llu is smaller than n = %llu", (unsigned long long)row_stride,
                    (unsigned long long)n);
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_replay: boards must be 16-byte aligned");
    if (misaligned(reward, 4) || misaligned(return_sum, 8) || misaligned(episodes, 4) || misaligned(valid_moves, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_replay: misaligned output array");
    if (step0 + T < step0) return fail(ENV2048_ERR_RANGE, "env2048_replay: step0 + T overflows");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    if ((row_stride & 3u) == 0u && !misaligned(actions, 4) && !misaligned(reward, 16) && !misaligned(done, 4))
        flags |= INTERNAL_FLAG_WIDE_OUTPUTS;  // every row of a full tile can move as dwordx4 / dword per lane
    RolloutArgs a{n, env_base, boards, return_sum, episodes, valid_moves, (uint32_t)seed, (uint32_t)(seed >> 32), step0, T, flags,
                  nullptr, nullptr, nullptr, actions, reward, done, row_stride};
    if (rollout_small(n, tile_hint))
        hipLaunchKernelGGL((rollout_kernel<false, true, 1>), rollout_grid(n, 1), dim3(BLOCK), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((rollout_kernel<false, true, BOARDS_PER_LANE>), rollout_grid(n, BOARDS_PER_LANE), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_replay launch");
}

int env2048_rollout_episodes(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed, uint64_t step0,
                             uint32_t T, uint32_t flags, uint32_t *ep_length, int64_t *ep_return, uint64_t *stats)
{
    if (n == 0) return ENV2048_OK;
    if (flags & ~(ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS | ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE))
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_rollout_episodes: unknown flag bits 0x%x", flags);
    if (!boards || !stats) return fail(ENV2048_ERR_INVALID_ARG, "env2048_rollout_episodes: boards and stats are required");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_rollout_episodes: boards must be 16-byte aligned");
    if (misaligned(stats, 8) || misaligned(ep_return, 8) || misaligned(ep_length, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_rollout_episodes: misaligned stats / ep_return / ep_length");
    if (step0 + T < step0) return fail(ENV2048_ERR_RANGE, "env2048_rollout_episodes: step0 + T overflows");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    RolloutArgs a{n, env_base, boards, nullptr, nullptr, nullptr, (uint32_t)seed, (uint32_t)(seed >> 32), step0, T, flags,
                  ep_length, ep_return, reinterpret_cast<unsigned long long *>(stats), nullptr, nullptr, nullptr, 0};
    if (rollout_small(n, flags))
        hipLaunchKernelGGL((rollout_kernel<true, false, 1>), rollout_grid(n, 1), dim3(BLOCK), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((rollout_kernel<true, false, BOARDS_PER_LANE>), rollout_grid(n, BOARDS_PER_LANE), dim3(BLOCK), 0, (hipStream_t)stream, a);
    return launched("env2048_rollout_episodes launch");
}

int env2048_afterstate(int device, void *stream, uint64_t n, const uint8_t *boards_in, const uint8_t *actions,
                       uint8_t *boards_out, int32_t *reward, uint8_t *changed)
{
    if (n == 0) return ENV2048_OK;
    if (!boards_in || !actions || !boards_out)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_afterstate: boards_in, actions and boards_out are required");
    if (misaligned(boards_in, 16) || misaligned(boards_out, 16))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstate: boards must be 16-byte aligned");
    if (misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstate: reward must be 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(afterstate_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards_in, actions,
                       boards_out, reward, changed);
    return launched("env2048_afterstate launch");
}

int env2048_afterstates4(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *out_boards, int32_t *out_reward,
                         uint8_t *out_legal)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !out_boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_afterstates4: boards and out_boards are required");
    if (misaligned(boards, 16) || misaligned(out_boards, 16) || misaligned(out_reward, 16))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstates4: boards, out_boards and out_reward must be 16-byte aligned");
    if (int rc = check_count(4 * n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(afterstates4_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, out_boards, out_reward,
                       out_legal);
    return launched("env2048_afterstates4 launch");
}

int env2048_spawn(int device, void *stream, uint64_t n, uint8_t *boards, const uint8_t *spawn_k,
                  const uint8_t *spawn_e, uint8_t *placed)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !spawn_k || !spawn_e)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_spawn: boards, spawn_k and spawn_e are required");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_spawn: boards must be 16-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(spawn_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, spawn_k, spawn_e,
                       placed);
    return launched("env2048_spawn launch");
}

int env2048_spawn_random(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                         uint64_t step_index, uint32_t count)
{
    if (n == 0 || count == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_spawn_random: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_spawn_random: boards must be 16-byte aligned");
    if (count > 16) return fail(ENV2048_ERR_INVALID_ARG, "env2048_spawn_random: count %u exceeds the 16 cells of a board", count);
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(spawn_random_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, env_base, boards,
                       (uint32_t)seed, (uint32_t)(seed >> 32), step_index, count);
    return launched("env2048_spawn_random launch");
}

int env2048_random_actions(int device, void *stream, uint64_t n, uint64_t env_base, uint64_t seed, uint64_t step_index, uint8_t *actions)
{
    if (n == 0) return ENV2048_OK;
    if (!actions) return fail(ENV2048_ERR_INVALID_ARG, "env2048_random_actions: actions is NULL");
    if (int rc = check_range(n, env_base)) return rc;
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(random_actions_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, env_base, (uint32_t)seed,
                       (uint32_t)(seed >> 32), step_index, actions);
    return launched("env2048_random_actions launch");
}

int env2048_chance_outcomes(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *out_boards, float *out_prob,
                            uint8_t *out_count)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !out_boards || !out_prob)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_chance_outcomes: boards, out_boards and out_prob are required");
    if (misaligned(boards, 16) || misaligned(out_boards, 16) || misaligned(out_prob, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_chance_outcomes: misaligned buffer");
    if (int rc = check_count(32 * n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(chance_outcomes_kernel, grid_for(32 * n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, out_boards,
                       out_prob, out_count);
    return launched("env2048_chance_outcomes launch");
}

int env2048_expectimax_ply(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *succ_boards, float *succ_prob,
                           uint8_t *succ_status, uint8_t *succ_count, int32_t *after_reward, uint8_t *legal)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !succ_boards || !succ_prob)
        return fail(ENV2048_ERR_INVALID_ARG, "env2048_expectimax_ply: boards, succ_boards and succ_prob are required");
    if (misaligned(boards, 16) || misaligned(succ_boards, 16) || misaligned(succ_prob, 4) || misaligned(after_reward, 16) ||
        misaligned(succ_count, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_expectimax_ply: boards / succ_boards / after_reward must be 16-byte, succ_prob / succ_count 4-byte aligned");
    if (int rc = check_count(128 * n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(expectimax_ply_kernel, dim3((unsigned)((n + PLY_BLOCK - 1) / PLY_BLOCK)), dim3(PLY_BLOCK), 0, (hipStream_t)stream, n,
                       boards, succ_boards, succ_prob, succ_status, succ_count, after_reward, legal);
    return launched("env2048_expectimax_ply launch");
}

int env2048_expectimax_depth2(int device, void *stream, uint64_t n, const uint8_t *boards, int32_t *terms, float *q, uint8_t *best_action,
                              float *value)
{
    if (n == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expectimax_depth2: boards is NULL");
    if (!terms && !q && !best_action && !value) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expectimax_depth2: no output requested");
    if (misaligned(boards, 16) || misaligned(terms, 16) || misaligned(q, 16) || misaligned(value, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_expectimax_depth2: boards / terms / q must be 16-byte, value 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(expectimax_depth2_kernel, dim3((unsigned)((n + D2_BLOCK - 1) / D2_BLOCK)), dim3(D2_BLOCK), 0, (hipStream_t)stream, n,
                       boards, terms, q, best_action, value, static_cast<double *>(nullptr));
    return launched("env2048_expectimax_depth2 launch");
}

int env2048_expectimax_depth2_f64(int device, void *stream, uint64_t n, const uint8_t *boards, double *value)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !value) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expectimax_depth2_f64: NULL argument");
    if (misaligned(boards, 16) || misaligned(value, 8)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_expectimax_depth2_f64: boards must be 16-byte, value 8-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(expectimax_depth2_kernel, dim3((unsigned)((n + D2_BLOCK - 1) / D2_BLOCK)), dim3(D2_BLOCK), 0, (hipStream_t)stream, n,
                       boards, static_cast<int32_t *>(nullptr), static_cast<float *>(nullptr), static_cast<uint8_t *>(nullptr),
                       static_cast<float *>(nullptr), value);
    return launched("env2048_expectimax_depth2_f64 launch");
}

int env2048_expand_count(int device, void *stream, uint64_t n, const uint8_t *boards, uint32_t *counts)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !counts) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expand_count: NULL argument");
    if (misaligned(boards, 16) || misaligned(counts, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_expand_count: boards must be 16-byte, counts 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(expand_count_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, counts);
    return launched("env2048_expand_count launch");
}

int env2048_expand(int device, void *stream, uint64_t n, const uint8_t *boards, const uint64_t *offsets, uint8_t *children, int32_t *reward,
                   uint8_t *n_empty)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !offsets || !reward || !n_empty) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expand: boards, offsets, reward and n_empty are required");
    if (misaligned(boards, 16) || misaligned(children, 16) || misaligned(offsets, 8) || misaligned(reward, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_expand: boards / children must be 16-byte, offsets 8-byte, reward 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(expand_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, offsets, children, reward, n_empty);
    return launched("env2048_expand launch");
}

int env2048_expectimax_backup(int device, void *stream, uint64_t n, const int32_t *reward, const uint8_t *n_empty, const uint64_t *offsets,
                              const double *child_value, double *value, double *q, uint8_t *best_action)
{
    if (n == 0) return ENV2048_OK;
    if (!reward || !n_empty || !offsets || !value) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expectimax_backup: reward, n_empty, offsets and value are required");
    if (misaligned(reward, 4) || misaligned(offsets, 8) || misaligned(child_value, 8) || misaligned(value, 8) || misaligned(q, 8))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_expectimax_backup: misaligned array");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(expectimax_backup_kernel, grid_for(n), dim3(BLOCK), 0, (hipStream_t)stream, n, reward, n_empty, offsets, child_value,
                       value, q, best_action);
    return launched("env2048_expectimax_backup launch");
}

int env2048_encode_onehot(int device, void *stream, uint64_t n, const uint8_t *boards, uint32_t width, uint8_t *out)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_encode_onehot: NULL argument");
    if (width == 0 || width > 255) return fail(ENV2048_ERR_INVALID_ARG, "env2048_encode_onehot: width %u not in [1, 255]", width);
    if (misaligned(out, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_encode_onehot: out must be 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    if (!misaligned(out, 16))
        hipLaunchKernelGGL(onehot_kernel16, dim3((unsigned)((n + 63) / 64)), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, width,
                           0xFFFFFFFFu / width + 1u, reinterpret_cast<uint4 *>(out));
    else
        hipLaunchKernelGGL(onehot_kernel, dim3((unsigned)((n + 63) / 64)), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, width,
                           reinterpret_cast<uint32_t *>(out));
    return launched("env2048_encode_onehot launch");
}

int env2048_status(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *done, uint8_t *legal,
                   float *obs, uint8_t *max_exp, uint8_t *n_empty)
{
    if (n == 0) return ENV2048_OK;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_status: boards is NULL");
    if (misaligned(boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_status: boards must be 16-byte aligned");
    if (misaligned(obs, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_status: obs must be 16-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(status_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, boards, done, legal, obs,
                       max_exp, n_empty);
    return launched("env2048_status launch");
}

#define ENV2048_DEFINE_PACK(SUFFIX, TYPE)                                                                          \
    int env2048_pack_##SUFFIX(int device, void *stream, uint64_t n, const TYPE *values, uint8_t *boards)           \
    {                                                                                                              \
        if (n == 0) return ENV2048_OK;                                                                             \
        if (!values || !boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_pack_" #SUFFIX ": NULL argument");   \
        if (misaligned(boards, 16) || misaligned(values, sizeof(TYPE)))                                            \
            return fail(ENV2048_ERR_ALIGNMENT, "env2048_pack_" #SUFFIX ": misaligned buffer");                     \
        if (int rc = check_count(4 * n)) return rc;                                                                \
        DeviceScope scope(device);                                                                                 \
        if (scope.rc) return scope.rc;                                                                             \
        hipLaunchKernelGGL(pack_kernel<TYPE>, grid_for(4 * n), dim3(BLOCK), 0, (hipStream_t)stream, 4 * n, values, \
                           reinterpret_cast<uint32_t *>(boards));                                                  \
        return launched("env2048_pack_" #SUFFIX " launch");                                                        \
    }                                                                                                              \
    int env2048_unpack_##SUFFIX(int device, void *stream, uint64_t n, const uint8_t *boards, TYPE *values)         \
    {                                                                                                              \
        if (n == 0) return ENV2048_OK;                                                                             \
        if (!values || !boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_unpack_" #SUFFIX ": NULL argument"); \
        if (misaligned(boards, 16) || misaligned(values, sizeof(TYPE)))                                            \
            return fail(ENV2048_ERR_ALIGNMENT, "env2048_unpack_" #SUFFIX ": misaligned buffer");                   \
        if (int rc = check_count(4 * n)) return rc;                                                                \
        DeviceScope scope(device);                                                                                 \
        if (scope.rc) return scope.rc;                                                                             \
        hipLaunchKernelGGL(unpack_kernel<TYPE>, grid_for(4 * n), dim3(BLOCK), 0, (hipStream_t)stream, 4 * n,       \
                           reinterpret_cast<const uint32_t *>(boards), values);                                    \
        return launched("env2048_unpack_" #SUFFIX " launch");                                                      \
    }

ENV2048_DEFINE_PACK(i32, int32_t)
ENV2048_DEFINE_PACK(i64, int64_t)

}  // extern "C"
