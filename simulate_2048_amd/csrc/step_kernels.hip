// step_kernels.hip -- the hot path: step_kernel / step_many_kernel / reset_kernel (gfx950) and their C-ABI entry points
// (env2048_reset, env2048_step, env2048_step_many, env2048_step_sequence, the *_plan queries) of include/env2048.h.
//
// All board arithmetic happens in VGPRs (board_ops.hpp); LDS only carries the direction-selector table, the redistributed
// Philox words and the reset queue.  No MFMA -- this path is integer/indexing work bounded by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "board_io.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"
#include "philox_strip.hpp"
#include "step_planner.hpp"
#include "helper_thread.hpp"
#include "direct_dispatch.hpp"

using namespace env2048;
using namespace env2048::host;

namespace {

// [step-kernel-source-begin]  (bench.py hashes the marked regions of the step path's sources to tell whether the committed
// rocprofv3 counters in profiles/step_pmc.json still describe this step kernel)
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
__constant__ uint32_t DIR_TABLE[32] = {ENV2048_DIR_SELECTOR_VALUES};

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
    uint8_t *snapshot = nullptr;  // RESIDENT only: where the first launch of a direct-dispatch chain copies the boards it loads
};

__device__ __forceinline__ DirSelectors load_selectors(const uint4 *dir_lds, uint32_t action)
{
    const uint4 f = dir_lds[2u * action], i = dir_lds[2u * action + 1u];
    return DirSelectors{f.x, f.y, f.z, f.w, i.x, i.y, i.z, i.w};
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
// RESIDENT (the direct-dispatch chain of direct_dispatch.hpp, launches whose packets carry no cache fences): boards are loaded
// at agent scope (sc1 buffer loads: past this CU's vector L1, which nothing invalidates between such launches and which may
// hold the tile as ANOTHER launch's workgroup on this CU left it; the XCD's L2 is the home of the tile for the whole chain).
template <bool EXTRAS, bool FULL, bool RANDOM, int BOARDS_PER_LANE, bool RESIDENT = false, bool SNAPSHOT = false>
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
    // (... and with the resident launches of the direct route, as plain wide stores: 5.22 vs 5.30 us per 2^20-board step, profiles/r05_r_*)
    const bool wide_outputs = (through || RESIDENT) && FULL && BOARDS_PER_LANE >= 2 && (a.flags & INTERNAL_FLAG_WIDE_OUTPUTS) != 0u;

#ifdef ENV2048_EXP_STAMPS  // diagnostic build: wave timeline stamps into the (otherwise unused) obs buffer
    uint64_t stamp0 = __builtin_amdgcn_s_memrealtime(), stamp1 = 0, stamp2 = 0, stamp3 = 0, stamp4 = 0;
#endif
    // 1. loads of the FIRST half of the tile (boards 0,1) -- and the selector table
    uint4 table_row = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 8u) table_row = reinterpret_cast<const uint4 *>(DIR_TABLE)[tid];
    Board bd[BOARDS_PER_LANE];
    uint32_t act[BOARDS_PER_LANE];
    constexpr int HALF = BOARDS_PER_LANE / 2 > 0 ? BOARDS_PER_LANE / 2 : 1;
    // (RESIDENT: the tile as a buffer -- a wave-uniform descriptor over its 16 * TILE_BOARDS bytes, clipped at the end of the
    // batch, so a lane beyond it reads zeros; the lane's byte offset travels in the 32-bit voffset)
    const uint64_t tile_boards_here = FULL ? (uint64_t)TILE_BOARDS : (tile0 < a.n ? (a.n - tile0 < TILE_BOARDS ? a.n - tile0 : TILE_BOARDS) : 0u);
    TileBuffer tile;
    if constexpr (RESIDENT) tile = make_tile_buffer(a.boards, tile0, tile_boards_here);
    auto issue_loads = [&](int j) {
        const uint64_t i = tile0 + 64u * j + lane;
        bd[j] = Board{0u, 0u, 0u, 0u};
        act[j] = 0u;
        if (FULL || i < a.n) {
            if constexpr (RESIDENT)
                bd[j] = load_board_agent(tile, 64u * j + lane);
            else
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
        if constexpr (SNAPSHOT) {  // (the first launch of a direct-dispatch chain: the boards as the sequence found them)
            const uint64_t i = tile0 + 64u * j + lane;
            if (FULL || i < a.n) store_board(a.snapshot, i, bd[j]);
        }
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
            store_obs_any_coalesced(a.obs, strip, tile0 + 64u * j, a.n, lane, bd[j], auto_reset && done[j], through || RESIDENT,
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
            if (through) {
                if (a.reward) store_x4_through(a.reward + tile0 + 4u * lane, u32x4{r4.x, r4.y, r4.z, r4.w});
                store_x1_through(a.done + tile0 + 4u * lane, d4);
            } else {
                if (a.reward) reinterpret_cast<uint4 *>(a.reward + tile0)[lane] = r4;
                reinterpret_cast<uint32_t *>(a.done + tile0)[lane] = d4;
            }
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

// The step kernel of a DIRECT-DISPATCH chain (direct_dispatch.hpp): K dependent launches of one part of a batch written as AQL
// packets into a queue of the library's own, the packets between the first and the last WITHOUT cache fences.  What makes that
// correct: block k owns the same boards in every launch, and as long as workgroup k keeps landing on the same XCD the tile is
// only ever touched through ONE XCD's L2, which stays its home for the whole chain (plain write-back stores; nothing leaves
// the chip between steps but the write-once rows).  Across CUs inside the XCD the tile may move from launch to launch: hence
// the agent-scope loads (step_block<..., RESIDENT>).  reward / done are written and never read; actions are read-only for the
// chain.  "Keeps landing on the same XCD" is checked, not assumed -- see step_resident_body.
constexpr uint32_t RESIDENT_PRETEND_MOVED = 2u;  // (test hook of direct_dispatch.hpp) this launch's workgroups claim another XCD

template <bool EXTRAS, bool RANDOM, int BPL, bool FIRST>
__device__ __forceinline__ void step_resident_body(uint8_t *boards, const uint8_t *actions, uint64_t n, uint64_t env_base, uint32_t flags,
                                                   uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi, int32_t *reward,
                                                   uint8_t *done, uint32_t mode, uint32_t *owner, uint32_t *placement_error, uint8_t *snapshot,
                                                   uint8_t *legal, void *obs)
{
    // Where am I, and is that where this block's boards live?  The first launch of a chain (it reads the boards from memory:
    // its packet carries the acquire) writes the answer through to memory; every later launch reads it back past the caches it
    // cannot trust for data another launch wrote (agent scope: L2-served after the first miss) and compares AFTER the step --
    // the load's latency hides under the step, and a mismatch voids the whole sequence anyway (direct_dispatch.hpp reruns it).
    uint32_t xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    xcc_id = (xcc_id & 0xfu) ^ ((mode & RESIDENT_PRETEND_MOVED) ? 1u : 0u);
    uint32_t home = xcc_id;
    if constexpr (FIRST) {
        if (threadIdx.x == 0u) __hip_atomic_store(owner + blockIdx.x, xcc_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        home = __hip_atomic_load(owner + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    StepArgs a{n, env_base, boards, actions, reward, done, EXTRAS ? legal : nullptr, EXTRAS ? obs : nullptr,
               RngKey{seed_lo, seed_hi, step_lo, step_hi}, flags, nullptr};
    a.snapshot = snapshot;
    __shared__ uint4 dir_lds[8];
    __shared__ __attribute__((aligned(16))) uint32_t word_lds[WAVES_PER_BLOCK][word_strip_words(BPL)];
    __shared__ uint32_t queue_state;
    __shared__ uint16_t reset_slot[block_boards(BPL)];
    const uint64_t block0 = (uint64_t)blockIdx.x * block_boards(BPL);
    if (block0 + block_boards(BPL) <= a.n)
        step_block<EXTRAS, true, RANDOM, BPL, true, FIRST>(a, dir_lds, word_lds, &queue_state, reset_slot);
    else
        step_block<EXTRAS, false, RANDOM, BPL, true, FIRST>(a, dir_lds, word_lds, &queue_state, reset_slot);
    if (home != xcc_id && threadIdx.x == 0u)  // (block + 1, the XCD this workgroup ran on, the XCD its boards live on)
        __hip_atomic_store(placement_error, (1u + (blockIdx.x & 0xffffu)) | (xcc_id << 16) | ((home & 0xffu) << 24), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}
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

// [step-kernel-source-begin]
// Enqueue one step launch (arguments already validated, device already selected).
void launch_step(const env2048_step_args &a, hipStream_t stream)
{
    uint32_t kernel_flags = a.flags & ~(ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_STORES_WRITE_BACK | ENV2048_FLAG_TILE_SMALL |
                                        ENV2048_FLAG_TILE_LARGE | ENV2048_FLAG_SINGLE_CHAIN | ENV2048_FLAG_NO_DIRECT);
    if (!misaligned(a.reward, 16) && !misaligned(a.done, 4)) kernel_flags |= INTERNAL_FLAG_WIDE_OUTPUTS;
    if (plan_write_through(a.n, a.flags)) kernel_flags |= INTERNAL_FLAG_WRITE_THROUGH;
#ifdef ENV2048_EXP_STAMPS
    const bool extras = false;
#else
    const bool extras = a.legal || a.obs || a.max_exp || (a.flags & ENV2048_FLAG_REWARD_LOG2_F32);
#endif
    const bool random_actions = (a.flags & ENV2048_FLAG_RANDOM_ACTIONS) != 0u;
    const bool small = plan_small_tiling(a.n, a.flags);
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

// Enqueue the live records of [first, last) -- pairwise independent, same variant, `lead` the first of them, `total` their
// boards -- as one step_many_kernel launch.
void launch_group(const env2048_step_args *args, uint64_t first, uint64_t last, uint64_t lead, uint64_t total, hipStream_t stream)
{
    const bool small = plan_small_tiling(total, args[lead].flags);
    const int bpl = small ? 1 : BOARDS_PER_LANE;
    FusedTable table;
    uint32_t max_blocks = 0, records = 0;
    for (uint64_t k = first; k < last; ++k) {
        const env2048_step_args &a = args[k];
        if (a.n == 0) continue;
        uint32_t kernel_flags = a.flags & ~(ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_STORES_WRITE_BACK | ENV2048_FLAG_TILE_SMALL |
                                            ENV2048_FLAG_TILE_LARGE | ENV2048_FLAG_SINGLE_CHAIN | ENV2048_FLAG_NO_DIRECT);
        if (!misaligned(a.reward, 16) && !misaligned(a.done, 4)) kernel_flags |= INTERNAL_FLAG_WIDE_OUTPUTS;
        if (plan_write_through(total, a.flags)) kernel_flags |= INTERNAL_FLAG_WRITE_THROUGH;
        const uint32_t blocks = (uint32_t)((a.n + block_boards(bpl) - 1) / block_boards(bpl));
        max_blocks = blocks > max_blocks ? blocks : max_blocks;
        table.rec[records++] = FusedRecord{a.boards, a.actions, a.n, a.env_base, kernel_flags, (uint32_t)a.seed, (uint32_t)(a.seed >> 32),
                                           (uint32_t)a.step_index, (uint32_t)(a.step_index >> 32), blocks, a.reward, a.done, a.legal, a.obs,
                                           a.max_exp};
    }
    const StepVariant v = variant_of(args[lead]);
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
// (when: plan_two_chains in tiling.hpp; the helper thread, its poll window and its cost: helper_thread.hpp)
struct DeviceLanes {  // per device, created on first use, never destroyed
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;  // re-recorded by every call (calls are serialised by helper->call_mutex)
    hipError_t create_error = hipSuccess;
    HelperThread *helper = nullptr;
};

PerDeviceRegistry<DeviceLanes> g_lanes;

DeviceLanes *lanes_for(int device)  // (the caller has selected `device`)
{
    return g_lanes.get(device, [device] {
        DeviceLanes *L = new DeviceLanes;
        L->create_error = hipStreamCreateWithFlags(&L->side, hipStreamNonBlocking);
        if (L->create_error == hipSuccess) L->create_error = hipEventCreateWithFlags(&L->fork, hipEventDisableTiming);
        if (L->create_error == hipSuccess) L->create_error = hipEventCreateWithFlags(&L->join, hipEventDisableTiming);
        if (L->create_error == hipSuccess) L->helper = new HelperThread([device] { (void)hipSetDevice(device); });
        return L;
    });
}

#ifdef ENV2048_EXP_SPLIT_MIN_STEPS  // diagnostic build (tools/microbench): thresholds from the environment
inline bool want_two_chains(uint64_t n, uint32_t count, uint32_t flags)
{
    static const uint32_t min_steps = getenv("ENV2048_SPLIT_MIN_STEPS") ? (uint32_t)atoi(getenv("ENV2048_SPLIT_MIN_STEPS")) : SPLIT_MIN_STEPS;
    static const uint64_t max_boards = getenv("ENV2048_SPLIT_MAX_LOG2") ? 1ull << atoi(getenv("ENV2048_SPLIT_MAX_LOG2")) : SPLIT_MAX_BOARDS;
    return !(flags & ENV2048_FLAG_SINGLE_CHAIN) && n >= SPLIT_MIN_BOARDS && n <= max_boards && count >= min_steps;
}
#else
inline bool want_two_chains(uint64_t n, uint32_t count, uint32_t flags) { return plan_two_chains(n, count, flags); }
#endif

}  // namespace

// [step-kernel-source-begin]
// The entry points of the direct-dispatch code object carry C names: direct_dispatch.hpp finds them by name in the image it
// loads through ROCr (the HIP runtime never launches them).
#define ENV2048_RESIDENT_KERNEL(NAME, EXTRAS, RANDOM, BPL, FIRST)                                                                                    \
    extern "C" __global__ __launch_bounds__(BLOCK) void NAME(uint8_t *boards, const uint8_t *actions, uint64_t n, uint64_t env_base, uint32_t flags, \
                                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi, int32_t *reward, \
                                                             uint8_t *done, uint32_t mode, uint32_t *owner, uint32_t *placement_error,          \
                                                             uint8_t *snapshot, uint8_t *legal, void *obs)                                      \
    {                                                                                                                                           \
        step_resident_body<EXTRAS, RANDOM, BPL, FIRST>(boards, actions, n, env_base, flags, seed_lo, seed_hi, step_lo, step_hi, reward, done,   \
                                                       mode, owner, placement_error, snapshot, legal, obs);                                     \
    }
// (_first: the launch that opens a chain -- it records every workgroup's XCD and copies the boards it loads into the snapshot;
// _extras: legal mask / observation / log-normalised reward, the observation rows leaving write-through so that they do not push
// the boards out of the L2s)
#define ENV2048_RESIDENT_FAMILY(STEM, EXTRAS)                                                      \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_small##STEM, EXTRAS, false, 1, false)            \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_small##STEM##_random, EXTRAS, true, 1, false)    \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_large##STEM, EXTRAS, false, BOARDS_PER_LANE, false) \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_large##STEM##_random, EXTRAS, true, BOARDS_PER_LANE, false) \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_small##STEM##_first, EXTRAS, false, 1, true)     \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_small##STEM##_random_first, EXTRAS, true, 1, true) \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_large##STEM##_first, EXTRAS, false, BOARDS_PER_LANE, true) \
    ENV2048_RESIDENT_KERNEL(env2048_step_resident_large##STEM##_random_first, EXTRAS, true, BOARDS_PER_LANE, true)
ENV2048_RESIDENT_FAMILY(, false)
ENV2048_RESIDENT_FAMILY(_extras, true)
#undef ENV2048_RESIDENT_FAMILY
#undef ENV2048_RESIDENT_KERNEL

// [step-kernel-source-end]

extern "C" {

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
    for_each_launch(count, args, [&](uint64_t first, uint64_t last, uint64_t lead, uint64_t live, uint64_t total) {
        if (live == 1)
            launch_step(args[lead], (hipStream_t)stream);
        else if (live > 1)
            launch_group(args, first, last, lead, total, (hipStream_t)stream);
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
    return (int)plan_step_many(count, args, launch_of, boards_per_lane, write_through);
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
    // The direct route (direct_dispatch.hpp): the sequence as AQL packets on the library's own queues, boards resident in the
    // XCD L2s between the first and the last launch.  NOT_TAKEN: nothing was enqueued, the stream route below runs the call.
    if (direct::eligible(n, count, flags)) {
        const direct::SequenceArgs seq{n, env_base, boards, actions, reward, done, legal, obs, row_stride, seed, step_index, count, flags};
        const int rc = direct::step_sequence(device, main_stream, seq);
        if (rc != direct::NOT_TAKEN && rc != direct::RERUN) return rc;  // (RERUN: the boards are as before the call -- the launches below redo it)
    }
    bool two_chains = want_two_chains(n, count, flags);
    // Another thread of the process may have a global-mode stream capture open: the calls below that are not plain launches
    // (capture query, stream / event creation) would invalidate it, so this thread takes the relaxed mode while it makes them.
    struct RelaxedCaptureMode {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        bool active;
        explicit RelaxedCaptureMode(bool on) : active(on && hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess) {}
        ~RelaxedCaptureMode()
        {
            if (active) (void)hipThreadExchangeStreamCaptureMode(&mode);
        }
    } relaxed(two_chains);
    if (two_chains && main_stream) {  // a stream that is being captured into a graph takes the plain form (no cross-thread work
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;  // inside a capture); the legacy NULL stream cannot be captured
        if (hipStreamIsCapturing(main_stream, &capture) != hipSuccess || capture != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            two_chains = false;
        }
    }
    DeviceLanes *L = two_chains ? lanes_for(device) : nullptr;
    if (!L || L->create_error != hipSuccess || !L->helper) {
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
    std::lock_guard<std::mutex> call(L->helper->call_mutex);  // one split call at a time per device
    hipError_t e = hipEventRecord(L->fork, main_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(L->side, L->fork, 0);
    if (e != hipSuccess) return hip_fail(e, "env2048_step_sequence: fork of the second chain");
    L->helper->post([&]() -> int {
        for (uint32_t t = 0; t < count; ++t) launch_step(half(t, 1), L->side);
        return (int)hipGetLastError();
    });
    for (uint32_t t = 0; t < count; ++t) launch_step(half(t, 0), main_stream);
    const hipError_t mine = hipGetLastError();
    const hipError_t theirs = (hipError_t)L->helper->wait();
    e = hipEventRecord(L->join, L->side);
    if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, L->join, 0);
    if (mine != hipSuccess) return hip_fail(mine, "env2048_step_sequence launch");
    if (theirs != hipSuccess) return hip_fail(theirs, "env2048_step_sequence launch (second chain)");
    if (e != hipSuccess) return hip_fail(e, "env2048_step_sequence: join of the second chain");
    return ENV2048_OK;
}

// What the library's helper thread for `device` has cost so far (helper_thread.hpp).  No device call is made.
int env2048_helper_stats(int device, env2048_helper_info *out)
{
    if (!out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_helper_stats: out is NULL");
    DeviceLanes *L = g_lanes.peek(device);
    const HelperThread *h = L ? L->helper : nullptr;
    out->created = h ? 1u : 0u;
    out->spin_us = h ? h->spin_us() : helper_spin_us_from_env();
    out->jobs = h ? h->jobs() : 0u;
    out->cpu_ns = h ? h->cpu_ns() : 0u;
    return ENV2048_OK;
}

// The direct route of `device` (direct_dispatch.hpp): starts it up on first use.
int env2048_direct_stats(int device, env2048_direct_info *out)
{
    if (!out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_direct_stats: out is NULL");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    direct::Device *d = direct::device_for(device);
    memset(out, 0, sizeof *out);
    if (!d) return fail(ENV2048_ERR_DEVICE, "env2048_direct_stats: device %d out of range", device);
    std::lock_guard<std::mutex> call(d->call_mutex);
    out->available = d->ok ? 1u : 0u;
    out->last_chains = d->last_chains;
    out->reruns = d->reruns;
    out->calls = d->calls;
    out->steps = d->steps;
    out->packets = d->packets;
    out->declined = d->declined;
    out->last_gpu_ns = d->last_gpu_ns;
    snprintf(out->reason, sizeof out->reason, "%s", d->reason);
    return ENV2048_OK;
}

int env2048_direct_eligible(uint64_t n, uint32_t count, uint32_t flags, int32_t extras)
{
    if (flags & ~STEP_FLAGS) return fail(ENV2048_ERR_INVALID_ARG, "env2048_direct_eligible: unknown flag bits 0x%x", flags);
    (void)extras;  // (since ABI 5 the route serves legal / obs / the normalised reward too)
    return direct::eligible(n, count, flags) ? 1 : 0;
}

int env2048_direct_plan(uint64_t n, uint32_t count, uint32_t flags, int32_t extras, env2048_direct_plan_info *out)
{
    if (!out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_direct_plan: out is NULL");
    if (flags & ~STEP_FLAGS) return fail(ENV2048_ERR_INVALID_ARG, "env2048_direct_plan: unknown flag bits 0x%x", flags);
    memset(out, 0, sizeof *out);
    direct::Parts parts;
    (void)extras;
    out->eligible = direct::eligible(n, count, flags) && direct::plan_parts(n, &parts) ? 1u : 0u;
    if (!out->eligible) return ENV2048_OK;
    out->chains = (uint32_t)parts.chains;
    for (int c = 0; c <= parts.chains; ++c) out->part_first[c] = parts.first[c];
    for (int c = 0; c < parts.chains; ++c) {
        out->part_blocks[c] = parts.blocks[c];
        out->part_boards_per_lane[c] = parts.small[c] ? 1u : (uint32_t)BOARDS_PER_LANE;
    }
    out->segment_steps = direct::SEGMENT_STEPS;
    for (uint32_t first = 0; first < count; first += direct::segment_steps(first, count)) ++out->segments;
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

}  // extern "C"
