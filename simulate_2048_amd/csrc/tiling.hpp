// tiling.hpp -- launch geometry and host launch policy constants shared by the kernels and the planner.
// [step-kernel-source-begin]
// Plain C++ (no HIP types): the planner that uses these is compiled by g++ for the CPU tests as well as by hipcc.
#pragma once
#include <stdint.h>

#include "env2048.h"

namespace env2048 {

#ifndef ENV2048_BLOCK
#define ENV2048_BLOCK 512
#endif
constexpr int BLOCK = ENV2048_BLOCK;
constexpr int WAVES_PER_BLOCK = BLOCK / 64;
#ifndef ENV2048_BOARDS_PER_LANE
#define ENV2048_BOARDS_PER_LANE 4
#endif
constexpr int BOARDS_PER_LANE = ENV2048_BOARDS_PER_LANE;  // the large-launch tiling (and the rollout kernels')
// Tile geometry as a function of the boards each lane handles.  The step kernel exists in two tilings: BPL = 4 (one
// Philox block feeds four boards of a lane's neighbourhood, constants and selector fetches amortise: best from 2^19
// boards up, where VALU issue is the limit) and BPL = 1 (a wave's dependent chain is one board long and four times
// as many CUs get work: 4.2 vs 5.2 us per step at 4 096 ... 262 144 boards, where latency is the limit).
constexpr int tile_boards(int bpl) { return 64 * bpl; }                             // per wave
constexpr int block_boards(int bpl) { return tile_boards(bpl) * WAVES_PER_BLOCK; }  // per block
constexpr int word_strip(int bpl)                                                   // uint32 words of a wave's LDS strip:
{                                                                                   // Philox blocks (+1 straddling, padded to
    return (tile_boards(bpl) + 4 + 4 + tile_boards(bpl) / 4 + 3) / 4 * 4;           // 16 B) + a byte per board, >= 256 words
}                                                                                   // where the obs transposition uses it
constexpr int word_strip_words(int bpl) { return word_strip(bpl) < 256 ? 256 : word_strip(bpl); }
constexpr uint64_t SMALL_TILING_MAX_BOARDS = 1ull << 18;  // launches up to this size use the BPL = 1 tiling
// Launches up to this many boards use write-through stores unless a STORES_* flag says otherwise: they win while
// boards + outputs of one launch stay within the 256 MiB Infinity Cache (measured 2^20..2^23 boards: -5..-13 %)
// and lose beyond it (2^24: +7 %).
constexpr uint64_t WRITE_THROUGH_MAX_BOARDS = 1ull << 23;

constexpr uint32_t INTERNAL_FLAG_WIDE_OUTPUTS = 0x80000000u;   // reward 16-B / done 4-B aligned: transposed wide stores
constexpr uint32_t INTERNAL_FLAG_WRITE_THROUGH = 0x40000000u;  // sc1 stores (host decides by launch size)

constexpr int MAX_FUSED_RECORDS = 32;  // records of one step_many_kernel launch: 32 x 96 B = 3 KiB of the 4 KiB argument block

// 16-bit observations (ENV2048_FLAG_OBS_F16 / _BF16): the same values -- e/16 with e <= 31 has five
// significant bits, exact in both formats -- at half the bytes, which is what bounds the trainer-facing path.
constexpr uint32_t OBS_F32 = 0u, OBS_BF16 = 2u;  // flags bits 2..3 (1 = f16)
constexpr uint32_t obs_format(uint32_t flags) { return (flags >> 2) & 3u; }

constexpr uint32_t STEP_FLAGS = ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS | ENV2048_FLAG_OBS_F16 | ENV2048_FLAG_OBS_BF16 |
                                ENV2048_FLAG_STORES_WRITE_THROUGH | ENV2048_FLAG_STORES_WRITE_BACK | ENV2048_FLAG_REWARD_LOG2_F32 |
                                ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE | ENV2048_FLAG_SINGLE_CHAIN | ENV2048_FLAG_NO_DIRECT;

// env2048_step_sequence: when K dependent steps of one batch run as two half-batch chains (step_kernels.hip)
constexpr uint64_t SPLIT_MIN_BOARDS = 1ull << 20;  // below: a half (< 2^19 boards) is latency- / launch-bound on its own (2^19: 6.1 vs 5.5 us)
constexpr uint64_t SPLIT_MAX_BOARDS = 1ull << 23;  // above: one launch already runs at the memory floor (2^23: 43.7 vs 48.4 us, 2^24: 118 vs 98)
constexpr uint32_t SPLIT_MIN_STEPS = 32;           // two enqueuing threads cost 6-7 us of host time per step while the queues fill: K = 20
                                                   // blocks are enqueue-bound (+3 % .. -3 %), K = 50: -10 %, 100: -12 %, >= 1000: -15..17 %

// Tiling and store policy of a launch over `total` boards (one record, or all records of a fused group together).
constexpr bool plan_small_tiling(uint64_t total, uint32_t flags)
{
    return (flags & ENV2048_FLAG_TILE_SMALL) ? true : (flags & ENV2048_FLAG_TILE_LARGE) ? false : total <= SMALL_TILING_MAX_BOARDS;
}
constexpr bool plan_write_through(uint64_t total, uint32_t flags)
{
    return (flags & ENV2048_FLAG_STORES_WRITE_THROUGH) ? true : (flags & ENV2048_FLAG_STORES_WRITE_BACK) ? false : total <= WRITE_THROUGH_MAX_BOARDS;
}
constexpr bool plan_two_chains(uint64_t n, uint32_t count, uint32_t flags)
{
    return !(flags & ENV2048_FLAG_SINGLE_CHAIN) && n >= SPLIT_MIN_BOARDS && n <= SPLIT_MAX_BOARDS && count >= SPLIT_MIN_STEPS;
}

constexpr uint64_t ROLLOUT_SMALL_MAX_ENVS = 1ull << 18;  // up to here one board per lane (more waves) beats shared Philox blocks
                                                         // (512 steps: 0.44-0.87 ms vs 1.29-1.41 ms; at 2^20: 3.04 vs 2.64 ms)
constexpr bool rollout_small(uint64_t n, uint32_t flags)
{
    return (flags & ENV2048_FLAG_TILE_SMALL) ? true : (flags & ENV2048_FLAG_TILE_LARGE) ? false : n <= ROLLOUT_SMALL_MAX_ENVS;
}

}  // namespace env2048
// [step-kernel-source-end]
