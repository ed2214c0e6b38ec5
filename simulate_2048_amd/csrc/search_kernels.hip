// search_kernels.hip -- true-environment search building blocks: afterstates, chance outcomes, expectimax ply / depth 2 /
// depth d (expand, back-up), injected spawns; kernels and their C-ABI entry points of include/env2048.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "board_io.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"

using namespace env2048;
using namespace env2048::host;

namespace {
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

}  // namespace

extern "C" {

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

}  // extern "C"
