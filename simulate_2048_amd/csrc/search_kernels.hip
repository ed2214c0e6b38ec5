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
                                                           int32_t *reward, uint8_t *changed_out, bool through)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Board bd = load_board(boards_in, i);
    uint32_t score;
    bool changed;
    afterstate(bd, (uint32_t)actions[i] & 3u, score, changed);
    store_board(boards_out, i, bd, through);
    if (reward) reward[i] = (int32_t)score;
    if (changed_out) changed_out[i] = changed ? 1 : 0;
}

// Afterstates of ALL four actions of every board (the expansion step of a search): the byte transpose is
// computed once, up/down slide the row dwords as planes, left/right the transposed ones, and the two
// directions of an axis differ only in the order the planes are handed to slide_merge_planes.
// out_boards[4*i + a], out_reward[4*i + a]; out_legal[i] bit a = action a changes the board.
// The four afterstates of board i lie side by side (64 bytes), so a lane storing its own four would put 16 bytes into each of 64
// different 64-byte segments with every store instruction; instead the wave's 256 afterstates go through a wave-private LDS
// strip ([action][lane]: conflict-free 16-byte writes) and leave as four store instructions of 1 KiB of consecutive bytes.
__global__ __launch_bounds__(BLOCK) void afterstates4_kernel(uint64_t n, const uint8_t *boards, uint8_t *out_boards,
                                                             int32_t *out_reward, uint8_t *out_legal, bool through)
{
    __shared__ uint4 strips[BLOCK / 64][4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t first = (uint64_t)blockIdx.x * BLOCK + 64u * wave;
    if (first >= n) return;  // (a whole wave; nothing below is shared between waves)
    const uint64_t i = first + lane;
    const bool active = i < n;
    const Board bd = active ? load_board(boards, i) : Board{0u, 0u, 0u, 0u};
    uint4(&strip)[4][64] = strips[wave];
    uint32_t c0, c1, c2, c3;
    transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, c0, c1, c2, c3);
    uint32_t legal = 0u;
    int4 rewards;
    {  // 0 = left: planes are the columns, wall side first
        uint32_t a = c0, b = c1, c = c2, d = c3, changed;
        rewards.x = (int32_t)slide_merge_planes(a, b, c, d, changed);
        Board out;
        transpose4x4(a, b, c, d, out.r0, out.r1, out.r2, out.r3);
        strip[0][lane] = make_uint4(out.r0, out.r1, out.r2, out.r3);
        legal |= changed != 0u ? 1u : 0u;
    }
    {  // 1 = up: planes are the rows
        uint32_t a = bd.r0, b = bd.r1, c = bd.r2, d = bd.r3, changed;
        rewards.y = (int32_t)slide_merge_planes(a, b, c, d, changed);
        strip[1][lane] = make_uint4(a, b, c, d);
        legal |= changed != 0u ? 2u : 0u;
    }
    {  // 2 = right: columns, last column is the wall
        uint32_t a = c3, b = c2, c = c1, d = c0, changed;
        rewards.z = (int32_t)slide_merge_planes(a, b, c, d, changed);
        Board out;
        transpose4x4(d, c, b, a, out.r0, out.r1, out.r2, out.r3);
        strip[2][lane] = make_uint4(out.r0, out.r1, out.r2, out.r3);
        legal |= changed != 0u ? 4u : 0u;
    }
    {  // 3 = down: rows, last row is the wall
        uint32_t a = bd.r3, b = bd.r2, c = bd.r1, d = bd.r0, changed;
        rewards.w = (int32_t)slide_merge_planes(a, b, c, d, changed);
        strip[3][lane] = make_uint4(d, c, b, a);
        legal |= changed != 0u ? 8u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the strip is wave-private: LDS ops of a wave are ordered
    __builtin_amdgcn_wave_barrier();
    uint4 *out = reinterpret_cast<uint4 *>(out_boards) + 4u * first;
    const uint64_t vectors = 4u * (n - first < 64u ? n - first : 64u);
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t q = 64u * j + lane;  // afterstate q of the wave: board q >> 2, action q & 3
        if (q < vectors) {
            const uint4 v = strip[q & 3u][q >> 2];
            if (through)
                store_x4_through(out + q, u32x4{v.x, v.y, v.z, v.w});
            else
                out[q] = v;
        }
    }
    if (!active) return;
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
struct D2After {  // one (board, action) entry of the LDS strip: 48 bytes, read as three 16-byte words
    uint32_t r0, r1, r2, r3;      // the afterstate
    uint32_t hv1, hv2, vv1, vv2;  // per-line merge exponents (line_merge_terms): rows as lines (byte j = row j), columns as lines
    uint32_t empties;             // cell mask of the afterstate's empty cells; 0 = action illegal
    uint32_t h_sum, v_sum;        // merge score of a horizontal / a vertical move on the afterstate (sum over its four lines)
    uint32_t reward;              // merge score of the move that led here (0 for an illegal action)
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

// Inclusive max-scan over the 64 lanes of a wave (values >= 0, identity 0): four row-local DPP shifts, then the last lane
// of rows 0 / 2 into rows 1 / 3 and the last lane of row 1 into rows 2 / 3 (the gfx9 row_bcast forms).
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)
{
#define ENV2048_SCAN_STEP(CTRL, ROWS) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xF, false))
    ENV2048_SCAN_STEP(0x111, 0xF);  // row_shr:1
    ENV2048_SCAN_STEP(0x112, 0xF);  // row_shr:2
    ENV2048_SCAN_STEP(0x114, 0xF);  // row_shr:4
    ENV2048_SCAN_STEP(0x118, 0xF);  // row_shr:8
    ENV2048_SCAN_STEP(0x142, 0xA);  // row_bcast:15 -> rows 1, 3
    ENV2048_SCAN_STEP(0x143, 0xC);  // row_bcast:31 -> rows 2, 3
#undef ENV2048_SCAN_STEP
    return v;
}

constexpr int D2_BLOCK = 256;
// A wave owns 32 boards, two lanes per board in phase 1.  Phase 2 is where the time goes and it runs with all 64 lanes whatever
// a wave owns; what the number sets is the LDS a wave needs (58 B per (board, action) entry), i.e. how many waves a CU holds
// to hide the LDS round trips of phase 2.  Measured on 4.5 M leaf boards: 64 boards per wave (8 waves per CU) 1080 us, 32 boards
// (20 waves per CU by LDS, then limited by registers) 829 us, 16 boards 1069 us (phase 1 half empty twice over).
constexpr int D2_WAVE_BOARDS = 32;
constexpr int D2_ENTRIES = 4 * D2_WAVE_BOARDS;
constexpr int D2_BLOCK_BOARDS = D2_WAVE_BOARDS * (D2_BLOCK / 64);

// Phase 2 works on CELLS, not successors: the tile-2 and the tile-4 successor of one empty cell of an afterstate differ from
// the afterstate in the same row and the same column only, and a move's merge score is a sum over its four lines -- so
//     best reward of a successor = max(h_sum - old row + new row,  v_sum - old column + new column)
// with the line scores of the afterstate computed once per (board, action) entry in phase 1 (line_merge_terms on the rows and
// on the columns) and the FOUR new lines of a cell (row and column, with a 2 and with a 4) scored by ONE four-line evaluation.
// That is one line evaluation per two successors instead of two whole-board slides per successor.
__global__ __launch_bounds__(D2_BLOCK) void expectimax_depth2_kernel(uint64_t n, const uint8_t *boards, int32_t *terms, float *q,
                                                                      uint8_t *best_action, float *value, double *value64)
{
    __shared__ __attribute__((aligned(16))) D2After strip_lds[D2_BLOCK / 64][D2_ENTRIES];
    __shared__ int32_t sums_lds[D2_BLOCK / 64][D2_ENTRIES][2];  // per (board, action): S2, S4
    __shared__ uint16_t head_lds[D2_BLOCK / 64][64];            // per 64-cell chunk: entry + 1 that STARTS at this position
    __shared__ uint16_t first_lds[D2_BLOCK / 64][D2_ENTRIES];   // first work item of every entry
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t board0 = ((uint64_t)blockIdx.x * (D2_BLOCK / 64) + wave) * (uint64_t)D2_WAVE_BOARDS;
    if (board0 >= n) return;  // wave-uniform
    D2After *strip = strip_lds[wave];
    int32_t(*sums)[2] = sums_lds[wave];
    uint16_t *head = head_lds[wave];
    uint16_t *first_of = first_lds[wave];

    // phase 1: TWO lanes per board -- lane b takes the horizontal moves (actions 0 and 2: the transposed board as planes),
    // lane b + 32 the vertical ones (1 and 3: the rows as planes) -> per action the afterstate, its line scores, its empty cells
    const uint32_t b = lane & 31u;
    const bool vertical = lane >= 32u;
    const uint64_t mine = board0 + b;
    const bool has_board = mine < n;
    const uint32_t e_first = 4u * b + (vertical ? 1u : 0u), e_second = e_first + 2u;  // my two entries
    uint32_t my_cells[2] = {0u, 0u};
    if (has_board) {
        const Board bd = load_board(boards, mine);
        uint32_t t0, t1, t2, t3;
        transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, t0, t1, t2, t3);
        // the planes of my first action (0: columns, 1: rows); my second action (2, 3) takes them in reverse order
        const uint32_t x0 = vertical ? bd.r0 : t0, x1 = vertical ? bd.r1 : t1, x2 = vertical ? bd.r2 : t2, x3 = vertical ? bd.r3 : t3;
#pragma unroll
        for (uint32_t second = 0; second < 2u; ++second) {
            uint32_t pa = second ? x3 : x0, pb = second ? x2 : x1, pc = second ? x1 : x2, pd = second ? x0 : x3, changed;
            const int32_t reward = (int32_t)slide_merge_planes(pa, pb, pc, pd, changed);
            const uint32_t y0 = second ? pd : pa, y1 = second ? pc : pb, y2 = second ? pb : pc, y3 = second ? pa : pd;  // planes in board order
            uint32_t u0, u1, u2, u3;
            transpose4x4(y0, y1, y2, y3, u0, u1, u2, u3);
            // afterstate as rows (as) and as columns (ac): for a vertical move the planes ARE the rows
            const Board as{vertical ? y0 : u0, vertical ? y1 : u1, vertical ? y2 : u2, vertical ? y3 : u3};
            const Board ac{vertical ? u0 : y0, vertical ? u1 : y1, vertical ? u2 : y2, vertical ? u3 : y3};
            const bool is_legal = changed != 0u;
            const uint32_t empt = is_legal ? empty_mask(as) : 0u;  // a legal move always leaves an empty cell
            uint32_t hv1, hv2, vv1, vv2;
            line_merge_terms(ac.r0, ac.r1, ac.r2, ac.r3, hv1, hv2);  // rows as lines: plane k = column k
            line_merge_terms(as.r0, as.r1, as.r2, as.r3, vv1, vv2);  // columns as lines: plane k = row k
            const uint32_t e = second ? e_second : e_first;
            strip[e] = D2After{as.r0, as.r1, as.r2, as.r3, hv1, hv2, vv1, vv2, empt, line_score_sum(hv1, hv2), line_score_sum(vv1, vv2),
                               is_legal ? (uint32_t)reward : 0u};
            sums[e][0] = 0;
            sums[e][1] = 0;
            my_cells[second] = ENV2048_POPC(empt);
        }
    } else {
        strip[e_first].empties = 0u;  // (the scan below reads the cell masks of all 128 entries)
        strip[e_second].empties = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // strip, sums, head and first_of are wave-private: LDS ops of a wave are ordered
    __builtin_amdgcn_wave_barrier();
    // empty cells are WORK ITEMS numbered densely over the wave's 128 (board, action) entries: entry e owns items
    // [first_of[e], first_of[e] + cells[e]).  On mid-game boards 17 of the 64 (action, cell) slots per board hold one, so
    // walking the slots would leave three quarters of the lanes idle.  Lane b (< 32) counts board b's four entries; exclusive
    // scan over the wave (the upper lanes contribute nothing).
    uint32_t cells[4] = {0u, 0u, 0u, 0u};
    if (!vertical) {
#pragma unroll
        for (uint32_t act = 0; act < 4u; ++act) cells[act] = ENV2048_POPC(strip[4u * b + act].empties);
    }
    const uint32_t mine_count = cells[0] + cells[1] + cells[2] + cells[3];
    uint32_t scan = mine_count;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = __shfl_up(scan, d);
        scan += lane >= d ? up : 0u;
    }
    const uint32_t total = __shfl(scan, 63);
    if (!vertical) {
        const uint32_t f0 = scan - mine_count, f1 = f0 + cells[0], f2 = f1 + cells[1], f3 = f2 + cells[2];
        first_of[4u * b + 0u] = (uint16_t)f0;
        first_of[4u * b + 1u] = (uint16_t)f1;
        first_of[4u * b + 2u] = (uint16_t)f2;
        first_of[4u * b + 3u] = (uint16_t)f3;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t my_first[2] = {first_of[e_first], first_of[e_second]};  // where my two entries start (for the heads below)

    // phase 2: 64 cells per iteration.  Which entry an item belongs to: every entry that STARTS inside the chunk drops its
    // number at its first position, a max-scan over the lanes spreads it to the items behind it, and items in front of the
    // chunk's first start belong to the entry the previous chunk ended in (`carry`).  Entry numbers grow with the item
    // number, entries without cells start nowhere.
    uint32_t carry = 0u;
    for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
        head[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t second = 0; second < 2u; ++second) {
            const uint32_t at = my_first[second] - w0;  // (wraps for entries in front of the chunk: >= 64 then)
            if (my_cells[second] != 0u && at < 64u) head[at] = (uint16_t)((second ? e_second : e_first) + 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t started = wave_scan_max(head[lane]);
        const uint32_t entry1 = max(started, carry);  // entry + 1 of this lane's item
        carry = __builtin_amdgcn_readlane(entry1, 63);
        const uint32_t w = w0 + lane;
        if (w < total) {
            const uint32_t e = entry1 - 1u;
            const uint4 rows = reinterpret_cast<const uint4 *>(&strip[e])[0];
            const uint4 lines = reinterpret_cast<const uint4 *>(&strip[e])[1];
            const uint4 rest = reinterpret_cast<const uint4 *>(&strip[e])[2];
            const uint32_t bit = kth_empty(rest.x, w - first_of[e]);  // my cell: the (w - first item of my entry)-th empty one
            const uint32_t pos = (uint32_t)__builtin_ctz(bit);  // 8 * column + {5, 6, 7, 4}[row]   (an entry with items has the bit)
            const uint32_t col8 = pos & 24u, row = (pos - 5u) & 3u, row8 = 8u * row;
            // row `row` of the afterstate, branch-free (left to itself hipcc turns the four-way choice into three divergent branches)
            const uint32_t odd = 0u - (row & 1u), upper = 0u - (row >> 1);
            const uint32_t old_row = ENV2048_BFI(upper, ENV2048_BFI(odd, rows.w, rows.z), ENV2048_BFI(odd, rows.y, rows.x));
            const uint32_t sel = 0x0C0C0400u + 0x0101u * (pos >> 3);  // byte `column` of a low and of a high dword
            const uint32_t old_col = byte_perm(byte_perm(rows.w, rows.z, sel), byte_perm(rows.y, rows.x, sel), 0x05040100u);
            // the afterstate's own score on these two lines, and the four new lines: (row, column) x (tile 2, tile 4)
            const uint32_t h_base = rest.y - line_score(lines.x, lines.y, row8);
            const uint32_t v_base = rest.z - line_score(lines.z, lines.w, col8);
            const uint32_t l0 = old_row | (1u << col8), l1 = old_col | (1u << row8), l2 = old_row | (2u << col8), l3 = old_col | (2u << row8);
            uint32_t p0, p1, p2, p3, v1, v2, f1, f2, s_row2, s_col2, s_row4, s_col4;
            transpose4x4(l0, l1, l2, l3, p0, p1, p2, p3);  // plane k = element k of the four lines
            line_merge_terms_flags(p0, p1, p2, p3, v1, v2, f1, f2);
            line_scores4(v1, v2, f1, f2, s_row2, s_col2, s_row4, s_col4);
            const uint32_t best2 = max(h_base + s_row2, v_base + s_col2);
            const uint32_t best4 = max(h_base + s_row4, v_base + s_col4);
            // integer sums: the order of the adds does not matter.  (Lanes of one entry hit the same two words; measured with the
            // atomics replaced by plain stores -- wrong sums -- the launch is 1.3 % shorter: not worth a segmented reduction.)
            atomicAdd(&sums[e][0], (int32_t)best2);
            atomicAdd(&sums[e][1], (int32_t)best4);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // phase 3: one board per lane again (lanes 0 .. 31)
    if (vertical || !has_board) return;
    // Q2 = r + (0.9 S2 + 0.1 S4) / n = X / D with the INTEGERS X = 10 n r + 9 S2 + S4 and D = 10 n: the arg-max compares the
    // fractions exactly (cross-multiplied in 64 bits; a tie keeps the smaller action) and the float outputs are ONE correctly
    // rounded double division each -- so device and oracle agree bit for bit, independent of fma contraction or summation order
    int64_t best_x = 0, best_d = 1;
    uint32_t best_a = 255u;
    float q4[4];
    int4 t[4];
#pragma unroll
    for (uint32_t act = 0; act < 4u; ++act) {
        const int32_t s2 = sums[4u * b + act][0], s4 = sums[4u * b + act][1], n_empty = (int32_t)cells[act];
        const int32_t reward = (int32_t)strip[4u * b + act].reward;
        t[act] = make_int4(reward, s2, s4, n_empty);
        if (n_empty != 0) {
            const int64_t d = 10 * (int64_t)n_empty, x = d * (int64_t)reward + 9 * (int64_t)s2 + (int64_t)s4;
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

// ---- offsets of a level on the device: count, scan, total -- no host round trip per level ---------------------------------
// The block of these launches is the header's ENV2048_EXPAND_OFFSETS_BLOCK, not the tunable BLOCK of tiling.hpp: the caller sizes `scratch`
// from it.
constexpr int OFFSETS_BLOCK = ENV2048_EXPAND_OFFSETS_BLOCK;
static_assert(OFFSETS_BLOCK % 64 == 0 && OFFSETS_BLOCK <= 1024, "expand_offsets_kernel: whole waves, one workgroup");
// expand_offsets_kernel: children of every board of a block of OFFSETS_BLOCK boards + the exclusive scan inside the block (wave scan
// by DPP-free shuffles, then the eight wave totals through LDS): offsets[i] = children of the block's boards before i,
// block_total[block] = children of the block.
__global__ __launch_bounds__(OFFSETS_BLOCK) void expand_offsets_kernel(uint64_t n, const uint8_t *boards, uint64_t *offsets, uint64_t *block_total)
{
    __shared__ uint32_t wave_total[OFFSETS_BLOCK / 64];
    const uint64_t i = (uint64_t)blockIdx.x * OFFSETS_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t count = 0u;
    if (i < n) {
        Board after[4];
        int32_t rewards[4];
        uint32_t legal;
        four_afterstates(load_board(boards, i), after, rewards, legal);
#pragma unroll
        for (uint32_t act = 0; act < 4u; ++act)
            if ((legal >> act) & 1u) count += 2u * ENV2048_POPC(empty_mask(after[act]));
    }
    uint32_t scan = count;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = __shfl_up(scan, d);
        scan += lane >= d ? up : 0u;
    }
    if (lane == 63u) wave_total[wave] = scan;
    __syncthreads();
    uint32_t before = 0u, all = 0u;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)(OFFSETS_BLOCK / 64); ++w) {
        const uint32_t t = wave_total[w];
        before += w < wave ? t : 0u;
        all += t;
    }
    if (i < n) offsets[i] = (uint64_t)(before + scan - count);
    if (threadIdx.x == 0u) block_total[blockIdx.x] = all;
}

// One workgroup: exclusive scan of the block totals in place (-> the first child of every block), the grand total behind
// them, and -- for a host that has to size the next frontier -- the total + a completion word in pinned host memory.
__global__ __launch_bounds__(1024) void scan_block_totals_kernel(uint64_t blocks, uint64_t *block_total, uint64_t *host_total, uint64_t sequence_value)
{
    __shared__ uint64_t part[1024];
    const uint64_t per = (blocks + 1023u) / 1024u, lo = min(blocks, (uint64_t)threadIdx.x * per), hi = min(blocks, lo + per);
    uint64_t sum = 0;
    for (uint64_t k = lo; k < hi; ++k) sum += block_total[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint64_t run = part[threadIdx.x] - sum;  // children before my chunk
    for (uint64_t k = lo; k < hi; ++k) {
        const uint64_t t = block_total[k];
        block_total[k] = run;
        run += t;
    }
    if (threadIdx.x == 1023u) {
        const uint64_t total = part[1023];
        block_total[blocks] = total;
        if (host_total) {
            host_total[0] = total;
            __threadfence_system();
            __hip_atomic_store(host_total + 1, sequence_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(OFFSETS_BLOCK) void add_block_base_kernel(uint64_t n, uint64_t *offsets, const uint64_t *block_base)
{
    const uint64_t i = (uint64_t)blockIdx.x * OFFSETS_BLOCK + threadIdx.x;
    if (i < n) offsets[i] += block_base[blockIdx.x];
}

// Children of a level, written the way a wave writes best: CONSECUTIVE lanes produce CONSECUTIVE children.  Phase 1, one
// board per lane: the four afterstates and their empty-cell masks into a wave-private LDS strip, reward / n_empty out.  Phase 2,
// 64 children per iteration (the dense numbering of the depth-2 kernel: every (board, action) entry that STARTS in the chunk
// drops its number at its first position, a DPP max-scan spreads it): lane l builds child w0 + l of the wave -- k-th empty cell
// of its entry's afterstate, tile 2 or 4 -- and stores its 16 bytes at the entry's global start + its rank there, so a store
// instruction of the wave writes 1 KiB contiguous (round 3: one lane walked the children of one board: 16-byte stores 544 B
// apart).  The global start of an entry comes from the caller's `offsets`, so any (even non-contiguous) offsets are honoured.
constexpr int EXPAND_BLOCK = 256;
struct ExpandEntry {  // 32 bytes
    uint32_t r0, r1, r2, r3;  // the afterstate
    uint32_t empties;         // its empty cells; 0 = illegal action
    uint32_t first;           // first child of this entry in the wave's dense numbering
    uint64_t start;           // global index of that child
};

__global__ __launch_bounds__(EXPAND_BLOCK) void expand_kernel(uint64_t n, const uint8_t *boards, const uint64_t *offsets, uint8_t *children,
                                                              int32_t *reward, uint8_t *n_empty)
{
    __shared__ __attribute__((aligned(16))) ExpandEntry strip_lds[EXPAND_BLOCK / 64][64 * 4];
    __shared__ uint16_t head_lds[EXPAND_BLOCK / 64][64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t board0 = ((uint64_t)blockIdx.x * (EXPAND_BLOCK / 64) + wave) * 64u;
    if (board0 >= n) return;  // wave-uniform
    ExpandEntry *strip = strip_lds[wave];
    uint16_t *head = head_lds[wave];
    const uint64_t mine = board0 + lane;
    uint32_t count[4] = {0u, 0u, 0u, 0u};  // children of my board's four actions
    uint32_t empt[4] = {0u, 0u, 0u, 0u};
    Board after[4] = {};
    uint64_t my_start = 0;
    if (mine < n) {
        int32_t rewards[4];
        uint32_t legal;
        four_afterstates(load_board(boards, mine), after, rewards, legal);
#pragma unroll
        for (uint32_t act = 0; act < 4u; ++act) {
            const bool ok = ((legal >> act) & 1u) != 0u;
            empt[act] = ok ? empty_mask(after[act]) : 0u;  // a legal move always leaves an empty cell
            count[act] = 2u * ENV2048_POPC(empt[act]);
            rewards[act] = ok ? rewards[act] : 0;
        }
        reinterpret_cast<int4 *>(reward)[mine] = make_int4(rewards[0], rewards[1], rewards[2], rewards[3]);
        reinterpret_cast<uint32_t *>(n_empty)[mine] = (count[0] >> 1) | ((count[1] >> 1) << 8) | ((count[2] >> 1) << 16) | ((count[3] >> 1) << 24);
        my_start = offsets[mine];
    }
    const uint32_t mine_count = count[0] + count[1] + count[2] + count[3];
    uint32_t scan = mine_count;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = __shfl_up(scan, d);
        scan += lane >= d ? up : 0u;
    }
    const uint32_t total = __shfl(scan, 63);
    uint32_t first[4];
    first[0] = scan - mine_count;
    first[1] = first[0] + count[0];
    first[2] = first[1] + count[1];
    first[3] = first[2] + count[2];
#pragma unroll
    for (uint32_t act = 0; act < 4u; ++act)
        strip[4u * lane + act] = ExpandEntry{after[act].r0, after[act].r1, after[act].r2, after[act].r3, empt[act], first[act],
                                             my_start + (first[act] - first[0])};
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // strip and head are wave-private: LDS ops of a wave are ordered
    __builtin_amdgcn_wave_barrier();
    uint32_t carry = 0u;
    for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
        head[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t act = 0; act < 4u; ++act) {
            const uint32_t at = first[act] - w0;  // (wraps for entries in front of the chunk: >= 64 then)
            if (count[act] != 0u && at < 64u) head[at] = (uint16_t)(4u * lane + act + 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t entry1 = max(wave_scan_max(head[lane]), carry);
        carry = __builtin_amdgcn_readlane(entry1, 63);
        const uint32_t w = w0 + lane;
        if (w < total) {
            const uint4 rows = reinterpret_cast<const uint4 *>(&strip[entry1 - 1u])[0];
            const uint4 rest = reinterpret_cast<const uint4 *>(&strip[entry1 - 1u])[1];  // empties, first, start (lo, hi)
            const uint32_t j = w - rest.y;                                               // my rank among the entry's children
            Board child{rows.x, rows.y, rows.z, rows.w};
            place_tile(child, kth_empty(rest.x, j >> 1), (j & 1u) != 0u);
            store_board(children, (((uint64_t)rest.w << 32) | rest.z) + j, child, true);  // (write-through, like every wide output)
        }
    }
}

// Values of a level from the values of its children: Q(s, a) = r + (0.9 S2 + 0.1 S4) / num_empty with S2 / S4 summed IN CHILD
// ORDER, every operation a separate correctly rounded double operation (the oracle's order of additions; fp contract off).
// One lane per (board, action): four times the parallelism of a lane per board, a lane's (S2, S4) pair per cell is one 16-byte
// load when the value array is 16-byte aligned (child ranges start at even indices), and the four lanes of a board sit in
// one quad: the arg-max is four DPP reads.
// the value lane A of my quad holds (DPP quad_perm broadcast; a double is two dwords)
template <int A>
__device__ __forceinline__ double quad_broadcast(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), A * 0x55, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), A * 0x55, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <bool ALIGNED16>
__global__ __launch_bounds__(BLOCK) void expectimax_backup_kernel(uint64_t n, const int32_t *reward, const uint8_t *n_empty,
                                                                  const uint64_t *offsets, const double *child_value, double *value,
                                                                  double *q, uint8_t *best_action)
{
#pragma clang fp contract(off)  // plain operators below, created under contract(off): no fma for 0.9 * S2 + 0.1 * S4 (HIP's __dmul_rn /
    // __dadd_rn are inline functions built under the default mode -- inlined, their multiply and add DO fuse)
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t i = t >> 2;
    const uint32_t act = (uint32_t)t & 3u;
    const bool valid = i < n;
    double qa = -__builtin_inf();
    if (valid) {
        const uint32_t ne4 = reinterpret_cast<const uint32_t *>(n_empty)[i];  // the four n_empty bytes of my board
        const uint32_t ne = (ne4 >> (8u * act)) & 0xFFu;
        const uint32_t before = act == 0u ? 0u : act == 1u ? (ne4 & 0xFFu) : act == 2u ? (ne4 & 0xFFu) + ((ne4 >> 8) & 0xFFu)
                                                                                     : (ne4 & 0xFFu) + ((ne4 >> 8) & 0xFFu) + ((ne4 >> 16) & 0xFFu);
        if (ne) {
            const double *at = child_value + offsets[i] + 2u * before;
            double s2 = 0.0, s4 = 0.0;
            // (offsets that are prefix sums of child counts are even; a caller's odd offset takes the 8-byte loads, it never faults)
            if (ALIGNED16 && (reinterpret_cast<uintptr_t>(at) & 15u) == 0u) {
                for (uint32_t k = 0; k < ne; ++k) {
                    const double2 pair = reinterpret_cast<const double2 *>(at)[k];
                    s2 = s2 + pair.x;
                    s4 = s4 + pair.y;
                }
            } else {
                for (uint32_t k = 0; k < ne; ++k) {
                    s2 = s2 + at[2u * k];
                    s4 = s4 + at[2u * k + 1u];
                }
            }
            const double t2 = 0.9 * s2, t4 = 0.1 * s4;
            const double mix = t2 + t4;
            const double mean = mix / (double)ne;
            qa = (double)reward[t] + mean;
        }
        if (q) q[t] = qa;
    }
    // the four actions of a board are the four lanes of a quad: first maximum over the legal ones (quad_perm broadcasts)
    double best_q = 0.0;
    uint32_t best_a = 255u;
    const double of_action[4] = {quad_broadcast<0>(qa), quad_broadcast<1>(qa), quad_broadcast<2>(qa), quad_broadcast<3>(qa)};
#pragma unroll
    for (uint32_t a2 = 0; a2 < 4u; ++a2) {
        if (of_action[a2] != -__builtin_inf() && (best_a == 255u || of_action[a2] > best_q)) {
            best_q = of_action[a2];
            best_a = a2;
        }
    }
    if (valid && act == 0u) {
        value[i] = best_a == 255u ? 0.0 : best_q;
        if (best_action) best_action[i] = (uint8_t)best_a;
    }
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
    store_board(boards, i, bd, true);  // (write-through, like every wide output)
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
    store_board(boards, i, bd, true);
}
// one thread per (board, outcome slot): 32 consecutive lanes write the 512 B of one board's outcomes.  STREAM: the output is
// larger than the Infinity Cache and nobody on the chip reads it back soon -- non-temporal stores (chosen on the host by size).
template <bool STREAM>
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
    if (STREAM) {
        typedef uint32_t vec4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(vec4{bd.r0, bd.r1, bd.r2, bd.r3}, reinterpret_cast<vec4 *>(out_boards) + t);
        __builtin_nontemporal_store(prob, out_prob + t);
    } else {
        store_board(out_boards, t, bd);  // (inside the Infinity Cache: plain stores -- write-through measured 4 % slower here)
        out_prob[t] = prob;
    }
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
                       boards_out, reward, changed, wide_stores_through());
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
                       out_legal, wide_stores_through());
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
    const char *force = getenv("ENV2048_EXP_OUTCOME_STORES");  // (A/B aid of tools/bench_all_entries.py: "stream" / "plain")
    const bool streaming = force ? force[0] == 's' : n * 641u > (256ull << 20);  // (the 256 MiB Infinity Cache)  // 32 x (16 + 4) + 1 bytes per board
    if (streaming)
        hipLaunchKernelGGL(chance_outcomes_kernel<true>, grid_for(32 * n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, out_boards, out_prob,
                           out_count);
    else
        hipLaunchKernelGGL(chance_outcomes_kernel<false>, grid_for(32 * n), dim3(BLOCK), 0, (hipStream_t)stream, n, boards, out_boards, out_prob,
                           out_count);
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
    hipLaunchKernelGGL(expectimax_depth2_kernel, dim3((unsigned)((n + D2_BLOCK_BOARDS - 1) / D2_BLOCK_BOARDS)), dim3(D2_BLOCK), 0, (hipStream_t)stream, n,
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
    hipLaunchKernelGGL(expectimax_depth2_kernel, dim3((unsigned)((n + D2_BLOCK_BOARDS - 1) / D2_BLOCK_BOARDS)), dim3(D2_BLOCK), 0, (hipStream_t)stream, n,
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

uint64_t env2048_expand_scratch_words(uint64_t n) { return (n + ENV2048_EXPAND_OFFSETS_BLOCK - 1) / ENV2048_EXPAND_OFFSETS_BLOCK + 1; }

int env2048_expand_offsets(int device, void *stream, uint64_t n, const uint8_t *boards, uint64_t *offsets, uint64_t *scratch,
                           uint64_t *host_total, uint64_t sequence_value)
{
    if (n == 0) return ENV2048_OK;
    if (!boards || !offsets || !scratch) return fail(ENV2048_ERR_INVALID_ARG, "env2048_expand_offsets: boards, offsets and scratch are required");
    if (misaligned(boards, 16) || misaligned(offsets, 8) || misaligned(scratch, 8) || misaligned(host_total, 8))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_expand_offsets: boards must be 16-byte, offsets / scratch / host_total 8-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const uint64_t blocks = env2048_expand_scratch_words(n) - 1;
    hipLaunchKernelGGL(expand_offsets_kernel, dim3((unsigned)blocks), dim3(OFFSETS_BLOCK), 0, (hipStream_t)stream, n, boards, offsets, scratch);
    hipLaunchKernelGGL(scan_block_totals_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, blocks, scratch, host_total, sequence_value);
    hipLaunchKernelGGL(add_block_base_kernel, dim3((unsigned)blocks), dim3(OFFSETS_BLOCK), 0, (hipStream_t)stream, n, offsets, scratch);
    return launched("env2048_expand_offsets launch");
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
    if (misaligned(reward, 16) || misaligned(n_empty, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_expand: reward must be 16-byte, n_empty 4-byte aligned");
    hipLaunchKernelGGL(expand_kernel, dim3((unsigned)((n + EXPAND_BLOCK - 1) / EXPAND_BLOCK)), dim3(EXPAND_BLOCK), 0, (hipStream_t)stream, n, boards,
                       offsets, children, reward, n_empty);
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
    if (misaligned(n_empty, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_expectimax_backup: n_empty must be 4-byte aligned");
    if (int rc = check_count(4 * n)) return rc;
    if (misaligned(child_value, 16))
        hipLaunchKernelGGL(expectimax_backup_kernel<false>, grid_for(4 * n), dim3(BLOCK), 0, (hipStream_t)stream, n, reward, n_empty, offsets,
                           child_value, value, q, best_action);
    else
        hipLaunchKernelGGL(expectimax_backup_kernel<true>, grid_for(4 * n), dim3(BLOCK), 0, (hipStream_t)stream, n, reward, n_empty, offsets,
                           child_value, value, q, best_action);
    return launched("env2048_expectimax_backup launch");
}

}  // extern "C"
