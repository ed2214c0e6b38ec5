// board_ops.hpp -- per-lane 2048 board arithmetic for gfx950 (CDNA4), header-only device code.
//
// One board lives in four VGPRs: dword r = row r, byte c = column c, byte value = log2(tile)
// (0 = empty).  All game logic is branch-free SWAR on those dwords: the four lines a move slides
// along are processed at once, one line per byte lane.
//
// Semantics follow the reference NumPy engine; the lines each routine restates are cited as
// [ref: file:lines] relative to the reference repo root.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace env2048 {

constexpr uint32_t K7F = 0x7F7F7F7Fu;
constexpr uint32_t K80 = 0x80808080u;
constexpr uint32_t K01 = 0x01010101u;

// P(tile 4) = 0.1 [ref: twentyfortyeight/core/gameboard.py:13,16-17] -> word < floor(0.1 * 2^32)
constexpr uint32_t SPAWN4_THRESHOLD = 429496729u;

constexpr uint32_t PHILOX_M0 = 0xD2511F53u;
constexpr uint32_t PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u;
constexpr uint32_t PHILOX_W1 = 0xBB67AE85u;

struct Board {
    uint32_t r0, r1, r2, r3;
};

// v_perm_b32: selector byte 0-3 picks a byte of `lo`, 4-7 a byte of `hi`.
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// (a & m) | (b & ~m) -- one v_bfi_b32.
__device__ __forceinline__ uint32_t bsel(uint32_t m, uint32_t a, uint32_t b)
{
    return (a & m) | (b & ~m);
}

// 0x7F in every byte of x that is non-zero, 0x00 elsewhere.  Bytes must be <= 0x80 (exponents
// are <= 31), so the per-byte add never carries into its neighbour.
__device__ __forceinline__ uint32_t nz7(uint32_t x)
{
    uint32_t f = (x + K7F) & K80;
    return f - (f >> 7);
}

// 4x4 byte transpose: rows <-> columns.
__device__ __forceinline__ void transpose4x4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t &c0,
                                             uint32_t &c1, uint32_t &c2, uint32_t &c3)
{
    uint32_t t0 = byte_perm(r1, r0, 0x05010400u);  // r0[0] r1[0] r0[1] r1[1]
    uint32_t t1 = byte_perm(r1, r0, 0x07030602u);  // r0[2] r1[2] r0[3] r1[3]
    uint32_t t2 = byte_perm(r3, r2, 0x05010400u);
    uint32_t t3 = byte_perm(r3, r2, 0x07030602u);
    c0 = byte_perm(t2, t0, 0x05040100u);
    c1 = byte_perm(t2, t0, 0x07060302u);
    c2 = byte_perm(t3, t1, 0x05040100u);
    c3 = byte_perm(t3, t1, 0x07060302u);
}

// Slide + merge four independent lines at once.  Plane a holds, in each byte lane, the cell
// nearest the wall the tiles move to; d the farthest.  Returns the merge score (sum of merged
// tile VALUES over the four lines).
// [ref: twentyfortyeight/core/gameboard.py:23-69 merge_column, :72-102 slide_and_merge;
//       reinforce/game/core.py:32-101 _compact_row/merge_row]
__device__ __forceinline__ uint32_t slide_merge_planes(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
    uint32_t na = nz7(a), nb = nz7(b), nc = nz7(c), nd = nz7(d);

    // 1) stable compaction toward plane a ("column[column != 0]", gameboard.py:46): a bubble
    //    network of "pull the neighbour in if I am empty" steps; the non-zero masks ride along.
#define ENV2048_PULL(pi, pj, ni, nj)       \
    {                                      \
        uint32_t moved = bsel(ni, pi, pj); \
        pj &= ni;                          \
        pi = moved;                        \
        uint32_t any = ni | nj;            \
        nj &= ni;                          \
        ni = any;                          \
    }
    ENV2048_PULL(a, b, na, nb)
    ENV2048_PULL(b, c, nb, nc)
    ENV2048_PULL(c, d, nc, nd)
    ENV2048_PULL(a, b, na, nb)
    ENV2048_PULL(b, c, nb, nc)
    ENV2048_PULL(a, b, na, nb)
#undef ENV2048_PULL

    // 2) greedy pairwise merge from the wall outward, each tile merging at most once
    //    (gameboard.py:56-67).  After compaction "b non-zero" implies "a non-zero", etc.
    uint32_t m_ab = (nz7(a ^ b) ^ K7F) & nb;
    uint32_t m_bc = (nz7(b ^ c) ^ K7F) & nc & ~m_ab;
    uint32_t m_cd = (nz7(c ^ d) ^ K7F) & nd & ~m_bc;
    uint32_t x_ab = m_ab & K01, x_bc = m_bc & K01, x_cd = m_cd & K01;
    uint32_t a1 = a + x_ab;  // merged tile: value * 2 == exponent + 1
    uint32_t b1 = b + x_bc;
    uint32_t c1 = c + x_cd;
    uint32_t d1 = d & ~m_cd;
    uint32_t s2 = m_ab | m_bc;
    // merged exponents, for the score: at most two merges per line (ab+cd) or one (bc)
    uint32_t v1 = (a1 & m_ab) | (b1 & m_bc);
    uint32_t v2 = c1 & m_cd;
    // 3) close the hole a merge leaves behind
    a = a1;
    b = bsel(m_ab, c1, b1);
    c = bsel(s2, d1, c1);
    d = d1 & ~s2;

    // score = sum over merged cells of 2^exponent.  Summing 1 << byte over all eight bytes adds
    // a spurious 1 for every byte without a merge; take those back out.
    uint32_t n_merged = __popc(x_ab | x_bc) + __popc(x_cd);
    uint32_t s = n_merged - 8u;
    s += 1u << (v1 & 0xFFu);
    s += 1u << ((v1 >> 8) & 0xFFu);
    s += 1u << ((v1 >> 16) & 0xFFu);
    s += 1u << (v1 >> 24);
    s += 1u << (v2 & 0xFFu);
    s += 1u << ((v2 >> 8) & 0xFFu);
    s += 1u << ((v2 >> 16) & 0xFFu);
    s += 1u << (v2 >> 24);
    return s;
}

// Afterstate of one move, no spawn [ref: gameboard.py:105-129 latent_state; directions pinned by
// tests/test_jax_game.py:101-135: 0 left, 1 up, 2 right, 3 down].  The reference rotates the
// board so that every move becomes "left"; here up/down use the row dwords as planes directly and
// left/right use the transposed board, with the plane order reversed for right/down.
// `changed` == can_move(rot90(board, action)) [ref: twentyfortyeight/core/gamemove.py:132-164].
__device__ __forceinline__ void afterstate(Board &bd, uint32_t action, uint32_t &score, bool &changed)
{
    uint32_t t0, t1, t2, t3;
    transpose4x4(bd.r0, bd.r1, bd.r2, bd.r3, t0, t1, t2, t3);
    const bool horizontal = (action & 1u) == 0u;
    const bool reversed = (action & 2u) != 0u;
    uint32_t x0 = horizontal ? t0 : bd.r0;
    uint32_t x1 = horizontal ? t1 : bd.r1;
    uint32_t x2 = horizontal ? t2 : bd.r2;
    uint32_t x3 = horizontal ? t3 : bd.r3;
    uint32_t p0 = reversed ? x3 : x0;
    uint32_t p1 = reversed ? x2 : x1;
    uint32_t p2 = reversed ? x1 : x2;
    uint32_t p3 = reversed ? x0 : x3;
    uint32_t q0 = p0, q1 = p1, q2 = p2, q3 = p3;
    score = slide_merge_planes(q0, q1, q2, q3);
    changed = (((q0 ^ p0) | (q1 ^ p1)) | ((q2 ^ p2) | (q3 ^ p3))) != 0u;
    uint32_t y0 = reversed ? q3 : q0;
    uint32_t y1 = reversed ? q2 : q1;
    uint32_t y2 = reversed ? q1 : q2;
    uint32_t y3 = reversed ? q0 : q3;
    uint32_t u0, u1, u2, u3;
    transpose4x4(y0, y1, y2, y3, u0, u1, u2, u3);
    bd.r0 = horizontal ? u0 : y0;
    bd.r1 = horizontal ? u1 : y1;
    bd.r2 = horizontal ? u2 : y2;
    bd.r3 = horizontal ? u3 : y3;
}

// Bit (8*c + 7 - r) set <=> cell (r, c) is EMPTY.
__device__ __forceinline__ uint32_t empty_mask(const Board &bd)
{
    uint32_t y0 = bd.r0 + K7F, y1 = bd.r1 + K7F, y2 = bd.r2 + K7F, y3 = bd.r3 + K7F;
    uint32_t occ = (y0 & 0x80808080u) | ((y1 >> 1) & 0x40404040u) | ((y2 >> 2) & 0x20202020u) |
                   ((y3 >> 3) & 0x10101010u);
    return occ ^ 0xF0F0F0F0u;
}

// Keep only the k-th set bit of an empty_mask, counting cells in row-major order -- the order of
// argwhere(state == 0) [ref: gameboard.py:275,287; reinforce/game/core.py:206-213].  Four rounds
// of "is it in the first half of what is left?" (rows 0-1 | row 0/2 | columns 0-1 | column 0/2).
__device__ __forceinline__ uint32_t kth_empty(uint32_t m, uint32_t k)
{
#define ENV2048_HALVE(FIRST)                \
    {                                       \
        uint32_t lo = m & (FIRST);          \
        uint32_t cnt = __popc(lo);          \
        bool beyond = k >= cnt;             \
        m = beyond ? (m ^ lo) : lo;         \
        k = beyond ? k - cnt : k;           \
    }
    ENV2048_HALVE(0xC0C0C0C0u)
    ENV2048_HALVE(0xA0A0A0A0u)
    ENV2048_HALVE(0x0000FFFFu)
    ENV2048_HALVE(0x00FF00FFu)
#undef ENV2048_HALVE
    return m;
}

// OR (1 << shift) into the cell selected by a one-bit empty_mask (no-op for cell_bit == 0).
__device__ __forceinline__ void place_tile(Board &bd, uint32_t cell_bit, uint32_t shift)
{
    bd.r0 |= ((cell_bit >> 7) & K01) << shift;
    bd.r1 |= ((cell_bit >> 6) & K01) << shift;
    bd.r2 |= ((cell_bit >> 5) & K01) << shift;
    bd.r3 |= ((cell_bit >> 4) & K01) << shift;
}

// Flat row-major index (4*r + c) of the cell a one-bit empty_mask selects.
__device__ __forceinline__ uint32_t cell_index(uint32_t cell_bit)
{
    uint32_t pos = (uint32_t)__ffs((int)cell_bit) - 1u;  // 8*c + 7 - r
    return 4u * (7u - (pos & 7u)) + (pos >> 3);
}

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// Spawn one tile from two Philox words [ref: gameboard.py:247-288 fill_cells with number_tile=1;
// reinforce/game/core.py:173-231]: uniform over the empty cells, tile 4 with probability 0.1.
// `enable` false leaves the board untouched (invalid move: no spawn, gameboard.py:325).
__device__ __forceinline__ uint32_t spawn_tile(Board &bd, uint32_t w_cell, uint32_t w_val, bool enable)
{
    uint32_t empt = empty_mask(bd);
    uint32_t n_empty = __popc(empt);
    uint32_t k = mulhi32(w_cell, n_empty);
    uint32_t bit = kth_empty(empt, k);
    bit = enable ? bit : 0u;
    place_tile(bd, bit, (w_val < SPAWN4_THRESHOLD) ? 1u : 0u);
    return n_empty - (enable ? 1u : 0u);  // empties left on the board
}

// Fresh board: two sequential single spawns on an empty board, from one reset block
// [ref: twentyfortyeight/envs/twentyfortyeight.py:100-101; reinforce/game/core.py:339-342].
__device__ __forceinline__ Board fresh_board(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    uint32_t cell1 = w0 >> 28;            // floor(w0 * 16 / 2^32)
    uint32_t k2 = mulhi32(w2, 15u);       // rank among the 15 cells still empty
    uint32_t cell2 = k2 + (k2 >= cell1 ? 1u : 0u);
    uint32_t v1 = ((w1 < SPAWN4_THRESHOLD) ? 2u : 1u) << (8u * (cell1 & 3u));
    uint32_t v2 = ((w3 < SPAWN4_THRESHOLD) ? 2u : 1u) << (8u * (cell2 & 3u));
    uint32_t row1 = cell1 >> 2, row2 = cell2 >> 2;
    Board bd;
    bd.r0 = (row1 == 0u ? v1 : 0u) | (row2 == 0u ? v2 : 0u);
    bd.r1 = (row1 == 1u ? v1 : 0u) | (row2 == 1u ? v2 : 0u);
    bd.r2 = (row1 == 2u ? v1 : 0u) | (row2 == 2u ? v2 : 0u);
    bd.r3 = (row1 == 3u ? v1 : 0u) | (row2 == 3u ? v2 : 0u);
    return bd;
}

// True iff no two horizontally or vertically adjacent cells are equal.  Only meaningful as a
// terminal test together with "board is full" [ref: gameboard.py:346-348 is_done].
__device__ __forceinline__ bool no_equal_neighbours(const Board &bd)
{
    // bit 7 of a byte of (x + K7F) is set iff that byte of x is non-zero, i.e. the pair differs
    uint32_t h = ((bd.r0 ^ (bd.r0 >> 8)) + K7F) & ((bd.r1 ^ (bd.r1 >> 8)) + K7F) &
                 ((bd.r2 ^ (bd.r2 >> 8)) + K7F) & ((bd.r3 ^ (bd.r3 >> 8)) + K7F);
    uint32_t v = ((bd.r0 ^ bd.r1) + K7F) & ((bd.r1 ^ bd.r2) + K7F) & ((bd.r2 ^ bd.r3) + K7F);
    // byte 3 of h compares column 3 with nothing: ignore it
    return (((h | 0x80000000u) & v) & K80) == K80;
}

// Legal-action bitmask (bit a <=> action a changes the board) and the two "mergeable pair"
// facts, from adjacency alone [ref: twentyfortyeight/core/gamemove.py:45-83 legal_actions_mask].
__device__ __forceinline__ uint32_t legal_mask(const Board &bd)
{
    uint32_t f0 = (bd.r0 + K7F) & K80, f1 = (bd.r1 + K7F) & K80;  // non-zero flags
    uint32_t f2 = (bd.r2 + K7F) & K80, f3 = (bd.r3 + K7F) & K80;
    // horizontal pairs (c, c+1): byte c of (x >> 8) is cell c+1; byte 3 pairs with "nothing"
    uint32_t g0 = f0 >> 8, g1 = f1 >> 8, g2 = f2 >> 8, g3 = f3 >> 8;
    uint32_t slide_left = (~f0 & g0) | (~f1 & g1) | (~f2 & g2) | (~f3 & g3);              // left empty, right tile
    uint32_t slide_right = ((f0 & ~g0) | (f1 & ~g1) | (f2 & ~g2) | (f3 & ~g3)) & 0x00808080u;  // left tile, right empty
    uint32_t h_merge = (f0 & ~((bd.r0 ^ (bd.r0 >> 8)) + K7F)) | (f1 & ~((bd.r1 ^ (bd.r1 >> 8)) + K7F)) |
                       (f2 & ~((bd.r2 ^ (bd.r2 >> 8)) + K7F)) | (f3 & ~((bd.r3 ^ (bd.r3 >> 8)) + K7F));
    uint32_t slide_up = (~f0 & f1) | (~f1 & f2) | (~f2 & f3);    // top empty, bottom tile
    uint32_t slide_down = (f0 & ~f1) | (f1 & ~f2) | (f2 & ~f3);  // top tile, bottom empty
    uint32_t v_merge = (f0 & ~((bd.r0 ^ bd.r1) + K7F)) | (f1 & ~((bd.r1 ^ bd.r2) + K7F)) |
                       (f2 & ~((bd.r2 ^ bd.r3) + K7F));
    h_merge &= K80;
    v_merge &= K80;
    uint32_t legal = 0;
    legal |= ((slide_left | h_merge) != 0u) ? 1u : 0u;
    legal |= ((slide_up | v_merge) != 0u) ? 2u : 0u;
    legal |= ((slide_right | h_merge) != 0u) ? 4u : 0u;
    legal |= ((slide_down | v_merge) != 0u) ? 8u : 0u;
    return legal;
}

__device__ __forceinline__ uint32_t count_empty(const Board &bd) { return __popc(empty_mask(bd)); }

// Philox4x32-10 (Salmon et al. 2011).  Key schedule is wave-uniform and stays in SGPRs.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        uint32_t hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
        uint32_t hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    o0 = c0;
    o1 = c1;
    o2 = c2;
    o3 = c3;
}

// RNG contract (include/env2048.h): key = seed, counter = (env_lo, step_lo, step_hi,
// stream + 2*env_hi) with env the GLOBAL environment index.
struct RngKey {
    uint32_t seed_lo, seed_hi, step_lo, step_hi;
};

__device__ __forceinline__ void env_block(const RngKey &key, uint64_t env, uint32_t stream, uint32_t &o0,
                                          uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
    philox4x32_10((uint32_t)env, key.step_lo, key.step_hi, stream + 2u * (uint32_t)(env >> 32), key.seed_lo,
                  key.seed_hi, o0, o1, o2, o3);
}

// One full environment transition of one board [ref: gameboard.py:291-325 next_state +
// twentyfortyeight.py:132-133; reinforce/game/env.py:103-119].  w* = the step block.
__device__ __forceinline__ void transition(Board &bd, uint32_t action, uint32_t w_cell, uint32_t w_val,
                                           uint32_t &reward, bool &changed, bool &done)
{
    afterstate(bd, action & 3u, reward, changed);  // unchanged board => score 0 already
    uint32_t empties_left = spawn_tile(bd, w_cell, w_val, changed);
    done = (empties_left == 0u) && no_equal_neighbours(bd);
}

}  // namespace env2048
