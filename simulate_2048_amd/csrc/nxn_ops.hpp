// nxn_ops.hpp -- per-lane arithmetic for boards of any size S x S (2 <= S <= 8), one byte per cell, header-only device code.
//
// The general-size counterpart of board_ops.hpp, with the board size a template parameter so that every loop unrolls and the
// board lives in registers: row r of the board is G = ceil(S / 4) dwords, byte c & 3 of dword c >> 2 = column c (bytes past
// column S - 1 are zero).  All game logic is branch-free SWAR on those dwords, FOUR LINES PER DWORD as in the 4 x 4 path: a
// move is turned into "slide the S planes towards plane 0" (plane k = cell k of every line counted from the wall, one line
// per byte lane), where the rows of the board (up / down) or of its transpose (left / right) are the planes.
// Cell values are log2(tile) <= 30 (include/env2048.h); the SWAR tests below are exact for bytes <= 0x7F.
// Compiles for the host as well (ENV2048_HOST_EMULATION, tests/host_emu/emu_nxn.cpp): checked there against the oracle for
// every size without a GPU.
//
// Semantics follow the reference NumPy engine; the lines each routine restates are cited as [ref: file:lines].
#pragma once

#include "board_ops.hpp"

namespace env2048 {
namespace nxn {

#if defined(ENV2048_HOST_EMULATION)
#define ENV2048_ALIGNBYTE(hi, lo, sh) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint64_t)(lo)) >> (8u * ((sh) & 3u))))
#else
#define ENV2048_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte(hi, lo, sh)
#endif

// low `bytes` bytes set (bytes <= 0: none, >= 4: all)
constexpr uint32_t low_bytes(int bytes) { return bytes <= 0 ? 0u : bytes >= 4 ? 0xFFFFFFFFu : (1u << (8 * bytes)) - 1u; }

template <int S>
struct Shape {
    static_assert(S >= 2 && S <= 8, "board sizes 2 ... 8");
    static constexpr int CELLS = S * S;
    static constexpr int G = (S + 3) / 4;            // dwords per row of the board == dwords per plane of a move
    static constexpr int DENSE = (CELLS + 3) / 4;    // dwords of the board as it lies in memory (row-major bytes)
    // bytes of dword g of a row (or plane) that belong to the board: columns (lines) 4g ... 4g + 3 below S
    static constexpr uint32_t valid(int g) { return low_bytes(S - 4 * g); }
    // bytes of dword g whose RIGHT neighbour (column + 1) also belongs to the board
    static constexpr uint32_t pair_valid(int g) { return low_bytes(S - 1 - 4 * g); }
};

// The board of the calling lane.
template <int S>
struct Rows {
    uint32_t w[S][Shape<S>::G];
};

// ---- memory image <-> rows ---------------------------------------------------------------------------------------------------
// dense[i] = bytes 4i ... 4i + 3 of the S*S row-major cell bytes (zero past the end).
template <int S>
ENV2048_DEV void rows_from_dense(const uint32_t (&dense)[Shape<S>::DENSE], Rows<S> &b)
{
    using Sh = Shape<S>;
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int g = 0; g < Sh::G; ++g) {
            const int o = r * S + 4 * g;  // first byte of this row dword in the dense image
            const int i = o >> 2, sh = o & 3;
            const uint32_t lo = dense[i];
            const uint32_t hi = i + 1 < Sh::DENSE ? dense[i + 1] : 0u;
            const uint32_t word = sh == 0 ? lo : ENV2048_ALIGNBYTE(hi, lo, (uint32_t)sh);
            b.w[r][g] = word & Sh::valid(g);
        }
}

template <int S>
ENV2048_DEV void dense_from_rows(const Rows<S> &b, uint32_t (&dense)[Shape<S>::DENSE])
{
    using Sh = Shape<S>;
#pragma unroll
    for (int i = 0; i < Sh::DENSE; ++i) dense[i] = 0u;
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int g = 0; g < Sh::G; ++g) {
            const int o = r * S + 4 * g;
            const int i = o >> 2, sh = o & 3;
            dense[i] |= b.w[r][g] << (8 * sh);  // (padding bytes of a row dword are zero: OR is enough)
            if (sh != 0 && i + 1 < Sh::DENSE) dense[i + 1] |= b.w[r][g] >> (32 - 8 * sh);
        }
}

// ---- transpose ---------------------------------------------------------------------------------------------------------------
// t.w[c][r >> 2] byte r & 3 = b.w[r][c >> 2] byte c & 3, by 4 x 4 byte blocks (8 v_perm_b32 each).
template <int S>
ENV2048_DEV void transpose(const Rows<S> &b, Rows<S> &t)
{
    using Sh = Shape<S>;
#pragma unroll
    for (int br = 0; br < Sh::G; ++br)
#pragma unroll
        for (int bc = 0; bc < Sh::G; ++bc) {
            const uint32_t r0 = b.w[4 * br][bc];
            const uint32_t r1 = 4 * br + 1 < S ? b.w[4 * br + 1 < S ? 4 * br + 1 : 0][bc] : 0u;
            const uint32_t r2 = 4 * br + 2 < S ? b.w[4 * br + 2 < S ? 4 * br + 2 : 0][bc] : 0u;
            const uint32_t r3 = 4 * br + 3 < S ? b.w[4 * br + 3 < S ? 4 * br + 3 : 0][bc] : 0u;
            uint32_t c0, c1, c2, c3;
            transpose4x4(r0, r1, r2, r3, c0, c1, c2, c3);
            t.w[4 * bc][br] = c0;
            if (4 * bc + 1 < S) t.w[4 * bc + 1 < S ? 4 * bc + 1 : 0][br] = c1;
            if (4 * bc + 2 < S) t.w[4 * bc + 2 < S ? 4 * bc + 2 : 0][br] = c2;
            if (4 * bc + 3 < S) t.w[4 * bc + 3 < S ? 4 * bc + 3 : 0][br] = c3;
        }
}

// ---- slide + merge -----------------------------------------------------------------------------------------------------------
// Four independent lines of S cells at once: byte lane j of p[k] = cell k of line j, k counted from the wall the tiles move
// to.  One pass over the cells in that order [ref: gameboard.py:23-69 merge_column: drop the zeros, merge equal neighbours from
// the wall outward, each tile merging at most once], with the write position of every line kept ONE-HOT over the output planes
// (at[t] = 0xFF in the byte lanes whose next tile lands in cell t) so that "append to the line" is a masked OR into every
// plane and never an indexed access.  Returns the merge score of the four lines (sum of the merged tile VALUES) and ORs the
// bytes that changed into `changed`.
template <int S>
ENV2048_DEV uint32_t slide_merge_lines(uint32_t (&p)[S], uint32_t &changed)
{
    uint32_t out[S], merged[S], at[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
        out[t] = 0u;
        merged[t] = 0u;
        at[t] = t == 0 ? 0xFFFFFFFFu : 0u;
    }
    uint32_t last = 0u;  // the tile before the write position, if it may still merge
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint32_t x = p[k];
        const uint32_t nz = nzmask(x);
        // merges into the tile before it (equal, non-zero); that tile is spent
        const uint32_t m = k == 0 ? 0u : ENV2048_BITOP3(nzmask(x ^ last), nz, nz, 0x0C);  // ~differs & nz
        const uint32_t placed = ENV2048_BITOP3(x, m, m, 0x30);                              // x & ~m
        const uint32_t advance = ENV2048_BITOP3(nz, m, m, 0x30);                            // nz & ~m
#pragma unroll
        for (int t = 1; t <= k; ++t) merged[t - 1] = ENV2048_BITOP3(merged[t - 1], m, at[t], 0xF8);  // | (m & at[t])
#pragma unroll
        for (int t = 0; t <= k; ++t) out[t] = ENV2048_BITOP3(out[t], placed, at[t], 0xF8);            // | (placed & at[t])
        last = ENV2048_BITOP3(placed, last, nz, 0xF4);  // placed | (last & ~nz): a zero cell changes nothing, a merge clears it
        if (k + 1 < S) {
#pragma unroll
            for (int t = k + 1; t >= 1; --t) at[t] = ENV2048_BFI(advance, at[t - 1], at[t]);
            at[0] = ENV2048_BITOP3(at[0], advance, advance, 0x30);
        }
    }
    uint32_t score = 0u;
#pragma unroll
    for (int t = 0; t + 1 < S; ++t) {
        const uint32_t flag = merged[t] & K01;
        out[t] += flag;  // merged tile: value * 2 == exponent + 1
        score += (shl_byte<0>(flag, out[t]) + shl_byte<1>(flag, out[t])) + (shl_byte<2>(flag, out[t]) + shl_byte<3>(flag, out[t]));
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        changed |= out[k] ^ p[k];
        p[k] = out[k];
    }
    return score;
}

// Afterstate of one move, no spawn [ref: gameboard.py:105-129 latent_state; directions pinned by tests/test_jax_game.py:101-135:
// 0 left, 1 up, 2 right, 3 down].  up / down slide along the columns: the rows of the board are the planes (in reverse order
// for down); left / right use the rows of the transpose.  `changed` == can_move(rot90(board, action)) [ref: gamemove.py:132-164].
template <int S>
ENV2048_DEV void afterstate(Rows<S> &b, uint32_t action, uint32_t &score, bool &changed)
{
    using Sh = Shape<S>;
    const bool vertical = (action & 1u) != 0u, reversed = (action & 2u) != 0u;
    Rows<S> t;
    transpose<S>(b, t);
    uint32_t diff = 0u;
    score = 0u;
    Rows<S> moved;
#pragma unroll
    for (int g = 0; g < Sh::G; ++g) {
        uint32_t p[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const uint32_t fwd = vertical ? b.w[k][g] : t.w[k][g];
            const uint32_t bwd = vertical ? b.w[S - 1 - k][g] : t.w[S - 1 - k][g];
            p[k] = reversed ? bwd : fwd;
        }
        score += slide_merge_lines<S>(p, diff);
#pragma unroll
        for (int k = 0; k < S; ++k) moved.w[k][g] = reversed ? p[S - 1 - k] : p[k];
    }
    changed = diff != 0u;
    Rows<S> back;
    transpose<S>(moved, back);
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int g = 0; g < Sh::G; ++g) b.w[r][g] = vertical ? moved.w[r][g] : back.w[r][g];
}

// ---- empty cells, spawn ------------------------------------------------------------------------------------------------------
// bit 7 of every byte of the board that holds an empty cell
template <int S>
ENV2048_DEV uint32_t empty_bits(uint32_t word, int g)
{
    return ENV2048_BITOP3(word + K7F, K80 & Shape<S>::valid(g), 0u, 0x0C);  // ~(word + 0x7F..) & 0x80.. & valid
}

template <int S>
ENV2048_DEV uint32_t count_empty(const Rows<S> &b)
{
    uint32_t n = 0u;
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int g = 0; g < Shape<S>::G; ++g) n += ENV2048_POPC(empty_bits<S>(b.w[r][g], g));
    return n;
}

// Exponent e into the k-th empty cell in row-major order [ref: gameboard.py:275,287 argwhere order]; k >= the number of empty
// cells or `enable` false: nothing.  Returns the number of cells that were empty before.
template <int S>
ENV2048_DEV uint32_t place_kth(Rows<S> &b, uint32_t k, uint32_t e, bool enable)
{
    using Sh = Shape<S>;
    uint32_t n_empty = 0u, in_row = 0u, row = 0xFFFFFFFFu, lo = 0u, hi = 0u;
#pragma unroll
    for (int r = 0; r < S; ++r) {  // the row the k-th empty cell lies in, its empty-cell bits and the rank within it
        const uint32_t e0 = empty_bits<S>(b.w[r][0], 0);
        const uint32_t e1 = Sh::G > 1 ? empty_bits<S>(b.w[r][Sh::G - 1], Sh::G - 1) : 0u;
        const uint32_t cnt = ENV2048_POPC(e0) + ENV2048_POPC(e1);
        const bool here = k >= n_empty && k - n_empty < cnt;
        in_row = here ? k - n_empty : in_row;
        row = here ? (uint32_t)r : row;
        lo = here ? e0 : lo;
        hi = here ? e1 : hi;
        n_empty += cnt;
    }
    // the in_row-th set bit of (lo, hi): which dword, which half, which byte
    const uint32_t c_lo = ENV2048_POPC(lo);
    const bool upper = in_row >= c_lo;
    uint32_t bits = upper ? hi : lo;
    uint32_t rank = upper ? in_row - c_lo : in_row;
    const uint32_t c_half = ENV2048_POPC(bits & 0x00008080u);
    const bool second = rank >= c_half;
    bits = second ? bits >> 16 : bits;
    rank = second ? rank - c_half : rank;
    const uint32_t odd = rank >= ((bits >> 7) & 1u) ? 1u : 0u;  // rank 0 and byte 0 empty -> byte 0; else byte 1
    const uint32_t byte = (second ? 2u : 0u) + odd;
    const uint32_t tile = (enable && row != 0xFFFFFFFFu) ? e << (8u * byte) : 0u;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        const uint32_t mine = row == (uint32_t)r ? tile : 0u;
        if (Sh::G > 1) {
            b.w[r][0] |= upper ? 0u : mine;
            b.w[r][Sh::G - 1] |= upper ? mine : 0u;
        } else {
            b.w[r][0] |= mine;
        }
    }
    return n_empty;
}

// One spawn from a step word (RNG contract of include/env2048.h) [ref: gameboard.py:247-288 fill_cells, number_tile = 1]:
// with m = word & ~3 and P = m * n_empty, the cell rank is P >> 32 and the tile is 4 iff (P mod 2^32) < SPAWN4_THRESHOLD.
// Returns the number of empty cells left afterwards.
template <int S>
ENV2048_DEV uint32_t spawn_word(Rows<S> &b, uint32_t word, bool enable)
{
    const uint32_t n_empty = count_empty<S>(b);
    const uint64_t prod = (uint64_t)(word & ~3u) * (uint64_t)n_empty;
    const bool spawn = enable && n_empty != 0u;
    place_kth<S>(b, (uint32_t)(prod >> 32), (uint32_t)prod < SPAWN4_THRESHOLD ? 2u : 1u, spawn);
    return n_empty - (spawn ? 1u : 0u);
}

// Fresh board: zeros + two sequential spawns from one reset block [ref: twentyfortyeight/envs/twentyfortyeight.py:100-101]:
// cell ranks floor(w0 * S*S / 2^32) over all cells and floor(w2 * (S*S - 1) / 2^32) over the cells left.
template <int S>
ENV2048_DEV void fresh(Rows<S> &b, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    using Sh = Shape<S>;
    const uint32_t c0 = ENV2048_UMULHI(w0, (uint32_t)Sh::CELLS);
    const uint32_t r1 = ENV2048_UMULHI(w2, (uint32_t)Sh::CELLS - 1u);
    const uint32_t c1 = r1 + (r1 >= c0 ? 1u : 0u);
    const uint32_t e0 = w1 < SPAWN4_THRESHOLD ? 2u : 1u, e1 = w3 < SPAWN4_THRESHOLD ? 2u : 1u;
    const uint32_t row0 = c0 / (uint32_t)S, col0 = c0 % (uint32_t)S, row1 = c1 / (uint32_t)S, col1 = c1 % (uint32_t)S;
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int g = 0; g < Sh::G; ++g) {
            const uint32_t a = (row0 == (uint32_t)r && (col0 >> 2) == (uint32_t)g) ? e0 << (8u * (col0 & 3u)) : 0u;
            const uint32_t c = (row1 == (uint32_t)r && (col1 >> 2) == (uint32_t)g) ? e1 << (8u * (col1 & 3u)) : 0u;
            b.w[r][g] = a | c;
        }
}

// ---- terminal test, legal moves ----------------------------------------------------------------------------------------------
// [ref: gameboard.py:328-348 is_done] + [gamemove.py:45-83 legal_actions_mask] from adjacency alone: a move is legal iff, along
// its axis, an empty cell sits directly before an occupied one (in the direction of the move) or two equal tiles are adjacent.
// Only bit 7 of every byte is meaningful in the intermediate words.
// COUNT false: `n_empty` comes in (the caller knows it, e.g. from the spawn that has just filled a cell) and is not recounted.
template <int S, bool COUNT = true>
ENV2048_DEV void status(const Rows<S> &b, bool &done, uint32_t &legal, uint32_t &n_empty)
{
    using Sh = Shape<S>;
    uint32_t h_merge = 0u, left = 0u, right = 0u, v_merge = 0u, up = 0u, down = 0u;
    if (COUNT) n_empty = 0u;
#pragma unroll
    for (int g = 0; g < Sh::G; ++g) {
        uint32_t hm = 0u, lf = 0u, rt = 0u, vm = 0u, u = 0u, d = 0u;
#pragma unroll
        for (int r = 0; r < S; ++r) {
            const uint32_t x = b.w[r][g];
            const uint32_t nx = x + K7F;  // bit 7: occupied
            if (COUNT) n_empty += ENV2048_POPC(ENV2048_BITOP3(nx, K80 & Sh::valid(g), 0u, 0x0C));
            if (Sh::pair_valid(g) != 0u) {  // the cell to the right of every cell of this dword
                const uint32_t next = g + 1 < Sh::G ? b.w[r][g + 1 < Sh::G ? g + 1 : g] : 0u;
                const uint32_t y = ENV2048_ALIGNBYTE(next, x, 1u);
                const uint32_t ny = y + K7F;
                const uint32_t differ = xor_add(x, y, K7F);
                hm = ENV2048_BITOP3(hm, differ, ny, 0xF2);  // | (~differ & ny)
                lf = ENV2048_BITOP3(lf, nx, ny, 0xF2);      // | (~nx & ny): empty, then occupied -> a move left slides
                rt = ENV2048_BITOP3(rt, ny, nx, 0xF2);      // | (~ny & nx)
            }
            if (r + 1 < S) {  // the cell below
                const uint32_t y = b.w[r + 1 < S ? r + 1 : r][g];
                const uint32_t ny = y + K7F;
                const uint32_t differ = xor_add(x, y, K7F);
                vm = ENV2048_BITOP3(vm, differ, ny, 0xF2);
                u = ENV2048_BITOP3(u, nx, ny, 0xF2);
                d = ENV2048_BITOP3(d, ny, nx, 0xF2);
            }
        }
        const uint32_t pairs = K80 & Sh::pair_valid(g), cells = K80 & Sh::valid(g);
        h_merge |= hm & pairs;
        left |= lf & pairs;
        right |= rt & pairs;
        v_merge |= vm & cells;
        up |= u & cells;
        down |= d & cells;
    }
    legal = ((left | h_merge) != 0u ? 1u : 0u) | ((up | v_merge) != 0u ? 2u : 0u) | ((right | h_merge) != 0u ? 4u : 0u) |
            ((down | v_merge) != 0u ? 8u : 0u);
    done = n_empty == 0u && (h_merge | v_merge) == 0u;
}

// Largest exponent on the board = log2(max tile).
template <int S>
ENV2048_DEV uint32_t max_exponent(const Rows<S> &b)
{
    uint32_t m = 0u;
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int g = 0; g < Shape<S>::G; ++g) {
            const uint32_t x = b.w[r][g];
            const uint32_t m01 = (x & 0xFFu) > ((x >> 8) & 0xFFu) ? (x & 0xFFu) : ((x >> 8) & 0xFFu);
            const uint32_t m23 = ((x >> 16) & 0xFFu) > (x >> 24) ? ((x >> 16) & 0xFFu) : (x >> 24);
            const uint32_t m4 = m01 > m23 ? m01 : m23;
            m = m > m4 ? m : m4;
        }
    return m;
}

// The step word of environment `env` (RNG contract of include/env2048.h: word env & 3 of the step block of env >> 2).
ENV2048_DEV uint32_t step_word(const RngKey &key, uint64_t env)
{
    uint32_t w0, w1, w2, w3;
    env_block(key, env >> 2, STREAM_STEP, w0, w1, w2, w3);
    const uint32_t sel = (uint32_t)env & 3u;
    return sel == 0u ? w0 : sel == 1u ? w1 : sel == 2u ? w2 : w3;
}

// One full transition [ref: gameboard.py:291-325 next_state + twentyfortyeight.py:106-133 step]: slide + merge (the move drawn
// from the step word with `random_action`), one spawn from the step word iff the board changed, terminal test; a finished game
// restarts from the reset block with `auto_reset` (`done` still reports the game that ended; `legal` describes the board left
// behind).
template <int S>
ENV2048_DEV void step_board(Rows<S> &b, const RngKey &key, uint64_t env, uint32_t action, bool random_action, bool auto_reset,
                            uint32_t &reward, bool &done, uint32_t &legal)
{
    const uint32_t word = step_word(key, env);
    bool changed;
    afterstate<S>(b, random_action ? word & 3u : action & 3u, reward, changed);  // an invalid move leaves the board as it was and scores 0
    uint32_t n_empty = spawn_word<S>(b, word, changed);  // (the empty cells left after the spawn)
    status<S, false>(b, done, legal, n_empty);
    if (done && auto_reset) {
        uint32_t w0, w1, w2, w3;
        env_block(key, env, STREAM_RESET, w0, w1, w2, w3);
        fresh<S>(b, w0, w1, w2, w3);
        bool fresh_done;
        status<S>(b, fresh_done, legal, n_empty);
    }
}

}  // namespace nxn
}  // namespace env2048
