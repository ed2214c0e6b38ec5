// nxn_kernels.hip -- boards of any size S x S, 2 <= S <= 8, and their C-ABI entry points (env2048_nxn_* of include/env2048.h).
//
// The reference's core functions take boards of any shape and its env has a `size` argument
// (twentyfortyeight/envs/twentyfortyeight.py:21; its own timing script runs 4 / 6 / 8, tests/test_perf_utils.py:47).  The hot
// path of this library is the packed 4 x 4 board (step_kernels.hip); this file is the general form behind the same API
// mirrors: one byte per cell, row-major, ONE BOARD PER LANE with the board held in LDS as cells[c][lane] (a lane's accesses
// to cell c hit consecutive bytes across the wave: four lanes per bank word, conflict-free), every line walked by plain
// loops.  A compatibility path sized for the reference's use (single boards, small batches), not a throughput path: for
// S = 4 it returns exactly what the packed kernels return (tests/test_nxn.py), about 20 x slower per board.
// Same RNG contract: the step word of (seed, env, step) picks the spawn, a fresh board takes two sequential spawns from the
// reset block (cell ranks over S*S and S*S - 1 empties).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"

using namespace env2048;
using namespace env2048::host;

namespace {

constexpr int NXN_BLOCK = 64;  // one wave per block: a lane's board is private, nothing is shared
constexpr int NXN_MAX = ENV2048_NXN_MAX_SIZE;
constexpr int NXN_MAX_CELLS = NXN_MAX * NXN_MAX;

struct Cells {  // the calling lane's board in LDS
    uint8_t *base;  // &cells[0][lane]
    uint32_t S;
    __device__ __forceinline__ uint32_t get(uint32_t c) const { return base[c * NXN_BLOCK]; }
    __device__ __forceinline__ void set(uint32_t c, uint32_t v) const { base[c * NXN_BLOCK] = (uint8_t)v; }
    __device__ __forceinline__ uint32_t count() const { return S * S; }
};

__device__ __forceinline__ void load_cells(const Cells &b, const uint8_t *boards, uint64_t i)
{
    const uint8_t *src = boards + i * b.count();
    for (uint32_t c = 0; c < b.count(); ++c) b.set(c, src[c]);
}

__device__ __forceinline__ void store_cells(const Cells &b, uint8_t *boards, uint64_t i)
{
    uint8_t *dst = boards + i * b.count();
    for (uint32_t c = 0; c < b.count(); ++c) dst[c] = (uint8_t)b.get(c);
}

// Cell k (counted from the wall the tiles move to) of line j of a move: 0 left, 1 up, 2 right, 3 down -- the directions
// rot90(k = action) + "slide left" + rot90 back give [ref: gameboard.py:127-129; pinned by tests/test_jax_game.py:101-135].
__device__ __forceinline__ uint32_t line_cell(uint32_t action, uint32_t j, uint32_t k, uint32_t S)
{
    const uint32_t kk = (action & 2u) ? S - 1u - k : k;
    return (action & 1u) ? kk * S + j : j * S + kk;
}

// Slide + merge every line of the board towards the wall of `action`, in place [ref: gameboard.py:23-69 merge_column,
// :72-102 slide_and_merge, :105-129 latent_state]: drop zeros, merge equal neighbours from the wall outward, each tile
// merging at most once; score = sum of the merged VALUES.  `changed` == can_move(rot90(board, action)) [gamemove.py:132-164].
__device__ __forceinline__ uint32_t nxn_move(const Cells &b, uint32_t action, bool &changed)
{
    const uint32_t S = b.S;
    uint32_t score = 0u;
    changed = false;
    for (uint32_t j = 0; j < S; ++j) {
        uint32_t v[NXN_MAX];
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NXN_MAX; ++k) v[k] = k < S ? b.get(line_cell(action, j, k, S)) : 0u;  // the whole line first:
        uint32_t w = 0u, last = 0u;  // w = cells written so far; last = the tile at w - 1 if it may still merge   (written in place below)
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)NXN_MAX; ++k) {
            const uint32_t x = v[k];
            if (x != 0u) {
                if (x == last) {  // merges into the tile before it; that tile is spent
                    b.set(line_cell(action, j, w - 1u, S), x + 1u);
                    score += 1u << ((x + 1u) & 31u);
                    last = 0u;
                    changed = true;
                } else {
                    changed = changed || k != w;  // it moves iff a gap (or a merge) lies between it and the wall
                    b.set(line_cell(action, j, w, S), x);
                    last = x;
                    ++w;
                }
            }
        }
        for (uint32_t k = w; k < S; ++k) b.set(line_cell(action, j, k, S), 0u);
    }
    return score;
}

__device__ __forceinline__ uint32_t nxn_count_empty(const Cells &b)
{
    uint32_t n = 0u;
    for (uint32_t c = 0; c < b.count(); ++c) n += b.get(c) == 0u ? 1u : 0u;
    return n;
}

// exponent e into the k-th empty cell in row-major order [ref: gameboard.py:275,287 argwhere order]; k out of range: nothing
__device__ __forceinline__ void nxn_place(const Cells &b, uint32_t k, uint32_t e)
{
    uint32_t seen = 0u;
    for (uint32_t c = 0; c < b.count(); ++c) {
        if (b.get(c) == 0u) {
            if (seen == k) {
                b.set(c, e);
                return;
            }
            ++seen;
        }
    }
}

// one spawn from a step word (RNG contract of include/env2048.h) [ref: gameboard.py:247-288 fill_cells, number_tile = 1]
__device__ __forceinline__ void nxn_spawn_word(const Cells &b, uint32_t word)
{
    const uint32_t n_empty = nxn_count_empty(b);
    if (n_empty == 0u) return;
    const uint64_t prod = (uint64_t)(word & ~3u) * (uint64_t)n_empty;
    nxn_place(b, (uint32_t)(prod >> 32), (uint32_t)prod < SPAWN4_THRESHOLD ? 2u : 1u);
}

// zeros + two sequential spawns from one reset block [ref: twentyfortyeight.py:100-101]
__device__ __forceinline__ void nxn_fresh(const Cells &b, const RngKey &key, uint64_t env)
{
    uint32_t w0, w1, w2, w3;
    env_block(key, env, STREAM_RESET, w0, w1, w2, w3);
    for (uint32_t c = 0; c < b.count(); ++c) b.set(c, 0u);
    nxn_place(b, mulhi32(w0, b.count()), w1 < SPAWN4_THRESHOLD ? 2u : 1u);
    nxn_place(b, mulhi32(w2, b.count() - 1u), w3 < SPAWN4_THRESHOLD ? 2u : 1u);
}

// [ref: gameboard.py:328-348 is_done] + [gamemove.py:45-83 legal_actions_mask] in one walk over the adjacent pairs
__device__ __forceinline__ void nxn_status(const Cells &b, bool &done, uint32_t &legal, uint32_t &n_empty, uint32_t &max_exp)
{
    const uint32_t S = b.S;
    bool h_merge = false, v_merge = false, left = false, right = false, up = false, down = false;
    n_empty = 0u;
    max_exp = 0u;
    for (uint32_t r = 0; r < S; ++r)
        for (uint32_t c = 0; c < S; ++c) {
            const uint32_t x = b.get(r * S + c);
            n_empty += x == 0u ? 1u : 0u;
            max_exp = max(max_exp, x);
            if (c + 1u < S) {
                const uint32_t y = b.get(r * S + c + 1u);
                h_merge = h_merge || (x != 0u && x == y);
                left = left || (x == 0u && y != 0u);
                right = right || (y == 0u && x != 0u);
            }
            if (r + 1u < S) {
                const uint32_t y = b.get((r + 1u) * S + c);
                v_merge = v_merge || (x != 0u && x == y);
                up = up || (x == 0u && y != 0u);
                down = down || (y == 0u && x != 0u);
            }
        }
    legal = ((left || h_merge) ? 1u : 0u) | ((up || v_merge) ? 2u : 0u) | ((right || h_merge) ? 4u : 0u) | ((down || v_merge) ? 8u : 0u);
    done = n_empty == 0u && !h_merge && !v_merge;
}

__device__ __forceinline__ uint32_t nxn_step_word(const RngKey &key, uint64_t env)
{
    uint32_t w[4];
    env_block(key, env >> 2, STREAM_STEP, w[0], w[1], w[2], w[3]);
    const uint32_t sel = (uint32_t)env & 3u;
    return sel == 0u ? w[0] : sel == 1u ? w[1] : sel == 2u ? w[2] : w[3];
}

#define NXN_PROLOGUE                                                        \
    __shared__ uint8_t cells_lds[NXN_MAX_CELLS][NXN_BLOCK];                 \
    const uint64_t i = (uint64_t)blockIdx.x * NXN_BLOCK + threadIdx.x;      \
    if (i >= n) return;                                                     \
    const Cells b{&cells_lds[0][threadIdx.x], size};

__global__ __launch_bounds__(NXN_BLOCK) void nxn_afterstate_kernel(uint64_t n, uint32_t size, const uint8_t *boards_in, const uint8_t *actions,
                                                                  uint8_t *boards_out, int32_t *reward, uint8_t *changed_out)
{
    NXN_PROLOGUE
    load_cells(b, boards_in, i);
    bool changed;
    const uint32_t score = nxn_move(b, actions[i] & 3u, changed);
    store_cells(b, boards_out, i);
    if (reward) reward[i] = (int32_t)score;
    if (changed_out) changed_out[i] = changed ? 1 : 0;
}

// [ref: gameboard.py:291-325 next_state + twentyfortyeight.py:106-133 step] one transition per board, in place
__global__ __launch_bounds__(NXN_BLOCK) void nxn_step_kernel(uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, const uint8_t *actions,
                                                            int32_t *reward, uint8_t *done_out, uint8_t *legal_out, RngKey key, uint32_t flags)
{
    NXN_PROLOGUE
    load_cells(b, boards, i);
    const uint64_t env = env_base + i;
    const uint32_t word = nxn_step_word(key, env);
    const uint32_t action = (flags & ENV2048_FLAG_RANDOM_ACTIONS) ? word & 3u : actions[i] & 3u;
    bool changed, done;
    const uint32_t score = nxn_move(b, action, changed);  // an invalid move leaves the board as it was and scores 0
    if (changed) nxn_spawn_word(b, word);
    uint32_t legal, n_empty, max_exp;
    nxn_status(b, done, legal, n_empty, max_exp);
    if (done && (flags & ENV2048_FLAG_AUTO_RESET)) {
        nxn_fresh(b, key, env);
        if (legal_out) {
            bool d2;
            nxn_status(b, d2, legal, n_empty, max_exp);
        }
    }
    store_cells(b, boards, i);
    if (reward) reward[i] = (int32_t)score;
    done_out[i] = done ? 1 : 0;
    if (legal_out) legal_out[i] = (uint8_t)legal;
}

__global__ __launch_bounds__(NXN_BLOCK) void nxn_status_kernel(uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *done_out, uint8_t *legal_out,
                                                              uint8_t *n_empty_out, uint8_t *max_exp_out)
{
    NXN_PROLOGUE
    load_cells(b, boards, i);
    bool done;
    uint32_t legal, n_empty, max_exp;
    nxn_status(b, done, legal, n_empty, max_exp);
    if (done_out) done_out[i] = done ? 1 : 0;
    if (legal_out) legal_out[i] = (uint8_t)legal;
    if (n_empty_out) n_empty_out[i] = (uint8_t)n_empty;
    if (max_exp_out) max_exp_out[i] = (uint8_t)max_exp;
}

__global__ __launch_bounds__(NXN_BLOCK) void nxn_reset_kernel(uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, uint8_t *done_out, RngKey key)
{
    NXN_PROLOGUE
    nxn_fresh(b, key, env_base + i);
    store_cells(b, boards, i);
    if (done_out) done_out[i] = 0;
}

__global__ __launch_bounds__(NXN_BLOCK) void nxn_spawn_random_kernel(uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, uint32_t seed_lo,
                                                                    uint32_t seed_hi, uint64_t step_index, uint32_t count)
{
    NXN_PROLOGUE
    load_cells(b, boards, i);
    for (uint32_t t = 0; t < count; ++t) {
        const uint64_t step = step_index + t;
        nxn_spawn_word(b, nxn_step_word(RngKey{seed_lo, seed_hi, (uint32_t)step, (uint32_t)(step >> 32)}, env_base + i));
    }
    store_cells(b, boards, i);
}

// [ref: gameboard.py:132-171 after_state] every (empty cell, value) outcome: 2 S S slots of S S bytes per board
__global__ __launch_bounds__(NXN_BLOCK) void nxn_chance_outcomes_kernel(uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *out, uint8_t *count_out)
{
    NXN_PROLOGUE
    load_cells(b, boards, i);
    const uint32_t cells = b.count();
    uint8_t *slots = out + i * 2u * cells * cells;
    for (uint32_t t = 0; t < 2u * cells * cells; ++t) slots[t] = 0;
    const uint32_t n_empty = nxn_count_empty(b);
    if (n_empty == 0u) {  // :156-157 a full board: itself
        for (uint32_t c = 0; c < cells; ++c) slots[c] = (uint8_t)b.get(c);
        count_out[i] = 1;
        return;
    }
    uint32_t k = 0u;
    for (uint32_t cell = 0; cell < cells; ++cell) {
        if (b.get(cell) != 0u) continue;
        for (uint32_t e = 1u; e <= 2u; ++e) {
            uint8_t *slot = slots + (size_t)(2u * k + (e - 1u)) * cells;
            for (uint32_t c = 0; c < cells; ++c) slot[c] = (uint8_t)(c == cell ? e : b.get(c));
        }
        ++k;
    }
    count_out[i] = (uint8_t)(2u * n_empty);
}

int check_nxn(const char *who, uint64_t n, uint32_t size)
{
    if (size < 2u || size > (uint32_t)NXN_MAX) return fail(ENV2048_ERR_INVALID_ARG, "%s: size must be in [2, %d], got %u", who, NXN_MAX, size);
    return check_count(n);
}

inline dim3 nxn_grid(uint64_t n) { return dim3((unsigned)((n + NXN_BLOCK - 1) / NXN_BLOCK)); }

}  // namespace

extern "C" {

int env2048_nxn_afterstate(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards_in, const uint8_t *actions,
                           uint8_t *boards_out, int32_t *reward, uint8_t *changed)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_afterstate", n, size)) return rc;
    if (!boards_in || !actions || !boards_out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_afterstate: boards_in, actions and boards_out are required");
    if (misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_nxn_afterstate: reward must be 4-byte aligned");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(nxn_afterstate_kernel, nxn_grid(n), dim3(NXN_BLOCK), 0, (hipStream_t)stream, n, size, boards_in, actions, boards_out, reward, changed);
    return launched("env2048_nxn_afterstate launch");
}

int env2048_nxn_step(int device, void *stream, uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, const uint8_t *actions,
                     int32_t *reward, uint8_t *done, uint8_t *legal, uint64_t seed, uint64_t step_index, uint32_t flags)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_step", n, size)) return rc;
    if (flags & ~(ENV2048_FLAG_AUTO_RESET | ENV2048_FLAG_RANDOM_ACTIONS)) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_step: unknown flag bits 0x%x", flags);
    if (!boards || !done) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_step: boards and done are required");
    if (!actions && !(flags & ENV2048_FLAG_RANDOM_ACTIONS)) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_step: actions is NULL without ENV2048_FLAG_RANDOM_ACTIONS");
    if (misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_nxn_step: reward must be 4-byte aligned");
    if (int rc = check_range(n, env_base)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)};
    hipLaunchKernelGGL(nxn_step_kernel, nxn_grid(n), dim3(NXN_BLOCK), 0, (hipStream_t)stream, n, size, env_base, boards, actions, reward, done, legal, key, flags);
    return launched("env2048_nxn_step launch");
}

int env2048_nxn_status(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *done, uint8_t *legal, uint8_t *n_empty,
                       uint8_t *max_exp)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_status", n, size)) return rc;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_status: boards is NULL");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(nxn_status_kernel, nxn_grid(n), dim3(NXN_BLOCK), 0, (hipStream_t)stream, n, size, boards, done, legal, n_empty, max_exp);
    return launched("env2048_nxn_status launch");
}

int env2048_nxn_reset(int device, void *stream, uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, uint8_t *done, uint64_t seed,
                      uint64_t step_index)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_reset", n, size)) return rc;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_reset: boards is NULL");
    if (int rc = check_range(n, env_base)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step_index, (uint32_t)(step_index >> 32)};
    hipLaunchKernelGGL(nxn_reset_kernel, nxn_grid(n), dim3(NXN_BLOCK), 0, (hipStream_t)stream, n, size, env_base, boards, done, key);
    return launched("env2048_nxn_reset launch");
}

int env2048_nxn_spawn_random(int device, void *stream, uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, uint64_t seed,
                             uint64_t step_index, uint32_t count)
{
    if (n == 0 || count == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_spawn_random", n, size)) return rc;
    if (!boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_spawn_random: boards is NULL");
    if (int rc = check_range(n, env_base)) return rc;
    if (step_index + count < step_index) return fail(ENV2048_ERR_RANGE, "env2048_nxn_spawn_random: step_index + count overflows");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(nxn_spawn_random_kernel, nxn_grid(n), dim3(NXN_BLOCK), 0, (hipStream_t)stream, n, size, env_base, boards, (uint32_t)seed,
                       (uint32_t)(seed >> 32), step_index, count);
    return launched("env2048_nxn_spawn_random launch");
}

int env2048_nxn_chance_outcomes(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *out_boards, uint8_t *out_count)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_chance_outcomes", n, size)) return rc;
    if (!boards || !out_boards || !out_count) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_chance_outcomes: NULL argument");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(nxn_chance_outcomes_kernel, nxn_grid(n), dim3(NXN_BLOCK), 0, (hipStream_t)stream, n, size, boards, out_boards, out_count);
    return launched("env2048_nxn_chance_outcomes launch");
}

}  // extern "C"
