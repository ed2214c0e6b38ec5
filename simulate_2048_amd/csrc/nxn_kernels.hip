// nxn_kernels.hip -- boards of any size S x S, 2 <= S <= 8, and their C-ABI entry points (env2048_nxn_* of include/env2048.h).
//
// The reference's core functions take boards of any shape and its env has a `size` argument
// (twentyfortyeight/envs/twentyfortyeight.py:21; its own timing script runs 4 / 6 / 8, tests/test_perf_utils.py:47).  The hot
// path of this library is the packed 4 x 4 board (step_kernels.hip); this file is the general form behind the same API
// mirrors: one byte per cell, row-major, ONE BOARD PER LANE, the board size a template parameter (seven instantiations of
// every kernel, chosen on the host), the board in registers and all game logic branch-free SWAR with four lines per dword
// (nxn_ops.hpp).  Memory traffic goes through a wave-private LDS strip: the 64 boards of a wave are 64 S S CONTIGUOUS bytes,
// fetched and written back as whole 16-byte vectors by the wave together (coalesced whatever S is) and picked up from /
// dropped into the strip by each lane in between (lane stride padded to an odd number of dwords for the even sizes; funnel
// shifts for the odd ones, whose boards do not start on dword boundaries).  A wave at the ragged end of the batch, or a batch
// that does not start on a 16-byte boundary, takes a per-lane byte path instead.
// For S = 4 every entry point returns exactly what the packed kernels return (tests/test_nxn.py).
// Same RNG contract: the step word of (seed, env, step) picks the spawn, a fresh board takes two sequential spawns from the
// reset block (cell ranks over S*S and S*S - 1 empties).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "nxn_ops.hpp"
#include "board_io.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"

using namespace env2048;
using namespace env2048::host;

namespace {

constexpr int NXN_BLOCK = 256;
constexpr int NXN_WAVES = NXN_BLOCK / 64;
constexpr int NXN_MAX = ENV2048_NXN_MAX_SIZE;

// ---- the wave's boards: global memory <-> LDS strip <-> registers -------------------------------------------------------------
template <int S>
struct Strip {
    using Sh = nxn::Shape<S>;
    static constexpr bool WHOLE = Sh::CELLS % 4 == 0;  // boards start on dword boundaries (even S)
    static constexpr int D = Sh::DENSE;
    // lane stride in the strip, in dwords: odd (or 1), so that the lanes' reads of dword i fall into different banks
    static constexpr int STRIDE = (WHOLE && D > 1 && D % 2 == 0) ? D + 1 : D;
    static constexpr bool PADDED = WHOLE && STRIDE != D;
    static constexpr int VECTORS = 4 * Sh::CELLS;  // 16-byte vectors in the 64 boards of a wave
    static constexpr int ROUNDS = (VECTORS + 63) / 64;
    static constexpr int DWORDS = (WHOLE ? 64 * STRIDE : 16 * Sh::CELLS) + 4;  // (+ slack: the funnel reads one dword ahead)
};

__device__ __forceinline__ void strip_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the strip is wave-private: LDS operations of a wave are ordered
    __builtin_amdgcn_wave_barrier();
}

// boards [first, first + 64) of the batch -> `b` of every lane (lanes past the end of the batch: an empty board)
template <int S>
__device__ __forceinline__ void load_board(uint32_t *strip, const uint8_t *boards, uint64_t first, uint64_t n, uint32_t lane, nxn::Rows<S> &b)
{
    using St = Strip<S>;
    const uint8_t *src = boards + first * St::Sh::CELLS;
    uint32_t dense[St::D];
    if (n - first >= 64u && ((uintptr_t)src & 15u) == 0u) {  // (wave-uniform)
        const uint4 *vec = reinterpret_cast<const uint4 *>(src);
#pragma unroll
        for (int round = 0; round < St::ROUNDS; ++round) {
            const uint32_t q = lane + 64u * round;
            if (q < (uint32_t)St::VECTORS) {
                const uint4 v = vec[q];
                if (St::PADDED) {
                    const uint32_t at = (4u * q / St::D) * St::STRIDE + 4u * q % St::D;  // D is 4 or 16 here: a vector never straddles boards
                    strip[at] = v.x;
                    strip[at + 1] = v.y;
                    strip[at + 2] = v.z;
                    strip[at + 3] = v.w;
                } else {
                    reinterpret_cast<uint4 *>(strip)[q] = v;
                }
            }
        }
        strip_fence();
        if (St::WHOLE) {
#pragma unroll
            for (int i = 0; i < St::D; ++i) dense[i] = strip[lane * St::STRIDE + i];
        } else {
            const uint32_t byte = lane * St::Sh::CELLS, base = byte >> 2, shift = byte & 3u;
            uint32_t raw[St::D + 1];
#pragma unroll
            for (int i = 0; i <= St::D; ++i) raw[i] = strip[base + i];
#pragma unroll
            for (int i = 0; i < St::D; ++i) dense[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], shift);
            dense[St::D - 1] &= nxn::low_bytes(St::Sh::CELLS - 4 * (St::D - 1));
        }
        strip_fence();
    } else {
#pragma unroll
        for (int i = 0; i < St::D; ++i) dense[i] = 0u;
        if (first + lane < n) {
            const uint8_t *mine = src + (size_t)lane * St::Sh::CELLS;
#pragma unroll
            for (int c = 0; c < St::Sh::CELLS; ++c) dense[c >> 2] |= (uint32_t)mine[c] << (8 * (c & 3));
        }
    }
    nxn::rows_from_dense<S>(dense, b);
}

template <int S>
__device__ __forceinline__ void store_board(uint32_t *strip, uint8_t *boards, uint64_t first, uint64_t n, uint32_t lane, const nxn::Rows<S> &b)
{
    using St = Strip<S>;
    uint8_t *dst = boards + first * St::Sh::CELLS;
    uint32_t dense[St::D];
    nxn::dense_from_rows<S>(b, dense);
    if (n - first >= 64u && ((uintptr_t)dst & 15u) == 0u) {
        if (St::WHOLE) {
#pragma unroll
            for (int i = 0; i < St::D; ++i) strip[lane * St::STRIDE + i] = dense[i];
        } else {  // neighbouring lanes share dwords: byte stores
            uint8_t *bytes = reinterpret_cast<uint8_t *>(strip) + lane * St::Sh::CELLS;
#pragma unroll
            for (int c = 0; c < St::Sh::CELLS; ++c) bytes[c] = (uint8_t)(dense[c >> 2] >> (8 * (c & 3)));
        }
        strip_fence();
        uint4 *vec = reinterpret_cast<uint4 *>(dst);
#pragma unroll
        for (int round = 0; round < St::ROUNDS; ++round) {
            const uint32_t q = lane + 64u * round;
            if (q < (uint32_t)St::VECTORS) {
                uint4 v;
                if (St::PADDED) {
                    const uint32_t at = (4u * q / St::D) * St::STRIDE + 4u * q % St::D;
                    v = make_uint4(strip[at], strip[at + 1], strip[at + 2], strip[at + 3]);
                } else {
                    v = reinterpret_cast<const uint4 *>(strip)[q];
                }
                store_x4_through(vec + q, u32x4{v.x, v.y, v.z, v.w});  // (write-through, like every wide output)
            }
        }
        strip_fence();
    } else if (first + lane < n) {
        uint8_t *mine = dst + (size_t)lane * St::Sh::CELLS;
#pragma unroll
        for (int c = 0; c < St::Sh::CELLS; ++c) mine[c] = (uint8_t)(dense[c >> 2] >> (8 * (c & 3)));
    }
}

#define NXN_PROLOGUE(S)                                                                      \
    __shared__ __attribute__((aligned(16))) uint32_t strips[NXN_WAVES][Strip<S>::DWORDS];    \
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;                        \
    const uint64_t first = ((uint64_t)blockIdx.x * NXN_WAVES + wave) * 64u;                  \
    if (first >= n) return; /* (a whole wave: there is no block-wide barrier below) */       \
    uint32_t *const strip = strips[wave];                                                    \
    const uint64_t i = first + lane;                                                         \
    const bool active = i < n;

template <int S>
__global__ __launch_bounds__(NXN_BLOCK) void nxn_afterstate_kernel(uint64_t n, const uint8_t *boards_in, const uint8_t *actions, uint8_t *boards_out,
                                                                  int32_t *reward, uint8_t *changed_out)
{
    NXN_PROLOGUE(S)
    nxn::Rows<S> b;
    load_board<S>(strip, boards_in, first, n, lane, b);
    uint32_t score;
    bool changed;
    nxn::afterstate<S>(b, active ? actions[i] & 3u : 0u, score, changed);
    store_board<S>(strip, boards_out, first, n, lane, b);
    if (!active) return;
    if (reward) reward[i] = (int32_t)score;
    if (changed_out) changed_out[i] = changed ? 1 : 0;
}

// [ref: gameboard.py:291-325 next_state + twentyfortyeight.py:106-133 step] one transition per board, in place
template <int S>
__global__ __launch_bounds__(NXN_BLOCK) void nxn_step_kernel(uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions, int32_t *reward,
                                                            uint8_t *done_out, uint8_t *legal_out, RngKey key, uint32_t flags)
{
    NXN_PROLOGUE(S)
    nxn::Rows<S> b;
    load_board<S>(strip, boards, first, n, lane, b);
    const bool random_action = (flags & ENV2048_FLAG_RANDOM_ACTIONS) != 0u;
    uint32_t score, legal;
    bool done;
    nxn::step_board<S>(b, key, env_base + i, (active && !random_action) ? actions[i] : 0u, random_action, (flags & ENV2048_FLAG_AUTO_RESET) != 0u,
                       score, done, legal);
    store_board<S>(strip, boards, first, n, lane, b);
    if (!active) return;
    if (reward) reward[i] = (int32_t)score;
    done_out[i] = done ? 1 : 0;
    if (legal_out) legal_out[i] = (uint8_t)legal;
}

template <int S>
__global__ __launch_bounds__(NXN_BLOCK) void nxn_status_kernel(uint64_t n, const uint8_t *boards, uint8_t *done_out, uint8_t *legal_out, uint8_t *n_empty_out,
                                                              uint8_t *max_exp_out)
{
    NXN_PROLOGUE(S)
    nxn::Rows<S> b;
    load_board<S>(strip, boards, first, n, lane, b);
    bool done;
    uint32_t legal, n_empty;
    nxn::status<S>(b, done, legal, n_empty);
    if (!active) return;
    if (done_out) done_out[i] = done ? 1 : 0;
    if (legal_out) legal_out[i] = (uint8_t)legal;
    if (n_empty_out) n_empty_out[i] = (uint8_t)n_empty;
    if (max_exp_out) max_exp_out[i] = (uint8_t)nxn::max_exponent<S>(b);
}

template <int S>
__global__ __launch_bounds__(NXN_BLOCK) void nxn_reset_kernel(uint64_t n, uint64_t env_base, uint8_t *boards, uint8_t *done_out, RngKey key)
{
    NXN_PROLOGUE(S)
    nxn::Rows<S> b;
    uint32_t w0, w1, w2, w3;
    env_block(key, env_base + i, STREAM_RESET, w0, w1, w2, w3);
    nxn::fresh<S>(b, w0, w1, w2, w3);
    store_board<S>(strip, boards, first, n, lane, b);
    if (active && done_out) done_out[i] = 0;
}

template <int S>
__global__ __launch_bounds__(NXN_BLOCK) void nxn_spawn_random_kernel(uint64_t n, uint64_t env_base, uint8_t *boards, uint32_t seed_lo, uint32_t seed_hi,
                                                                    uint64_t step_index, uint32_t count)
{
    NXN_PROLOGUE(S)
    nxn::Rows<S> b;
    load_board<S>(strip, boards, first, n, lane, b);
    for (uint32_t t = 0; t < count; ++t) {
        const uint64_t step = step_index + t;
        nxn::spawn_word<S>(b, nxn::step_word(RngKey{seed_lo, seed_hi, (uint32_t)step, (uint32_t)(step >> 32)}, env_base + i), true);
    }
    store_board<S>(strip, boards, first, n, lane, b);
    (void)active;
}

// ---- chance outcomes [ref: gameboard.py:132-171 after_state]: every (empty cell, value) outcome, 2 S S slots of S S bytes per
// board (cell in row-major rank, tile 2 then 4), unused slots zero; a full board: itself, count 1.  One board per lane, the
// board in registers as its memory image: the slot of (cell c, tile e) is that image with byte c set, stored dword by dword
// (16-byte vectors when S S is a multiple of 16 and the output is aligned for them; any other alignment: the compiler's
// byte-safe copy).  Every lane writes its own 2 S S S S bytes exactly once.
template <int S, int ALIGN>
__device__ __forceinline__ void store_slot(uint8_t *slot, const uint32_t (&image)[nxn::Shape<S>::DENSE])
{
    constexpr int CELLS = nxn::Shape<S>::CELLS, D = nxn::Shape<S>::DENSE;
    uint8_t *dst = static_cast<uint8_t *>(__builtin_assume_aligned(slot, ALIGN));
    if (CELLS % 4 == 0) {
        __builtin_memcpy(dst, image, CELLS);
    } else {
        __builtin_memcpy(dst, image, 4 * (D - 1));
        const uint32_t tail = image[D - 1];
#pragma unroll
        for (int b = 0; b < CELLS - 4 * (D - 1); ++b) dst[4 * (D - 1) + b] = (uint8_t)(tail >> (8 * b));
    }
}

template <int S, int ALIGN>
__device__ __forceinline__ void write_outcomes(uint8_t *slots, const uint32_t (&dense)[nxn::Shape<S>::DENSE], uint32_t n_empty)
{
    constexpr int CELLS = nxn::Shape<S>::CELLS, D = nxn::Shape<S>::DENSE;
    uint32_t zero[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zero[d] = 0u;
    uint32_t used = 0u;
    if (n_empty == 0u) {  // :156-157 a full board: itself
        store_slot<S, ALIGN>(slots, dense);
        used = 1u;
    } else {
#pragma unroll
        for (int c = 0; c < CELLS; ++c) {
            if (((dense[c >> 2] >> (8 * (c & 3))) & 0xFFu) != 0u) continue;
            uint32_t image[D];
#pragma unroll
            for (int d = 0; d < D; ++d) image[d] = dense[d];
            image[c >> 2] = dense[c >> 2] | (1u << (8 * (c & 3)));
            store_slot<S, ALIGN>(slots + (size_t)used * CELLS, image);
            image[c >> 2] = dense[c >> 2] | (2u << (8 * (c & 3)));
            store_slot<S, ALIGN>(slots + (size_t)(used + 1u) * CELLS, image);
            used += 2u;
        }
    }
    for (uint32_t t = used; t < 2u * CELLS; ++t) store_slot<S, ALIGN>(slots + (size_t)t * CELLS, zero);
}

template <int S>
__global__ __launch_bounds__(NXN_BLOCK) void nxn_chance_outcomes_kernel(uint64_t n, const uint8_t *boards, uint8_t *out, uint8_t *count_out)
{
    NXN_PROLOGUE(S)
    using Sh = nxn::Shape<S>;
    nxn::Rows<S> b;
    load_board<S>(strip, boards, first, n, lane, b);
    if (!active) return;  // (nothing below is shared with other lanes)
    uint32_t dense[Sh::DENSE];
    nxn::dense_from_rows<S>(b, dense);
    const uint32_t n_empty = nxn::count_empty<S>(b);
    uint8_t *slots = out + i * 2u * Sh::CELLS * Sh::CELLS;
    // (2 S S S S bytes per board: a slot is 16-byte aligned iff the output is and S S is a multiple of 16, 4-byte aligned iff ... of 4)
    if (Sh::CELLS % 16 == 0 && ((uintptr_t)out & 15u) == 0u)
        write_outcomes<S, 16>(slots, dense, n_empty);
    else if (Sh::CELLS % 4 == 0 && ((uintptr_t)out & 3u) == 0u)
        write_outcomes<S, 4>(slots, dense, n_empty);
    else
        write_outcomes<S, 1>(slots, dense, n_empty);
    count_out[i] = (uint8_t)(n_empty == 0u ? 1u : 2u * n_empty);
}

// The same enumeration with ONE BOARD PER WAVE (even sizes, a 4-byte aligned output): the 2 S S slots of a board are one
// contiguous block of 2 S^4 bytes, and the per-lane form above writes 64 such blocks 2 S^4 bytes apart with every store
// instruction -- a power-of-two stride for S = 4, 8 that lands on few memory channels (1.0 / 1.9 TB/s written).  Here the 64
// lanes of a wave write consecutive dwords of one board's block: lane l of round r owns dword 64 r + l, i.e. bytes 4 ... of
// slot t = (256 r + 4 l) / (S S), and builds it from the board (in LDS) and the rank-ordered list of the board's empty cells
// (a ballot over "cell c is empty" + mbcnt gives every empty cell its rank; the list lives in LDS too).  Waves stride over
// the boards of the batch.
constexpr int OUTCOME_WAVES_PER_BLOCK = 4;

template <int S>
__global__ __launch_bounds__(64 * OUTCOME_WAVES_PER_BLOCK) void nxn_chance_outcomes_wave_kernel(uint64_t n, const uint8_t *boards, uint32_t *out, uint8_t *count_out)
{
    constexpr uint32_t CELLS = S * S, OUT_DWORDS = 2u * CELLS * CELLS / 4u;
    static_assert(CELLS % 4 == 0, "even sizes only");
    __shared__ __attribute__((aligned(16))) uint32_t cells_lds[OUTCOME_WAVES_PER_BLOCK][16];
    __shared__ uint8_t rank_lds[OUTCOME_WAVES_PER_BLOCK][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint8_t *const cell_bytes = reinterpret_cast<uint8_t *>(cells_lds[wave]);
    const uint64_t waves = (uint64_t)gridDim.x * OUTCOME_WAVES_PER_BLOCK;
    for (uint64_t board = (uint64_t)blockIdx.x * OUTCOME_WAVES_PER_BLOCK + wave; board < n; board += waves) {
        const uint32_t value = lane < CELLS ? boards[board * CELLS + lane] : 0xFFu;
        const bool empty = value == 0u;
        const uint64_t mask = __ballot(empty);
        const uint32_t n_empty = (uint32_t)__popcll(mask);
        strip_fence();  // (the previous board's reads of the two strips are done)
        if (lane < CELLS) cell_bytes[lane] = (uint8_t)value;
        if (empty) rank_lds[wave][__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint8_t)lane;
        strip_fence();
        const uint32_t slots_used = n_empty ? 2u * n_empty : 1u;  // [ref: gameboard.py:156-157] a full board: itself
        uint32_t *const block = out + board * OUT_DWORDS;
#pragma unroll 4
        for (uint32_t w = lane; w < OUT_DWORDS; w += 64u) {
            const uint32_t slot = 4u * w / CELLS, first_cell = 4u * w % CELLS;
            uint32_t word = cells_lds[wave][first_cell >> 2];
            if (n_empty) {
                const uint32_t cell = rank_lds[wave][(slot >> 1) & 63u];  // (& 63: slots past the used ones read a stale byte, zeroed below)
                if ((cell >> 2) == (first_cell >> 2)) word |= ((slot & 1u) + 1u) << (8u * (cell & 3u));  // tile 2 then tile 4 into the empty cell
            }
            block[w] = slot < slots_used ? word : 0u;  // (dword stores: non-temporal ones gain nothing at this width, measured)
        }
        if (lane == 0u) count_out[board] = (uint8_t)slots_used;
    }
}

int check_nxn(const char *who, uint64_t n, uint32_t size)
{
    if (size < 2u || size > (uint32_t)NXN_MAX) return fail(ENV2048_ERR_INVALID_ARG, "%s: size must be in [2, %d], got %u", who, NXN_MAX, size);
    return check_count(n);
}

inline dim3 nxn_grid(uint64_t n) { return dim3((unsigned)((n + NXN_BLOCK - 1) / NXN_BLOCK)); }

// the instantiation of KERNEL for board size `size` (validated by check_nxn), launched on `grid` x NXN_BLOCK
#define NXN_LAUNCH(KERNEL, size, grid, stream, ...)                                                                \
    switch (size) {                                                                                                \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;   \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;   \
    case 4: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;   \
    case 5: hipLaunchKernelGGL(KERNEL<5>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;   \
    case 6: hipLaunchKernelGGL(KERNEL<6>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;   \
    case 7: hipLaunchKernelGGL(KERNEL<7>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;   \
    default: hipLaunchKernelGGL(KERNEL<8>, grid, dim3(NXN_BLOCK), 0, (hipStream_t)(stream), __VA_ARGS__); break;  \
    }

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
    NXN_LAUNCH(nxn_afterstate_kernel, size, nxn_grid(n), stream, n, boards_in, actions, boards_out, reward, changed)
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
    NXN_LAUNCH(nxn_step_kernel, size, nxn_grid(n), stream, n, env_base, boards, actions, reward, done, legal, key, flags)
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
    NXN_LAUNCH(nxn_status_kernel, size, nxn_grid(n), stream, n, boards, done, legal, n_empty, max_exp)
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
    NXN_LAUNCH(nxn_reset_kernel, size, nxn_grid(n), stream, n, env_base, boards, done, key)
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
    NXN_LAUNCH(nxn_spawn_random_kernel, size, nxn_grid(n), stream, n, env_base, boards, (uint32_t)seed, (uint32_t)(seed >> 32), step_index, count)
    return launched("env2048_nxn_spawn_random launch");
}

int env2048_nxn_chance_outcomes(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *out_boards, uint8_t *out_count)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_nxn("env2048_nxn_chance_outcomes", n, size)) return rc;
    if (!boards || !out_boards || !out_count) return fail(ENV2048_ERR_INVALID_ARG, "env2048_nxn_chance_outcomes: NULL argument");
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    if (size % 2u == 0u && ((uintptr_t)out_boards & 3u) == 0u) {  // one board per wave: coalesced stores
        const unsigned blocks = (unsigned)((n + OUTCOME_WAVES_PER_BLOCK - 1) / OUTCOME_WAVES_PER_BLOCK < 8192u ? (n + OUTCOME_WAVES_PER_BLOCK - 1) / OUTCOME_WAVES_PER_BLOCK : 8192u);
        uint32_t *out_words = reinterpret_cast<uint32_t *>(out_boards);
        const dim3 block(64 * OUTCOME_WAVES_PER_BLOCK);
        switch (size) {
        case 2: hipLaunchKernelGGL(nxn_chance_outcomes_wave_kernel<2>, dim3(blocks), block, 0, (hipStream_t)stream, n, boards, out_words, out_count); break;
        case 4: hipLaunchKernelGGL(nxn_chance_outcomes_wave_kernel<4>, dim3(blocks), block, 0, (hipStream_t)stream, n, boards, out_words, out_count); break;
        case 6: hipLaunchKernelGGL(nxn_chance_outcomes_wave_kernel<6>, dim3(blocks), block, 0, (hipStream_t)stream, n, boards, out_words, out_count); break;
        default: hipLaunchKernelGGL(nxn_chance_outcomes_wave_kernel<8>, dim3(blocks), block, 0, (hipStream_t)stream, n, boards, out_words, out_count); break;
        }
        return launched("env2048_nxn_chance_outcomes launch");
    }
    NXN_LAUNCH(nxn_chance_outcomes_kernel, size, nxn_grid(n), stream, n, boards, out_boards, out_count)
    return launched("env2048_nxn_chance_outcomes launch");
}

}  // extern "C"
