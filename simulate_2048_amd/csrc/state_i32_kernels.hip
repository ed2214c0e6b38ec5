// state_i32_kernels.hip -- the reinforce/game GameState API on int32 value boards (one launch per env.py call) and its
// C-ABI entry points of include/env2048.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "board_io.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"

#include <string.h>

#include <atomic>
#include <chrono>

using namespace env2048;
using namespace env2048::host;

namespace {
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

// ---- coalesced int32 boards --------------------------------------------------------------------------------------------------
// A board of int32 values is 64 bytes, so a lane that loads / stores its own board touches 16 bytes in each of 64 different
// 64-byte segments with every instruction.  The batch kernels below go through a wave-private 1 KiB LDS strip instead: the
// 256 rows of a wave's 64 boards are moved by four instructions of 1 KiB of CONSECUTIVE bytes each (lane l of instruction j
// owns row (64 j + l) & 3 of board (64 j + l) >> 2) and converted to / from the packed form on that lane; the strip holds one
// packed dword per row.  (At 2^22 boards: status 4.7 -> see profiles/r04_ee, r04_ff.)
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;  // (no packed row is all ones: exponents are <= 31)

__device__ __forceinline__ void strip_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the strip is wave-private: LDS operations of a wave are ordered
    __builtin_amdgcn_wave_barrier();
}

// boards [first, first + 64) of `boards` -> the packed board of every lane (boards past n: empty)
__device__ __forceinline__ Board load_rows_coalesced(const int32_t *boards, uint32_t *strip, uint64_t first, uint64_t n, uint32_t lane)
{
    const int4 *rows = reinterpret_cast<const int4 *>(boards) + 4 * first;
    const uint64_t live = 4u * (n - first < 64u ? n - first : 64u);
    strip_sync();
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t q = 64u * j + lane;
        strip[q] = q < live ? pack4(rows[q]) : 0u;
    }
    strip_sync();
    const uint4 v = reinterpret_cast<const uint4 *>(strip)[lane];
    return Board{v.x, v.y, v.z, v.w};
}

// the packed board of every lane -> boards [first, first + 64) of `boards` as int32 values (`skip` lanes: nothing is written
// for their board); `mirror` != 0: also `mirror` bytes further on (the caller's pinned-host copy of its results)
__device__ __forceinline__ void store_rows_coalesced(int32_t *boards, int64_t mirror, uint32_t *strip, uint64_t first, uint64_t n, uint32_t lane,
                                                     const Board &bd, bool skip, bool through)
{
    strip_sync();
    reinterpret_cast<uint4 *>(strip)[lane] = skip ? make_uint4(NO_ROW, NO_ROW, NO_ROW, NO_ROW) : make_uint4(bd.r0, bd.r1, bd.r2, bd.r3);
    strip_sync();
    int4 *rows = reinterpret_cast<int4 *>(boards) + 4 * first;
    const uint64_t live = 4u * (n - first < 64u ? n - first : 64u);
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t q = 64u * j + lane;
        const uint32_t x = strip[q];
        if (x != NO_ROW && q < live) {
            const int4 v = unpack4(x);
            if (through)
                store_x4_through(rows + q, u32x4{(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w});
            else
                rows[q] = v;
            if (mirror) *reinterpret_cast<int4 *>(reinterpret_cast<char *>(rows + q) + mirror) = v;
        }
    }
}

// the f32 observation of every lane's board, coalesced (board_io.hpp), and its host-mirror copy
__device__ __forceinline__ void store_obs_rows_coalesced(float *obs, int64_t mirror, uint32_t *strip, uint64_t first, uint64_t n, uint32_t lane,
                                                         const Board &bd, bool skip, bool through)
{
    store_obs_coalesced(obs, strip, first, n, lane, bd, skip, through);
    if (mirror) store_obs_coalesced(reinterpret_cast<float *>(reinterpret_cast<char *>(obs) + mirror), strip, first, n, lane, bd, skip);
}

// Completion signal of a SINGLE-WORKGROUP launch with a host mirror (the *_signalled entry points): once every thread's
// stores -- the device copies and the host-mirror copies -- are performed at system scope, one thread stores
// `sequence_value` into the caller's pinned host word.  The host polls that word instead of synchronising the stream
// (an interrupt-driven wake-up that costs more than the whole move at the reference's batch sizes).  Every thread of the
// block calls this exactly once, on every path, as its last action; the kernel then ends by itself -- nothing on the GPU
// ever waits for the host.  sequence == NULL (wave-uniform): no barrier, no fence, nothing.
static_assert(BLOCK >= ENV2048_SIGNAL_MAX_ENVS, "a signalled launch (n <= ENV2048_SIGNAL_MAX_ENVS) must be ONE workgroup: signal_sequence orders with a workgroup barrier");
__device__ __forceinline__ void signal_sequence(uint64_t *sequence, uint64_t value)
{
    if (!sequence) return;
    __threadfence_system();  // this thread's result stores are visible to the host before ...
    __syncthreads();         // ... any thread of the (only) block gets past here
    if (threadIdx.x == 0) __hip_atomic_store(sequence, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One environment: `*_in` -> `*_out` (the same pointers for the in-place entry point): a finished game copies through.
// COOP: the board comes in packed (`bd`, loaded by the wave together) and the new board / the observation are NOT stored here
// -- the caller stores them for the whole wave (store_rows_coalesced / store_obs_rows_coalesced) when `moved` comes back true
// (a frozen game's board is copied through here, value by value, as it came in).
template <bool COOP>
__device__ __forceinline__ void step_state_i32_one(uint64_t i, uint64_t env_base, const int32_t *boards_in, int32_t *boards_out,
                                                   const int32_t *actions, const int32_t *step_count_in, int32_t *step_count_out,
                                                   const uint8_t *done_in, uint8_t *done_out, const float *total_in, float *total_out,
                                                   float *reward_out, int32_t *max_tile, RngKey key, const uint2 *keys, float *obs_out,
                                                   uint32_t *legal4_out, int64_t mirror, Board &bd, bool &moved)
{
    // `mirror` != 0: every output element is ALSO written `mirror` bytes further on -- the caller's pinned-host copy of its
    // result block (a few hundred bytes at the reference's batch sizes), so that reading the results back needs no copy
    auto put = [mirror](auto *p, auto v) {
        *p = v;
        if (mirror) *reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + mirror) = v;
    };
    const int4 *rows_in = reinterpret_cast<const int4 *>(boards_in) + 4 * i;
    int4 *rows_out = reinterpret_cast<int4 *>(boards_out) + 4 * i;
    if (!COOP) bd = Board{pack4(rows_in[0]), pack4(rows_in[1]), pack4(rows_in[2]), pack4(rows_in[3])};
    const bool was_done = done_in[i] != 0;
    moved = !was_done;
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
        if (!COOP) {
            put(rows_out + 0, unpack4(bd.r0));
            put(rows_out + 1, unpack4(bd.r1));
            put(rows_out + 2, unpack4(bd.r2));
            put(rows_out + 3, unpack4(bd.r3));
        }
        legal_bits = legal_mask(bd);
        done = legal_bits == 0u ? 1 : 0;
        steps += 1;
        total += reward;
    } else if (legal4_out) {
        legal_bits = legal_mask(bd);  // (a frozen game: the mask of the board as it stands)
    }
    // what the caller asks about the NEW state right after a move (reinforce/training/self_play.py:316-317), in the same
    // launch: observation and (N, 4) legal bytes of the board this step leaves behind -- a finished game's unchanged board
    if (obs_out && !COOP) {
        store_obs(obs_out, i, bd);
        if (mirror) store_obs(reinterpret_cast<float *>(reinterpret_cast<char *>(obs_out) + mirror), i, bd);
    }
    if (legal4_out) put(legal4_out + i, (legal_bits & 1u) | ((legal_bits & 2u) << 7) | ((legal_bits & 4u) << 14) | ((legal_bits & 8u) << 21));
    if (was_done && !in_place) {  // the values as they came in (not re-derived from the packed form)
        put(rows_out + 0, rows_in[0]);
        put(rows_out + 1, rows_in[1]);
        put(rows_out + 2, rows_in[2]);
        put(rows_out + 3, rows_in[3]);
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

__global__ __launch_bounds__(BLOCK) void step_state_i32_kernel(uint64_t n, uint64_t env_base, const int32_t *boards_in,
                                                               int32_t *boards_out, const int32_t *actions,
                                                               const int32_t *step_count_in, int32_t *step_count_out,
                                                               const uint8_t *done_in, uint8_t *done_out,
                                                               const float *total_in, float *total_out, float *reward_out,
                                                               int32_t *max_tile, RngKey key, const uint2 *keys, float *obs_out,
                                                               uint32_t *legal4_out, int64_t mirror, uint64_t *sequence,
                                                               uint64_t sequence_value, bool through)
{
    __shared__ __attribute__((aligned(16))) uint32_t strips[BLOCK / 64][256];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (block_for(): a small launch runs a smaller workgroup)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t first = i - lane;
    uint32_t *const strip = strips[threadIdx.x >> 6];
    if (first < n) {  // (wave-uniform)
        Board bd = load_rows_coalesced(boards_in, strip, first, n, lane);
        bool moved = false;
        if (i < n)
            step_state_i32_one<true>(i, env_base, boards_in, boards_out, actions, step_count_in, step_count_out, done_in, done_out, total_in,
                                     total_out, reward_out, max_tile, key, keys, obs_out, legal4_out, mirror, bd, moved);
        store_rows_coalesced(boards_out, mirror, strip, first, n, lane, bd, !moved, through);
        if (obs_out) store_obs_rows_coalesced(obs_out, mirror, strip, first, n, lane, bd, i >= n, through);
    }
    signal_sequence(sequence, sequence_value);  // (outside the bounds check: every thread of the block takes part)
}

// The same move with the keys and actions of a SMALL batch travelling in the kernel-argument block (env2048_*_inline):
// the host copies them out of ordinary host memory at call time, so nothing has to be staged in pinned or device memory and
// the kernel's first loads do not cross the bus (measured at N = 8: 2 us less from launch to completion than reading the
// same 96 bytes from pinned host memory).  One workgroup; lane i reads entry i of the block.
struct InlineInputs {
    uint2 keys[ENV2048_INLINE_MAX_ENVS];
    int32_t actions[ENV2048_INLINE_MAX_ENVS];
};

__global__ __launch_bounds__(ENV2048_INLINE_MAX_ENVS) void step_state_i32_inline_kernel(
    uint64_t n, const int32_t *boards_in, int32_t *boards_out, const int32_t *step_count_in, int32_t *step_count_out, const uint8_t *done_in,
    uint8_t *done_out, const float *total_in, float *total_out, float *reward_out, int32_t *max_tile, float *obs_out, uint32_t *legal4_out,
    int64_t mirror, uint64_t *sequence, uint64_t sequence_value, const InlineInputs in)
{
    const uint64_t i = threadIdx.x;
    if (i < n) {  // (the latency path, at most one wave of boards: every lane moves its own board)
        Board bd;
        bool moved;
        step_state_i32_one<false>(i, 0u, boards_in, boards_out, in.actions, step_count_in, step_count_out, done_in, done_out, total_in, total_out,
                                  reward_out, max_tile, RngKey{0u, 0u, 0u, 0u}, in.keys, obs_out, legal4_out, mirror, bd, moved);
    }
    signal_sequence(sequence, sequence_value);
}

// batched_reset with one PRNG key per environment, straight to int32 value boards: board i = the fresh board of
// environment 0 at step 0 of the stream seeded with keys[i] (= create_initial_board(keys[i])).
// step_count / done / total (each may be NULL): the bookkeeping fields of a fresh GameState, written as zeros.
// (The board and its observation are stored by the caller, for the whole wave at once.)
__device__ __forceinline__ Board reset_keyed_i32_one(uint64_t i, const uint2 *keys, uint32_t *legal4_out, int32_t *step_count, uint8_t *done,
                                                     float *total, int64_t mirror)
{
    auto put = [mirror](auto *p, auto v) {  // (see step_state_i32_kernel)
        *p = v;
        if (mirror) *reinterpret_cast<decltype(p)>(reinterpret_cast<char *>(p) + mirror) = v;
    };
    const uint2 k = keys[i];
    uint32_t w0, w1, w2, w3;
    env_block(RngKey{k.y, k.x, 0u, 0u}, 0u, STREAM_RESET, w0, w1, w2, w3);
    const Board bd = fresh_board(w0, w1, w2, w3);
    if (legal4_out) {
        const uint32_t bits = legal_mask(bd);
        put(legal4_out + i, (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21));
    }
    if (step_count) put(step_count + i, (int32_t)0);
    if (done) put(done + i, (uint8_t)0);
    if (total) put(total + i, 0.0f);
    return bd;
}

__global__ __launch_bounds__(BLOCK) void reset_keyed_i32_kernel(uint64_t n, const uint2 *keys, int32_t *boards, float *obs_out,
                                                                uint32_t *legal4_out, int32_t *step_count, uint8_t *done, float *total,
                                                                int64_t mirror, uint64_t *sequence, uint64_t sequence_value, bool through)
{
    __shared__ __attribute__((aligned(16))) uint32_t strips[BLOCK / 64][256];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t first = i - lane;
    uint32_t *const strip = strips[threadIdx.x >> 6];
    if (first < n) {  // (wave-uniform)
        Board bd{0u, 0u, 0u, 0u};
        if (i < n) bd = reset_keyed_i32_one(i, keys, legal4_out, step_count, done, total, mirror);
        store_rows_coalesced(boards, mirror, strip, first, n, lane, bd, i >= n, through);
        if (obs_out) store_obs_rows_coalesced(obs_out, mirror, strip, first, n, lane, bd, i >= n, through);
    }
    signal_sequence(sequence, sequence_value);
}

// What the reinforce/game callers ask about a batch of int32 value boards, in one launch: the f32
// observation, the legal-action mask as four 0/1 BYTES per board (a torch/NumPy bool array) and the
// JAX-flavoured done flag ("no action changes the board").
__global__ __launch_bounds__(BLOCK) void status_i32_kernel(uint64_t n, const int32_t *boards, float *obs, uint32_t *legal4,
                                                           uint8_t *done, bool through)
{
    __shared__ __attribute__((aligned(16))) uint32_t strips[BLOCK / 64][256];
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t first = i - lane;
    if (first >= n) return;  // (a whole wave; nothing below is shared between waves)
    uint32_t *const strip = strips[threadIdx.x >> 6];
    const Board bd = load_rows_coalesced(boards, strip, first, n, lane);
    if (obs) store_obs_coalesced(obs, strip, first, n, lane, bd, i >= n, through);
    if (i >= n) return;
    if (legal4 || done) {
        const uint32_t bits = legal_mask(bd);
        if (legal4) legal4[i] = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);
        if (done) done[i] = bits == 0u ? 1 : 0;
    }
}

// latent_state / slide_and_merge / merge_row of the JAX-shaped API on int32 value boards [ref: reinforce/game/core.py:129-170,
// :104-126, :43-101]: slide + merge towards `action` (one move for the whole batch, or one per board), no spawn.
__global__ __launch_bounds__(BLOCK) void afterstate_i32_kernel(uint64_t n, const int32_t *boards_in, const int32_t *actions, uint32_t action,
                                                               int32_t *boards_out, int32_t *reward, uint8_t *changed_out, bool through)
{
    __shared__ __attribute__((aligned(16))) uint32_t strips[BLOCK / 64][256];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t first = i - lane;
    if (first >= n) return;  // (a whole wave; nothing below is shared between waves)
    uint32_t *const strip = strips[threadIdx.x >> 6];
    Board bd = load_rows_coalesced(boards_in, strip, first, n, lane);
    uint32_t score;
    bool changed;
    afterstate(bd, ((actions && i < n) ? (uint32_t)actions[i] : action) & 3u, score, changed);
    store_rows_coalesced(boards_out, 0, strip, first, n, lane, bd, i >= n, through);
    if (i >= n) return;
    if (reward) reward[i] = (int32_t)score;
    if (changed_out) changed_out[i] = changed ? 1 : 0;
}

// A launch that fits one workgroup gets a workgroup of only as many waves as it has boards for (at the reference's batch
// sizes, N = 2 ... 8, one wave instead of eight: less to dispatch, and the completion signal's barrier spans one wave).
inline dim3 block_for(uint64_t n) { return dim3(n >= (uint64_t)BLOCK ? (unsigned)BLOCK : (unsigned)((n + 63) / 64 * 64)); }

int check_sequence(const char *who, uint64_t n, int64_t host_mirror_delta, const uint64_t *host_sequence)
{
    if (!host_sequence) return ENV2048_OK;
    if (misaligned(host_sequence, 8)) return fail(ENV2048_ERR_ALIGNMENT, "%s: host_sequence must be 8-byte aligned", who);
    if (!host_mirror_delta) return fail(ENV2048_ERR_INVALID_ARG, "%s: a completion word goes with a host mirror (host_mirror_delta != 0)", who);
    if (n > ENV2048_SIGNAL_MAX_ENVS)
        return fail(ENV2048_ERR_INVALID_ARG, "%s: a completion word needs a single-workgroup launch (n <= %d), got n = %llu", who,
                    ENV2048_SIGNAL_MAX_ENVS, (unsigned long long)n);
    return ENV2048_OK;
}

}  // namespace

extern "C" {

static int step_state_i32_common(const char *who, int device, void *stream, uint64_t n, uint64_t env_base, const uint32_t *keys,
                                 const int32_t *boards_in, int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                 int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in,
                                 float *total_reward_out, float *reward, int32_t *max_tile, uint64_t seed, uint64_t step_index,
                                 float *obs_out = nullptr, uint8_t *legal4_out = nullptr, int64_t host_mirror_delta = 0,
                                 uint64_t *host_sequence = nullptr, uint64_t sequence_value = 0)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_sequence(who, n, host_mirror_delta, host_sequence)) return rc;
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
    hipLaunchKernelGGL(step_state_i32_kernel, grid_for(n), block_for(n), 0, (hipStream_t)stream, n, env_base, boards_in, boards_out,
                       actions, step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward,
                       max_tile, key, reinterpret_cast<const uint2 *>(keys), obs_out, reinterpret_cast<uint32_t *>(legal4_out),
                       host_mirror_delta, host_sequence, sequence_value, wide_stores_through());
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

int env2048_reset_state_i32_keyed_signalled(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards, int32_t *step_count,
                                            uint8_t *done, float *total_reward, float *obs, uint8_t *legal4, int64_t host_mirror_delta,
                                            uint64_t *host_sequence, uint64_t sequence_value)
{
    if (n == 0) return ENV2048_OK;
    if (int rc = check_sequence("env2048_reset_state_i32_keyed_signalled", n, host_mirror_delta, host_sequence)) return rc;
    if (!keys || !boards) return fail(ENV2048_ERR_INVALID_ARG, "env2048_reset_keyed_i32: keys and boards are required");
    if (misaligned(keys, 8) || misaligned(boards, 16) || misaligned(obs, 16) || misaligned(legal4, 4) || misaligned(step_count, 4) ||
        misaligned(total_reward, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "env2048_reset_keyed_i32: keys must be 8-byte, boards / obs 16-byte, the other arrays 4-byte aligned");
    if (host_mirror_delta & 15) return fail(ENV2048_ERR_ALIGNMENT, "env2048_reset_state_i32_keyed: host_mirror_delta must be a multiple of 16 bytes");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(reset_keyed_i32_kernel, grid_for(n), block_for(n), 0, (hipStream_t)stream, n, reinterpret_cast<const uint2 *>(keys),
                       boards, obs, reinterpret_cast<uint32_t *>(legal4), step_count, done, total_reward, host_mirror_delta, host_sequence,
                       sequence_value, wide_stores_through());
    return launched("env2048_reset_keyed_i32 launch");
}

int env2048_reset_state_i32_keyed(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards, int32_t *step_count,
                                  uint8_t *done, float *total_reward, float *obs, uint8_t *legal4, int64_t host_mirror_delta)
{
    return env2048_reset_state_i32_keyed_signalled(device, stream, n, keys, boards, step_count, done, total_reward, obs, legal4,
                                                   host_mirror_delta, nullptr, 0);
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

int env2048_step_state_i32_keyed_signalled(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                           int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                           int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in,
                                           float *total_reward_out, float *reward, int32_t *max_tile, float *obs, uint8_t *legal4,
                                           int64_t host_mirror_delta, uint64_t *host_sequence, uint64_t sequence_value)
{
    if (n && !keys) return fail(ENV2048_ERR_INVALID_ARG, "env2048_step_state_i32_keyed_signalled: keys is NULL");
    return step_state_i32_common("env2048_step_state_i32_keyed_signalled", device, stream, n, 0, keys, boards_in, boards_out, actions,
                                 step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward, max_tile,
                                 0, 0, obs, legal4, host_mirror_delta, host_sequence, sequence_value);
}

int env2048_step_state_i32_keyed_inline(int device, void *stream, uint64_t n, const uint32_t *host_keys, const int32_t *boards_in,
                                        int32_t *boards_out, const int32_t *host_actions, const int32_t *step_count_in,
                                        int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out, const float *total_reward_in,
                                        float *total_reward_out, float *reward, int32_t *max_tile, float *obs, uint8_t *legal4,
                                        int64_t host_mirror_delta, uint64_t *host_sequence, uint64_t sequence_value)
{
    const char *who = "env2048_step_state_i32_keyed_inline";
    if (n == 0) return ENV2048_OK;
    if (n > ENV2048_INLINE_MAX_ENVS) return fail(ENV2048_ERR_INVALID_ARG, "%s: n = %llu exceeds ENV2048_INLINE_MAX_ENVS (%d)", who, (unsigned long long)n, ENV2048_INLINE_MAX_ENVS);
    if (!host_keys || !host_actions) return fail(ENV2048_ERR_INVALID_ARG, "%s: host_keys and host_actions are required", who);
    if (int rc = check_sequence(who, n, host_mirror_delta, host_sequence)) return rc;
    if (misaligned(obs, 16) || misaligned(legal4, 4)) return fail(ENV2048_ERR_ALIGNMENT, "%s: obs must be 16-byte, legal4 4-byte aligned", who);
    if (host_mirror_delta & 15) return fail(ENV2048_ERR_ALIGNMENT, "%s: host_mirror_delta must be a multiple of 16 bytes", who);
    if (host_mirror_delta && boards_in == boards_out)
        return fail(ENV2048_ERR_INVALID_ARG, "%s: a host mirror goes with the out-of-place form (outputs in one block of their own)", who);
    if (!boards_in || !boards_out || !done_in || !done_out || !reward) return fail(ENV2048_ERR_INVALID_ARG, "%s: boards, done and reward are required", who);
    if ((step_count_in == nullptr) != (step_count_out == nullptr) || (total_reward_in == nullptr) != (total_reward_out == nullptr))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: step_count / total_reward need both their in and out array", who);
    if (misaligned(boards_in, 16) || misaligned(boards_out, 16)) return fail(ENV2048_ERR_ALIGNMENT, "%s: boards must be 16-byte aligned", who);
    if (misaligned(host_keys, 4) || misaligned(host_actions, 4) || misaligned(step_count_in, 4) || misaligned(step_count_out, 4) ||
        misaligned(total_reward_in, 4) || misaligned(total_reward_out, 4) || misaligned(reward, 4) || misaligned(max_tile, 4))
        return fail(ENV2048_ERR_ALIGNMENT, "%s: misaligned 4-byte array", who);
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    InlineInputs in;  // (entries beyond n are never read by the kernel)
    memcpy(in.keys, host_keys, n * sizeof(uint2));
    memcpy(in.actions, host_actions, n * sizeof(int32_t));
    hipLaunchKernelGGL(step_state_i32_inline_kernel, dim3(1), dim3(ENV2048_INLINE_MAX_ENVS), 0, (hipStream_t)stream, n, boards_in, boards_out,
                       step_count_in, step_count_out, done_in, done_out, total_reward_in, total_reward_out, reward, max_tile, obs,
                       reinterpret_cast<uint32_t *>(legal4), host_mirror_delta, host_sequence, sequence_value, in);
    return launched(who);
}

// Host side of the completion word: poll until *host_sequence == value (1) or timeout_us have passed (0).  No device call.
int env2048_wait_sequence(const uint64_t *host_sequence, uint64_t value, uint32_t timeout_us)
{
    if (!host_sequence || misaligned(host_sequence, 8)) return fail(ENV2048_ERR_INVALID_ARG, "env2048_wait_sequence: host_sequence is NULL or misaligned");
    const auto *word = reinterpret_cast<const std::atomic<uint64_t> *>(host_sequence);
    if (word->load(std::memory_order_acquire) == value) return 1;
    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(timeout_us);
    for (;;) {
        for (int spin = 0; spin < 64; ++spin) {
            if (word->load(std::memory_order_acquire) == value) return 1;
            __builtin_ia32_pause();
        }
        if (std::chrono::steady_clock::now() >= until) return word->load(std::memory_order_acquire) == value ? 1 : 0;
    }
}

int env2048_step_state_i32(int device, void *stream, uint64_t n, uint64_t env_base, int32_t *boards, const int32_t *actions,
                           int32_t *step_count, uint8_t *done_io, float *total_reward, float *reward, int32_t *max_tile,
                           uint64_t seed, uint64_t step_index)
{
    return env2048_step_state_i32_to(device, stream, n, env_base, boards, boards, actions, step_count, step_count, done_io, done_io,
                                     total_reward, total_reward, reward, max_tile, seed, step_index);
}

int env2048_afterstate_i32(int device, void *stream, uint64_t n, const int32_t *boards_in, const int32_t *actions, uint32_t action,
                           int32_t *boards_out, int32_t *reward, uint8_t *changed)
{
    if (n == 0) return ENV2048_OK;
    if (!boards_in || !boards_out) return fail(ENV2048_ERR_INVALID_ARG, "env2048_afterstate_i32: boards_in and boards_out are required");
    if (misaligned(boards_in, 16) || misaligned(boards_out, 16)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstate_i32: boards must be 16-byte aligned");
    if (misaligned(actions, 4) || misaligned(reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "env2048_afterstate_i32: actions / reward must be 4-byte aligned");
    if (int rc = check_count(n)) return rc;
    DeviceScope scope(device);
    if (scope.rc) return scope.rc;
    hipLaunchKernelGGL(afterstate_i32_kernel, grid_for(n), block_for(n), 0, (hipStream_t)stream, n, boards_in, actions, action, boards_out, reward,
                       changed, wide_stores_through());
    return launched("env2048_afterstate_i32 launch");
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
                       reinterpret_cast<uint32_t *>(legal4), done, wide_stores_through());
    return launched("env2048_status_i32 launch");
}

}  // extern "C"
