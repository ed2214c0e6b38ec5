// rollout_kernels.hip -- many steps per launch with the boards in registers (rollout / replay), episode segmentation of
// recorded rows, and their C-ABI entry points of include/env2048.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "env2048.h"
#include "board_ops.hpp"
#include "board_io.hpp"
#include "tiling.hpp"
#include "device_scope.hpp"
#include "philox_strip.hpp"

using namespace env2048;
using namespace env2048::host;

namespace {

// plane-space transitions of the multi-step kernels (board_ops.hpp, transition_oriented): 16 selector records
// [orientation][action] followed by 4 rank-mask records [action], 16 bytes each
__constant__ uint32_t ORIENT_TABLE[80] = {ENV2048_ORIENT_SELECTOR_VALUES, ENV2048_ORIENT_RANK_MASK_VALUES};

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

constexpr uint32_t EPISODE_ROWS_IN_FLIGHT = 8;  // rows whose loads are issued together (the walk itself is sequential; 16 / 32 measured: no gain)

// Both kernels exist for E = 1 and E = 4 adjacent games per lane.  With one game per lane a wave reads 64 bytes of a `done` /
// `max_exp` row per load instruction -- a stream of tiny requests (1.9 - 2.4 TB/s of row bytes, profiles/r04_jj_*); with four,
// a lane takes a dword of flags and 16 bytes of rewards per row and writes 16 bytes of `episode_of`, and walks four independent
// games.  E = 4 needs n and row_stride to be multiples of 4 and the rows aligned for those accesses (the host checks).
template <int E>
__device__ __forceinline__ uint32_t load_flags(const uint8_t *row, uint64_t i0)
{
    if (E == 4) return *reinterpret_cast<const uint32_t *>(row + i0);
    return row[i0];
}

// Records per game, in closed form -- no walk: a window of R > 0 rows holds one record per finished game plus the open one at
// its end (the last row did not finish a game); a FROZEN game holds exactly one (self_play.py:359,368: it stops recording at
// its first `done`).  So the pass is a sum of flags down the column: for E = 4 one dword add per row and lane (flags
// normalised to 0 / 1 per byte; the byte sums are widened every 255 rows).
template <int E>
__global__ __launch_bounds__(256) void count_episodes_kernel(EpisodeRows a, uint32_t *counts)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * E;
    if (i0 >= a.n) return;
    const bool frozen = (a.flags & ENV2048_EPISODES_FROZEN) != 0u;
    uint32_t count[E], phys = a.first_row, packed = 0u, last = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) count[e] = 0u;
    if (!frozen) {
        uint32_t since_widened = 0u;
        for (uint32_t r0 = 0; r0 < a.rows; r0 += EPISODE_ROWS_IN_FLIGHT) {
            uint32_t d[EPISODE_ROWS_IN_FLIGHT];
            const uint32_t batch = min(EPISODE_ROWS_IN_FLIGHT, a.rows - r0);
#pragma unroll
            for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
                d[u] = 0u;
                if (u < batch) {
                    d[u] = load_flags<E>(a.done + (uint64_t)phys * a.row_stride, i0);
                    phys = phys + 1u == a.ring_rows ? 0u : phys + 1u;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
                if (u < batch) {
                    last = (((d[u] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d[u]) >> 7 & 0x01010101u;  // 1 in every byte that is not zero
                    packed += last;
                }
            }
            since_widened += batch;
            if (since_widened > 255u - EPISODE_ROWS_IN_FLIGHT) {  // (before any byte can pass 255)
#pragma unroll
                for (int e = 0; e < E; ++e) count[e] += (packed >> (8 * e)) & 0xFFu;
                packed = 0u;
                since_widened = 0u;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t finished = count[e] + ((packed >> (8 * e)) & 0xFFu);
        const uint32_t open = a.rows != 0u && ((last >> (8 * e)) & 1u) == 0u ? 1u : 0u;
        counts[i0 + e] = frozen ? (a.rows != 0u ? 1u : 0u) : finished + open;
    }
}

// STREAM_OF (E = 4 only): `episode_of` is a write stream beyond the Infinity Cache -- non-temporal stores (device_scope.hpp).
template <int E, bool STREAM_OF = false>
__global__ __launch_bounds__(256) void segment_episodes_kernel(EpisodeRows a, const uint64_t *offsets, env2048_episode *episodes,
                                                               int32_t *episode_of)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * E;
    if (i0 >= a.n) return;
    const bool frozen = (a.flags & ENV2048_EPISODES_FROZEN) != 0u;
    const bool reward_f32 = (a.flags & ENV2048_EPISODES_REWARD_F32) != 0u;
    uint64_t slot[E];
    uint32_t start[E], length[E], best[E], phys = a.first_row, r = 0u;
    double total[E];
    bool stop[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        slot[e] = offsets[i0 + e];
        start[e] = length[e] = best[e] = 0u;
        total[e] = 0.0;
        stop[e] = false;
    }
    auto emit = [&](int e, bool terminated) {
        env2048_episode rec;
        rec.env = (uint32_t)(i0 + e);
        rec.start = start[e];
        rec.length = length[e];
        rec.terminated = terminated ? 1 : 0;
        rec.max_exp = (uint8_t)best[e];
        rec.reserved = 0;
        rec.total_reward = total[e];
        episodes[slot[e]++] = rec;
    };
    bool all_stopped = false;
    while (r < a.rows && !(all_stopped && !episode_of)) {  // (with episode_of the rows after the end of a frozen game get their -1)
        uint32_t d[EPISODE_ROWS_IN_FLIGHT], m[EPISODE_ROWS_IN_FLIGHT];
        uint32_t w[EPISODE_ROWS_IN_FLIGHT][E];  // reward bits (int32 or float32)
        uint64_t at[EPISODE_ROWS_IN_FLIGHT];
        const uint32_t batch = min(EPISODE_ROWS_IN_FLIGHT, a.rows - r);
#pragma unroll
        for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
            d[u] = m[u] = 0u;
            at[u] = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) w[u][e] = 0u;
            if (u < batch) {
                at[u] = (uint64_t)phys * a.row_stride + i0;
                phys = phys + 1u == a.ring_rows ? 0u : phys + 1u;
                d[u] = load_flags<E>(a.done, at[u]);
                if (a.reward) {
                    if (E == 4) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(static_cast<const uint32_t *>(a.reward) + at[u]);
                        w[u][0] = v.x, w[u][E > 1 ? 1 : 0] = v.y, w[u][E > 2 ? 2 : 0] = v.z, w[u][E > 3 ? 3 : 0] = v.w;
                    } else {
                        w[u][0] = static_cast<const uint32_t *>(a.reward)[at[u]];
                    }
                }
                if (a.max_exp) m[u] = load_flags<E>(a.max_exp, at[u]);
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < EPISODE_ROWS_IN_FLIGHT; ++u) {
            if (u < batch) {
                int32_t owner[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    owner[e] = -1;  // frozen: rows after the end of the game belong to no episode
                    if (!stop[e]) {
                        total[e] += reward_f32 ? (double)__uint_as_float(w[u][e]) : (double)(int32_t)w[u][e];
                        best[e] = max(best[e], (m[u] >> (8 * e)) & 0xFFu);
                        ++length[e];
                        owner[e] = (int32_t)slot[e];
                        if ((d[u] >> (8 * e)) & 0xFFu) {
                            emit(e, true);
                            start[e] = r + u + 1u;
                            length[e] = 0u;
                            best[e] = 0u;
                            total[e] = 0.0;
                            stop[e] = frozen;
                        }
                    }
                }
                if (episode_of) {
                    if (E == 4 && STREAM_OF) {
                        typedef int32_t vec4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(vec4{owner[0], owner[E > 1 ? 1 : 0], owner[E > 2 ? 2 : 0], owner[E > 3 ? 3 : 0]},
                                                    reinterpret_cast<vec4 *>(episode_of + at[u]));
                    } else if (E == 4)
                        *reinterpret_cast<int4 *>(episode_of + at[u]) = make_int4(owner[0], owner[E > 1 ? 1 : 0], owner[E > 2 ? 2 : 0], owner[E > 3 ? 3 : 0]);
                    else
                        episode_of[at[u]] = owner[0];
                }
            }
        }
        r += batch;
        all_stopped = true;
#pragma unroll
        for (int e = 0; e < E; ++e) all_stopped = all_stopped && stop[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (length[e]) emit(e, false);
}

// E = 4 when every access of the four-games-per-lane form is aligned
inline bool episodes_by_four(uint64_t n, uint64_t row_stride, const void *done, const void *reward, const void *max_exp, const void *episode_of)
{
    return n % 4 == 0 && row_stride % 4 == 0 && !misaligned(done, 4) && !misaligned(max_exp, 4) && !misaligned(reward, 16) && !misaligned(episode_of, 16);
}

}  // namespace

extern "C" {

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
    if (episodes_by_four(n, row_stride, done, nullptr, nullptr, nullptr))
        hipLaunchKernelGGL(count_episodes_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, counts);
    else
        hipLaunchKernelGGL(count_episodes_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, counts);
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
    if (episodes_by_four(n, row_stride, done, reward, max_exp, episode_of) && episode_of && stream_output((uint64_t)rows * n * 4u))
        hipLaunchKernelGGL((segment_episodes_kernel<4, true>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, offsets,
                           episodes, episode_of);
    else if (episodes_by_four(n, row_stride, done, reward, max_exp, episode_of))
        hipLaunchKernelGGL((segment_episodes_kernel<4, false>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, offsets,
                           episodes, episode_of);
    else
        hipLaunchKernelGGL(segment_episodes_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, offsets, episodes,
                           episode_of);
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

}  // extern "C"
