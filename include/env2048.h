/*
 * env2048.h -- C-ABI of the MI355X-native batched 2048 environment (libenv2048.so).
 *
 * The reference (schalappe/simulate_2048) is pure Python: it has no FFI/plugin interface for
 * this path, only ordinary Python call signatures.  This header therefore DEFINES the boundary a
 * maintainer would bind with ctypes; each entry point names the reference
 * function(s) it replaces as [replaces: file:lines] (paths relative to the reference repo root).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer on HIP device `device` (plain hipMalloc memory or the
 *     storage of a torch tensor); the caller owns every buffer; the library allocates nothing
 *     and keeps no per-batch state.  Environment state is exactly `boards` + (seed, step_index).
 *     "Device pointer" means device-ACCESSIBLE: pinned (hipHostMalloc) host memory is as good wherever a few hundred bytes
 *     move -- the mirrors stage small key / action arrays through it and env2048_*_mirrored write result copies into it.
 *     The only process-lifetime state is per device and independent of any batch: one internal stream, one helper thread
 *     and two events, created by the first env2048_step_sequence call that runs as two chains (see there, and "The helper
 *     thread" below it for what that thread costs and how to switch its polling off).
 *   - `stream` is a hipStream_t (NULL = the device's default stream).  Calls are ASYNCHRONOUS on
 *     that stream; the caller synchronises.  Entry points are re-entrant and thread-safe.  (One exception, stronger, not
 *     weaker: an env2048_step_sequence call that takes the direct route returns with its results complete -- see there.)
 *   - Return value: 0 (ENV2048_OK) or a negative error code; env2048_last_error() returns a
 *     thread-local message for the last failure on the calling thread.  No exceptions, no traps
 *     on bad *data* (actions are taken mod 4; exponents > 31 are undefined input).
 *   - n == 0 is a successful no-op.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with ENV2048_ERR_DEVICE.
 *
 * Board layout ("packed log2"): 16 bytes per board, byte i = 4*row + col (row-major, the order
 * of ndarray.flatten()), value e: 0 = empty, e >= 1 means tile 2^e.  `boards` must be 16-byte
 * aligned; int32/float/int64 arrays naturally aligned.
 *
 * Actions: 0 = left, 1 = up, 2 = right, 3 = down  [twentyfortyeight/envs/twentyfortyeight.py:19].
 *
 * RNG contract (build-defined; the reference's PCG64 / threefry bitstreams are not part of its
 * tested behaviour): Philox4x32-10, key = (seed_lo, seed_hi),
 * counter = (index_lo, step_lo, step_hi, stream + 2*index_hi).  env = env_base + i is always the
 * GLOBAL environment index, so results do not depend on how a batch is sliced across GPUs.
 *   stream 0 ("step words"): index = env >> 2 -- one Philox block serves four neighbouring
 *       environments; environment env owns word w = block[env & 3]:
 *           w & 3                      the on-device random action (RANDOM_ACTIONS, rollout)
 *           m = w & ~3, P = m * n_empty (64-bit)
 *           spawn cell  k = P >> 32    -> k-th empty cell in row-major order (uniform)
 *           spawn tile  4 (e=2) iff (P mod 2^32) < 429496729 (= floor(0.1 * 2^32)), else 2 (e=1)
 *       (P mod 2^32 is the arithmetic-decoding remainder: uniform again and independent of k.)
 *   stream 1 ("reset block"): index = env, words (r0,r1,r2,r3):
 *       first spawn  cell r0 >> 28 (of 16 empties),               tile 4 iff r1 < 429496729
 *       second spawn cell floor(r2 * 15 / 2^32) among the other 15, tile 4 iff r3 < 429496729
 *       -- two sequential single spawns on an empty board.
 */
#ifndef ENV2048_H
#define ENV2048_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(ENV2048_BUILDING)
#define ENV2048_EXPORT __attribute__((visibility("default")))
#else
#define ENV2048_EXPORT
#endif

#define ENV2048_ABI_VERSION 5

#define ENV2048_OK 0
#define ENV2048_ERR_INVALID_ARG (-1) /* NULL required pointer, bad flag bits */
#define ENV2048_ERR_ALIGNMENT (-2)   /* misaligned buffer */
#define ENV2048_ERR_DEVICE (-3)      /* no such device / no HIP runtime / no GPU */
#define ENV2048_ERR_RANGE (-4)       /* env_base + n overflows the 2^63 global index range */
#define ENV2048_ERR_HIP_BASE (-1000) /* HIP runtime error e is reported as -1000 - e */

/* step / rollout flags */
#define ENV2048_FLAG_AUTO_RESET 1u     /* a board that became terminal is replaced by a fresh one */
#define ENV2048_FLAG_RANDOM_ACTIONS 2u /* ignore `actions`; use the low two bits of the step word */
#define ENV2048_FLAG_OBS_F16 4u        /* env2048_step only: `obs` is n x 16 IEEE binary16 instead of float32 */
#define ENV2048_FLAG_OBS_BF16 8u       /* env2048_step only: `obs` is n x 16 bfloat16 (excludes OBS_F16)      */
/* env2048_step* only, tuning hints that change no result: by default a launch of up to 2^23 boards stores
 * write-through (nothing is left dirty in the XCD L2s for the end-of-kernel write-back; the working set of
 * back-to-back steps then lives in the 256 MiB Infinity Cache) and a larger one stores write-back.  These two
 * bits override the size rule (they exclude each other). */
#define ENV2048_FLAG_STORES_WRITE_THROUGH 16u
#define ENV2048_FLAG_STORES_WRITE_BACK 32u
/* env2048_step* only, tuning hints that change no result: the step kernel exists in two tilings -- one board per lane
 * (short dependent chain, four times as many workgroups: faster up to 2^18 boards per launch, where latency is the
 * limit) and four boards per lane (shared Philox blocks, amortised constants: faster above, where instruction issue is
 * the limit).  By default the launch size decides; these two bits override it (they exclude each other).  The rollout /
 * replay / rollout_episodes kernels have the same two tilings (switch at 2^18 environments) and accept the same hints. */
#define ENV2048_FLAG_TILE_SMALL 128u
#define ENV2048_FLAG_TILE_LARGE 256u
/* env2048_step_sequence only, a tuning hint that changes no result: never split the batch into two half-batch chains
 * (see env2048_step_sequence). */
#define ENV2048_FLAG_SINGLE_CHAIN 512u
/* env2048_step_sequence only, a tuning hint that changes no result: never take the direct-dispatch route (the library's own
 * queues, see env2048_step_sequence, "The direct route"): the call stays `count` asynchronous launches on `stream`. */
#define ENV2048_FLAG_NO_DIRECT 1024u
/* env2048_step* only: `reward` is an n-float32 array receiving the LOG-NORMALISED reward log2(score) / 16 (0 for a
 * score of 0) -- normalize_reward of the reference (twentyfortyeight/utils/normalize.py:6-33, the `normalize=True`
 * constructor flag, twentyfortyeight/envs/twentyfortyeight.py:80-81) for a whole batch; within 1 ulp (float32) of the
 * reference's float64 value. */
#define ENV2048_FLAG_REWARD_LOG2_F32 64u

ENV2048_EXPORT int env2048_abi_version(void);
ENV2048_EXPORT const char *env2048_last_error(void);
/* Number of HIP devices visible to the process, or a negative error code. */
ENV2048_EXPORT int env2048_device_count(void);

/*
 * Fresh boards: zeros + two spawned tiles.  Writes boards[i] for i in [0,n) and, if non-NULL,
 * done[i] = 0.  Uses the stream-1 (reset) block of (env_base + i, step_index).
 * [replaces: twentyfortyeight/envs/twentyfortyeight.py:84-104 reset;
 *            reinforce/game/core.py:324-343 create_initial_board; reinforce/game/env.py:55-76 reset,
 *            :176-190 batched_reset]
 */
ENV2048_EXPORT int env2048_reset(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint8_t *done,
                  uint64_t seed, uint64_t step_index);

/*
 * One environment transition per board, in place:
 *     a = action & 3;  after = slide+merge(board, a);  reward = sum of merged tile values
 *     if after != board: board = after with one spawned tile   (valid move)
 *     else:              board unchanged, reward = 0, no spawn   (invalid move)
 *     done = board full and no equal horizontally/vertically adjacent pair (post-spawn board)
 *     if (flags & AUTO_RESET) and done: board = fresh board from the stream-1 block of the
 *                                       same (env, step_index); done[i] still reports 1
 * A board that is already terminal has no valid move, so stepping it is a no-op with reward 0
 * and done = 1 -- the reference's explicit done-guard (reinforce/game/env.py:103-104).
 * Outputs (each may be NULL except boards/done; `actions` may be NULL only with RANDOM_ACTIONS):
 *     reward[i]  int32 score delta            legal[i]  bit a set <=> action a legal on the
 *     done[i]    0/1                                     board written back
 *     obs[16*i..] log2(tile)/16 of the board written back (0 for empty): float32, or with
 *                 ENV2048_FLAG_OBS_F16 / _BF16 the same values as 16-bit floats (e/16, e <= 31, is exact in
 *                 all three formats) -- half the bytes of the output that bounds the trainer-facing path
 * [replaces: twentyfortyeight/core/gameboard.py:291-325 next_state, :328-348 is_done;
 *            twentyfortyeight/envs/twentyfortyeight.py:106-133 step;
 *            reinforce/game/core.py:234-272 next_state, :275-321 legal_actions_mask/is_done,
 *            :346-377 encode_observation; reinforce/game/env.py:79-133 step, :193-245 batched_*]
 */
ENV2048_EXPORT int env2048_step(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards,
                 const uint8_t *actions, int32_t *reward, uint8_t *done, uint8_t *legal, void *obs,
                 uint64_t seed, uint64_t step_index, uint32_t flags);

/*
 * Several steps enqueued by ONE call on `stream`, with the RESULT of executing the records in order -- for callers whose
 * per-call host cost (argument marshalling, device selection) would otherwise rival the 8-microsecond kernel: replaying
 * staged actions, stepping several independent batches, rotating buffer sets.  Every record is one env2048_step call with
 * the same meaning and the same checks; all records are validated before the first launch, so a failing call has enqueued
 * nothing.  `reserved` must be 0.
 *
 * Independent records share a launch.  Two records depend on each other iff one WRITES bytes the other reads or writes
 * (boards: read + written; actions: read; reward / done / legal / obs / max_exp: written).  A run of consecutive records
 * that are pairwise independent -- disjoint batches -- yields the same bytes in any order, so the library enqueues it as
 * ONE kernel launch whose workgroups walk the records one after the other (at most 32 records per launch; the records must
 * also want the same outputs kind -- plain vs legal/obs/max_exp/normalised reward --, the same action source and the same
 * tiling hint, sizes within 2x of each other, and none may use `step_counter`).  No kernel boundary separates those
 * batches, so the stores of one overlap the loads of the next: batches that do not live in the 256 MiB Infinity Cache
 * run at the memory floor of the access pattern instead of paying a pipeline fill and drain per launch, and small batches
 * stop being bound by the launch rate.  Dependent records (the same boards stepped K times) are launched one by one, in
 * order, as before.  The decision depends on the records alone (addresses, sizes, flags); env2048_step_many_plan reports
 * it.  Tiling and store policy of a shared launch follow the size rules of env2048_step applied to the launch's TOTAL
 * board count.  A record with n == 0 is a no-op wherever it stands -- any number of them, leading, trailing or between
 * live records: it enqueues nothing and neither joins nor separates its neighbours (only its flags / reserved field are
 * validated); planning is one pass over the records and terminates on every input.
 */
typedef struct env2048_step_args {
    uint64_t n, env_base;
    uint8_t *boards;
    const uint8_t *actions;
    int32_t *reward;
    uint8_t *done;
    uint8_t *legal;
    void *obs;
    uint64_t seed, step_index;
    uint32_t flags, reserved;
    uint8_t *max_exp; /* optional, records only: max_exp[i] = log2(max tile) of board i after move + spawn and BEFORE an
                         auto-reset replaces it -- for a game that ends in this step, the max tile it reached
                         (info.max_tile, reinforce/game/env.py:127-131; Trajectory.max_tile, self_play.py:389) */
    uint64_t *step_counter; /* optional, records only: a device uint64.  When given, the launch's step index is
                               *step_counter + step_index (the field above becomes an OFFSET).  A hipGraph that captured K
                               such launches with offsets 0 .. K-1 followed by env2048_advance_counter(counter, K) can be
                               REPLAYED: every replay continues the random stream where the previous one stopped -- the
                               launch-bound per-move loop (reinforce/training/self_play.py:314-368) then costs 2-3 us per
                               step instead of the 3-4 us of host launch cost. */
} env2048_step_args;
ENV2048_EXPORT int env2048_step_many(int device, void *stream, uint64_t count, const env2048_step_args *args);

/*
 * How env2048_step would run a launch of n boards with these flags: boards_per_lane (1 = the small tiling, 4 = the large
 * one) and write_through (1 = write-through stores, 0 = write-back), each may be NULL.  No device is touched.
 * [no reference counterpart: launch policy of this build, exposed so that callers and benchmarks need not restate it]
 */
ENV2048_EXPORT int env2048_step_plan(uint64_t n, uint32_t flags, int32_t *boards_per_lane, int32_t *write_through);

/*
 * How env2048_step_many would run these records: launch_of[k] = index of the kernel launch record k goes out with
 * (records with equal values share one launch; required, `count` entries), and per record the tiling / store policy of
 * that launch (each may be NULL).  Returns the number of launches (>= 0) or a negative error code; validates the records
 * exactly like env2048_step_many and touches no device.
 */
ENV2048_EXPORT int env2048_step_many_plan(uint64_t count, const env2048_step_args *args, uint32_t *launch_of, int32_t *boards_per_lane,
                                          int32_t *write_through);
/* *counter += delta on `stream` (one thread): the node that ends a captured graph of step_counter launches. */
ENV2048_EXPORT int env2048_advance_counter(int device, void *stream, uint64_t *counter, uint64_t delta);

/* How env2048_step_sequence would run: chains (1 or 2 half-batch chains), and tiling / store policy of its launches (each may be
 * NULL).  No device is touched. */
ENV2048_EXPORT int env2048_step_sequence_plan(uint64_t n, uint32_t count, uint32_t flags, int32_t *chains, int32_t *boards_per_lane,
                                              int32_t *write_through);

/*
 * `count` consecutive transitions of ONE batch: launch t (t = 0 .. count-1) is
 *     env2048_step(device, stream, n, env_base, boards, actions[t], reward + t*row_stride, done + t*row_stride,
 *                  legal + t*row_stride, obs + 16*t*row_stride elements, seed, step_index + t, flags)
 * `actions` is a HOST array of `count` device pointers (NULL with RANDOM_ACTIONS).  row_stride = 0 makes every
 * step overwrite the same output arrays; row_stride >= n records step t in row t of [count, row_stride] arrays
 * (the per-step rows of reinforce/training/self_play.py:359-365 without a host round trip).
 *
 * Two chains.  The environments of a batch do not depend on each other, so for 2^20 <= n <= 2^23 and count >= 32 (and unless
 * `stream` is being captured or ENV2048_FLAG_SINGLE_CHAIN is given) the batch is cut in two halves that are stepped as two
 * independent chains of `count` launches each: the first on `stream`, enqueued by the calling thread, the second on an internal
 * stream of the device, enqueued concurrently by a helper thread of the library (one per device, created on first use).  The
 * second chain starts behind everything already on `stream` and `stream` waits for it before anything enqueued after the call,
 * so the call keeps the semantics of `count` launches on `stream`; the halves slide against each other on the GPU (one half's
 * load phase and store drain under the other's arithmetic), which a single chain of single-generation launches cannot do:
 * 6.5-6.7 us instead of 7.7-7.9 us per 2^20-board step in long sequences (count = 50: -10 %, 20: break-even; smaller batches and
 * batches beyond the Infinity Cache lose, hence the bounds).  Results are bit-identical.  env2048_step_sequence_plan reports
 * the decision.
 *
 * The direct route.  A sequence of count >= 8 steps (ENV2048_DIRECT_MIN_STEPS) over n <= 1.5 x 2^20 boards (with none of NO_DIRECT,
 * STORES_WRITE_THROUGH, TILE_* among its flags; any outputs), on a stream that is not being captured, does not go
 * through the HIP runtime at all: the library writes the `count` kernel-dispatch packets of each chain (two half-batch chains
 * from 2^18 boards, else one) into user-mode queues of its own -- created on a device's first such call together with a second
 * copy of the step kernels loaded through ROCr -- from the calling thread alone (0.13-0.25 us of host time per packet: no
 * helper thread).  Only the first packet of a chain carries an acquire fence and only the last a release fence (system scope);
 * between them the boards never leave the XCD L2 their workgroups run on: block k owns the same boards in every launch, and a
 * queue deals workgroup k of every dispatch to the same XCD (seen to change once, around the start-up of the process's queues;
 * never in 24 000 fuzzed sequences since).  That holds in practice, not by contract, so it is CHECKED: the first launch records every workgroup's XCD (and copies the
 * boards it loads into a snapshot buffer of the library, 16 n bytes of device memory kept per device), every later launch
 * compares; after one mismatch anywhere in the sequence the call puts the snapshot back and runs the sequence again through
 * the stream route, so the caller gets the same bytes either way (env2048_direct_stats counts the reruns).  5.3-5.5 us per
 * 2^20-board step at count = 200 and 6.3-6.5 at count = 20, where the stream route needs 7.0-8.0 and 9.0-9.2.  Results are
 * bit-identical.
 *   SEMANTICS ON THIS ROUTE: the call first waits until everything enqueued on `stream` has completed (hipStreamSynchronize if
 *   the stream is busy), and RETURNS WHEN THE SEQUENCE HAS COMPLETED -- every output is in memory, visible to the host, to
 *   `stream` and to any other stream.  That is stronger than the asynchronous contract above, not different from it for a
 *   caller that synchronises anyway; a caller that wants the call to return at once passes ENV2048_FLAG_NO_DIRECT.  The
 *   calling thread polls for completion (about 8 us per step), then sleeps in the runtime.  Calls on one device are
 *   serialised.  The route switches itself off (and env2048_direct_info says why) where it cannot run: no large-BAR access to
 *   device memory, a device that is not gfx950 with 8 XCDs, ENV2048_DIRECT=0 in the environment.
 * [replaces: the env.step calls of the move loop, reinforce/training/self_play.py:314-368]
 */
ENV2048_EXPORT int env2048_step_sequence(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards,
                                         const uint8_t *const *actions, int32_t *reward, uint8_t *done, uint8_t *legal, void *obs,
                                         uint64_t row_stride, uint64_t seed, uint64_t step_index, uint32_t count, uint32_t flags);

/*
 * The helper thread (what a maintainer embedding the library needs to know).
 *   - Exists only after the first env2048_step_sequence call on a device that ran as two chains; a process that never makes
 *     such a call (or passes ENV2048_FLAG_SINGLE_CHAIN every time) has no extra thread, stream or event.  One per device,
 *     detached, never joined; a child created by fork() does not inherit it and creates its own on its first split call.
 *   - What it does: enqueues the second chain's `count` launches on the internal stream while the caller's thread enqueues
 *     the first chain.  It never synchronises with the GPU and touches none of the caller's memory.
 *   - Poll window: after a job it polls (a `pause` loop on one atomic flag) for the next one for ENV2048_HELPER_SPIN_US
 *     microseconds -- an environment variable read once when the process's first helper is created; default 300, 0 = go to
 *     sleep at once, capped at 100 000 -- and then sleeps on a condition variable.  While polling it occupies one hardware
 *     thread: a stepping loop that calls every few microseconds keeps it busy all the time (one core per device in use),
 *     an idle process pays nothing after the window.  With 0 every call pays a futex wake-up (tens of microseconds) instead,
 *     which only long sequences amortise.  CPU-pinned callers (one core per worker) should set 0 or SINGLE_CHAIN.
 *   - Concurrency: split calls on one device are serialised by a mutex held from posting the second chain until it has
 *     been enqueued, so two threads stepping different batches on the same device are served one after the other (each
 *     call still returns as soon as its own launches are enqueued); calls on different devices do not interact.
 *   - env2048_helper_stats reports what the helper of `device` has cost so far: created (0/1), spin_us (the window in
 *     force), jobs served, and cpu_ns = thread CPU time consumed (CLOCK_THREAD_CPUTIME_ID, sampled after every job and
 *     poll window).  It makes no device call.  bench.py prints it for the K = 200 leg.
 *
 * Two more environment switches exist for A/B measurements only (read once per process, no effect on results):
 * ENV2048_EXP_STREAM_STORES=1 / 0 forces non-temporal / plain stores in the write-stream kernels that otherwise choose by
 * output size (env2048_chance_outcomes, env2048_encode_onehot, episode_of of env2048_segment_episodes);
 * ENV2048_EXP_THROUGH=0 turns off the write-through stores of the other kernels' 16-byte outputs.
 */
typedef struct env2048_helper_info {
    uint32_t created, spin_us;
    uint64_t jobs, cpu_ns;
} env2048_helper_info;
ENV2048_EXPORT int env2048_helper_stats(int device, env2048_helper_info *out);

/*
 * The direct route of env2048_step_sequence on `device`: available (1 = ready; the call starts the route up if this is its
 * first use on the device -- two queues, 256 KiB of argument blocks), `reason` (why not, or what it found), and counters:
 * sequences run, their steps and packets, calls that were eligible by size and flags but had to take the stream route (a
 * captured stream), sequences that had to be run again on the stream route (see env2048_step_sequence), the number of chains
 * of the last sequence and its duration ON THE DEVICE (first packet's start to last packet's end, from the queues' dispatch
 * timestamps; 0 when ENV2048_DIRECT_TIMING=0).  env2048_direct_eligible answers, without touching a device, whether a call
 * of this shape would be offered to the route (`extras` is ignored: the route serves legal / obs / the normalised reward too).
 */
typedef struct env2048_direct_info {
    uint32_t available, last_chains;
    uint64_t calls, steps, packets, declined, reruns, last_gpu_ns;
    char reason[160];
} env2048_direct_info;
ENV2048_EXPORT int env2048_direct_stats(int device, env2048_direct_info *out);
ENV2048_EXPORT int env2048_direct_eligible(uint64_t n, uint32_t count, uint32_t flags, int32_t extras);
/* How the direct route would run a sequence (no device is touched): eligible (as above); the chains (1 or 2) and for each the
 * boards it steps -- part c = [part_first[c], part_first[c + 1]) -- with the workgroups and boards per lane of its launches; and
 * the segments the argument blocks are written in (the first is one step, the rest at most segment_steps). */
typedef struct env2048_direct_plan_info {
    uint32_t eligible, chains;
    uint64_t part_first[3];
    uint32_t part_blocks[2], part_boards_per_lane[2];
    uint32_t segments, segment_steps;
} env2048_direct_plan_info;
ENV2048_EXPORT int env2048_direct_plan(uint64_t n, uint32_t count, uint32_t flags, int32_t extras, env2048_direct_plan_info *out);

/*
 * The reinforce/game form of one transition, fused: boards are int32 tile VALUES (n x 16, in
 * place) as in GameState.board; semantics are those of reinforce/game/env.py:79-133 --
 *     if done_io[i]:  nothing changes, reward[i] = 0, done stays 1           (done-guard, :103-104)
 *     else:           board = next_state(board, action) with the step word of (env_base+i, step_index),
 *                     reward[i] = float32 score (0 for an invalid move), step_count[i] += 1 (also on
 *                     invalid moves, :116), total_reward[i] += reward[i],
 *                     done_io[i] = "no action changes the board" (the JAX is_done, core.py:320-321)
 *     max_tile[i] = largest tile value of the board written back (info dict, :127-131)
 * actions are int32 (taken mod 4).  step_count, total_reward, max_tile may be NULL.
 * [replaces: reinforce/game/env.py:79-133 step, :193-211 batched_step]
 */
ENV2048_EXPORT int env2048_step_state_i32(int device, void *stream, uint64_t n, uint64_t env_base, int32_t *boards,
                                          const int32_t *actions, int32_t *step_count, uint8_t *done_io, float *total_reward,
                                          float *reward, int32_t *max_tile, uint64_t seed, uint64_t step_index);

/*
 * The same transition written to NEW arrays -- the functional form the reference API has (batched_step
 * returns a new GameState and leaves its argument alone, reinforce/game/env.py:193-211): every `*_in` array
 * is only read, every `*_out` array is written for all n environments (a finished game is copied through).
 * Passing the same pointer for an in/out pair gives the in-place call above.  step_count and total_reward
 * need both of their arrays or neither.
 * [replaces: reinforce/game/env.py:79-133 step, :193-211 batched_step]
 */
ENV2048_EXPORT int env2048_step_state_i32_to(int device, void *stream, uint64_t n, uint64_t env_base, const int32_t *boards_in,
                                             int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                             int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out,
                                             const float *total_reward_in, float *total_reward_out, float *reward,
                                             int32_t *max_tile, uint64_t seed, uint64_t step_index);

/*
 * The same transition with ONE PRNG KEY PER ENVIRONMENT -- what jax.vmap(step)(states, actions, keys) means in the
 * reference (reinforce/game/env.py:193-211; the caller splits a fresh key for every environment and move,
 * reinforce/training/self_play.py:351-354): keys is n x 2 uint32, and environment i draws exactly what the scalar
 * form draws for a batch of one with seed = (keys[i][0] << 32) | keys[i][1], env_base 0, step_index 0.  Row i of
 * the result therefore equals the single-environment step with keys[i], and it does not depend on i or on the
 * other rows.
 * [replaces: reinforce/game/env.py:79-133 step under jax.vmap, :193-211 batched_step]
 */
ENV2048_EXPORT int env2048_step_state_i32_keyed(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                                int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                                int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out,
                                                const float *total_reward_in, float *total_reward_out, float *reward,
                                                int32_t *max_tile);

/*
 * Fresh int32 VALUE boards (n x 16), one PRNG key per environment: board i = the board env2048_reset gives a batch of
 * one with seed = (keys[i][0] << 32) | keys[i][1], env_base 0, step_index 0 (zeros + two sequential spawns).
 * [replaces: reinforce/game/core.py:324-343 create_initial_board under jax.vmap; reinforce/game/env.py:176-190 batched_reset]
 */
ENV2048_EXPORT int env2048_reset_keyed_i32(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards);

/*
 * The two keyed entry points above with what the caller asks about the NEW state right after them, written by the SAME
 * launch: obs (n x 16 float32, log2(tile)/16) and legal4 (n x 4 bytes, 0/1 per action: a bool array) of the boards the
 * call leaves behind (a frozen game's unchanged board included); each may be NULL.  The self-play loop asks the
 * environment for exactly these between two moves (reinforce/training/self_play.py:316-317 after :354 / :300), so a move
 * costs ONE launch instead of three.  [replaces: batched_step + batched_get_observation + batched_get_legal_actions,
 * reinforce/game/env.py:193-245]
 */
ENV2048_EXPORT int env2048_step_state_i32_keyed_status(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                                       int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                                       int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out,
                                                       const float *total_reward_in, float *total_reward_out, float *reward,
                                                       int32_t *max_tile, float *obs, uint8_t *legal4);
ENV2048_EXPORT int env2048_reset_keyed_i32_status(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards, float *obs,
                                                  uint8_t *legal4);

/*
 * The same two calls with a HOST MIRROR of the results, for callers that read every state back (the reference's loop does:
 * np.array(observations), np.array(legal_masks), np.array(rewards), np.array(dones), reinforce/training/self_play.py:346-356).
 * The caller keeps ALL output arrays of a call in one block of device memory and owns a block of pinned, device-accessible
 * host memory of the same layout; host_mirror_delta = (host block address) - (device block address), a multiple of 16.
 * Every output element the kernel writes at address p it also writes at p + host_mirror_delta, so once the launch has
 * completed (one stream synchronisation) the host block holds the results: no device-to-host copy, no second
 * synchronisation per array.  A few hundred bytes cross the bus at the reference's batch sizes (N = 2 ... 8); meant for
 * small batches.  host_mirror_delta = 0: no mirror (then identical to the _status forms).
 * env2048_reset_state_i32_keyed also writes the bookkeeping fields of a fresh GameState (step_count 0, done 0,
 * total_reward 0.0; each pointer may be NULL) so that a reset is ONE launch and lands in the mirror whole.
 * env2048_step_state_i32_keyed_mirrored requires the out-of-place form (boards_in != boards_out).
 * [replaces: batched_reset / batched_step + the caller's np.array(...) read-backs, reinforce/game/env.py:176-211,
 *  reinforce/training/self_play.py:298-300,346-356]
 */
ENV2048_EXPORT int env2048_reset_state_i32_keyed(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards,
                                                 int32_t *step_count, uint8_t *done, float *total_reward, float *obs, uint8_t *legal4,
                                                 int64_t host_mirror_delta);
ENV2048_EXPORT int env2048_step_state_i32_keyed_mirrored(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                                         int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                                         int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out,
                                                         const float *total_reward_in, float *total_reward_out, float *reward,
                                                         int32_t *max_tile, float *obs, uint8_t *legal4, int64_t host_mirror_delta);

/*
 * The mirrored calls with a COMPLETION WORD, for launches of at most ENV2048_SIGNAL_MAX_ENVS environments (one workgroup):
 * host_sequence is a device-accessible, 8-byte aligned word of PINNED HOST memory owned by the caller (typically 16 bytes
 * reserved at the end of the host mirror block).  After every result store of the launch -- to the device block and to the
 * host mirror -- has been performed at system scope, the kernel stores `sequence_value` there (release, system scope) and
 * ends.  The caller hands out increasing values (a launch serial) and, instead of synchronising the stream -- an
 * interrupt-driven wake-up of 20 us and more around a 4 us kernel -- polls the word: once it reads `sequence_value`, the host
 * mirror holds the results of that launch.  The kernel terminates by itself in every case; nothing on the GPU waits for the
 * host; a caller that never looks at the word loses nothing.  host_sequence == NULL: exactly the _mirrored form.
 * Errors: host_sequence without a mirror (host_mirror_delta == 0), misaligned, or n > ENV2048_SIGNAL_MAX_ENVS ->
 * ENV2048_ERR_INVALID_ARG / _ALIGNMENT, nothing launched (larger batches keep the stream-synchronisation path).
 * env2048_wait_sequence is the host side: poll *host_sequence until it equals `value` (returns 1) or timeout_us microseconds
 * have passed (returns 0: the caller falls back to synchronising the stream, so a lost launch can never hang the host);
 * negative on a NULL / misaligned pointer.  It makes no device call and holds no lock.
 * [replaces: batched_reset / batched_step + the caller's np.array(...) read-backs, reinforce/game/env.py:176-211,
 *  reinforce/training/self_play.py:298-300,346-356]
 */
#define ENV2048_SIGNAL_MAX_ENVS 512
ENV2048_EXPORT int env2048_reset_state_i32_keyed_signalled(int device, void *stream, uint64_t n, const uint32_t *keys, int32_t *boards,
                                                           int32_t *step_count, uint8_t *done, float *total_reward, float *obs,
                                                           uint8_t *legal4, int64_t host_mirror_delta, uint64_t *host_sequence,
                                                           uint64_t sequence_value);
ENV2048_EXPORT int env2048_step_state_i32_keyed_signalled(int device, void *stream, uint64_t n, const uint32_t *keys, const int32_t *boards_in,
                                                          int32_t *boards_out, const int32_t *actions, const int32_t *step_count_in,
                                                          int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out,
                                                          const float *total_reward_in, float *total_reward_out, float *reward,
                                                          int32_t *max_tile, float *obs, uint8_t *legal4, int64_t host_mirror_delta,
                                                          uint64_t *host_sequence, uint64_t sequence_value);
ENV2048_EXPORT int env2048_wait_sequence(const uint64_t *host_sequence, uint64_t value, uint32_t timeout_us);
/* The same completion word for ANY work: the device stores `value` into the pinned host word once everything enqueued on
 * `stream` before this call has completed (one stream memory operation, no kernel); poll it with env2048_wait_sequence.
 * What the single-board API mirrors do instead of synchronising the stream after their one launch
 * (twentyfortyeight/envs/twentyfortyeight.py:106-133 step, twentyfortyeight/core/gameboard.py, gamemove.py). */
ENV2048_EXPORT int env2048_stream_signal(int device, void *stream, uint64_t *host_word, uint64_t value);

/*
 * env2048_step_state_i32_keyed_signalled for at most ENV2048_INLINE_MAX_ENVS environments with the per-move INPUTS TAKEN FROM
 * ORDINARY HOST MEMORY: host_keys (n x 2 uint32) and host_actions (n int32) are host pointers -- pageable memory is fine --
 * that the library reads DURING THE CALL and sends along in the kernel-argument block; the caller may overwrite them as soon
 * as the call returns, and nothing needs to be staged in pinned or device memory.  This is the shape of the reference's move
 * loop: keys and actions are produced on the host for every move (reinforce/training/self_play.py:333-354, N = 2 ... 8 games).
 * All other pointers are device pointers as before; host_mirror_delta / host_sequence as in the _signalled form (both
 * optional).  Results are those of env2048_step_state_i32_keyed on the same keys and actions.
 * [replaces: batched_step with host-side keys / actions, reinforce/game/env.py:193-211]
 */
#define ENV2048_INLINE_MAX_ENVS 64
ENV2048_EXPORT int env2048_step_state_i32_keyed_inline(int device, void *stream, uint64_t n, const uint32_t *host_keys, const int32_t *boards_in,
                                                       int32_t *boards_out, const int32_t *host_actions, const int32_t *step_count_in,
                                                       int32_t *step_count_out, const uint8_t *done_in, uint8_t *done_out,
                                                       const float *total_reward_in, float *total_reward_out, float *reward,
                                                       int32_t *max_tile, float *obs, uint8_t *legal4, int64_t host_mirror_delta,
                                                       uint64_t *host_sequence, uint64_t sequence_value);

/*
 * Per-move queries of the reinforce/game callers on int32 tile VALUE boards (n x 16), one launch, each
 * output optional:
 *     obs[16*i..]    float32 log2(tile)/16, 0 for empty      (encode_observation, core.py:346-377)
 *     legal4[4*i+a]  1 iff action a changes board i, else 0  (legal_actions_mask, core.py:275-302)
 *     done[i]        1 iff no action changes the board       (the JAX is_done, core.py:305-321)
 * [replaces: reinforce/game/env.py:214-245 batched_get_observation / batched_get_legal_actions]
 */
ENV2048_EXPORT int env2048_status_i32(int device, void *stream, uint64_t n, const int32_t *boards, float *obs, uint8_t *legal4,
                                      uint8_t *done);
/*
 * Afterstates on int32 tile VALUE boards (n x 16), one launch: slide + merge towards `actions[i]` -- or towards `action` for
 * every board when `actions` is NULL -- with no spawn; boards_out may be boards_in.  reward[i] (optional) = the merge score,
 * changed[i] (optional) = 1 iff the move changes board i.
 * [replaces: reinforce/game/core.py:129-170 latent_state, :104-126 slide_and_merge (action 0), :43-101 merge_row (a row in an
 *            otherwise empty board)]
 */
ENV2048_EXPORT int env2048_afterstate_i32(int device, void *stream, uint64_t n, const int32_t *boards_in, const int32_t *actions,
                                          uint32_t action, int32_t *boards_out, int32_t *reward, uint8_t *changed);

/*
 * T consecutive transitions per board, boards held in registers, actions drawn on device
 * (low two bits of each step word): identical to T calls of env2048_step(..., step0 + t,
 * flags | RANDOM_ACTIONS).  Per-environment outputs (each may be NULL):
 *     return_sum[i]  int64 sum of rewards      episodes[i]  number of transitions with done = 1
 *     valid_moves[i] number of transitions that changed the board
 * [replaces: the env side of reinforce/training/self_play.py:314-368 play_parallel_games loop]
 */
ENV2048_EXPORT int env2048_rollout(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                    uint64_t step0, uint32_t T, int64_t *return_sum, uint32_t *episodes, uint32_t *valid_moves,
                    uint32_t flags);

/*
 * T consecutive transitions with PRE-STAGED actions in one launch, boards held in registers: actions, reward and done
 * are [T, row_stride] row arrays and the result is identical to
 *     env2048_step_sequence(device, stream, n, env_base, boards, {actions + t*row_stride}, reward, done, NULL, NULL,
 *                           row_stride, seed, step0, T, flags)
 * (reward / done may be NULL; flags: AUTO_RESET only).  Per board-step it moves 6 bytes (action in, reward + done out)
 * instead of the 38 of a step launch and pays no launch: for replaying recorded games, verifying stored trajectories and
 * scoring open-loop action sequences against the true environment.  return_sum / episodes / valid_moves as in
 * env2048_rollout (each may be NULL).
 * [replaces: the env.step calls of the move loop, reinforce/training/self_play.py:314-368, when the actions are known]
 */
ENV2048_EXPORT int env2048_replay(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions,
                                  int32_t *reward, uint8_t *done, uint64_t row_stride, uint64_t seed, uint64_t step0, uint32_t T,
                                  uint32_t flags, int64_t *return_sum, uint32_t *episodes, uint32_t *valid_moves);

/*
 * The same rollout reporting per-EPISODE statistics, accumulated on device -- what evaluate_games computes from a list
 * of trajectories (reinforce/training/self_play.py:503-521: mean/std/min/max of the total reward, mean and max of the
 * max tile, mean length, "reached 2048/4096/..." counts) for millions of games without moving a trajectory.
 * stats: ENV2048_ROLLOUT_STATS_WORDS uint64 words that the call ADDS to (so several calls / slices / GPUs accumulate;
 * the caller zeroes them first and sets word 4 to UINT64_MAX):
 *     [0] finished games   [1] sum of their lengths (moves, invalid ones included)   [2] sum of their total rewards
 *     [3] sum of squared total rewards   [4] min total reward   [5] max total reward   [6] sum of squared lengths
 *     [7] unused   [8 + e] finished games whose largest tile was 2^e  (e = 0 .. 31)
 * ep_length / ep_return (each may be NULL): the length / reward so far of the game every environment is IN, read at the
 * start and written at the end, so that games spanning several calls are counted whole (NULL: games in progress at the
 * start of the call count from that point).
 * [replaces: evaluate_games, reinforce/training/self_play.py:476-521, on random-action rollouts]
 */
#define ENV2048_ROLLOUT_STATS_WORDS 40
/* Without AUTO_RESET a finished game stays on its board: it is counted once, and its ep_length becomes this marker (its
 * ep_return keeps the game's total), so that further calls leave it alone. */
#define ENV2048_EPISODE_OVER 0xFFFFFFFFu
ENV2048_EXPORT int env2048_rollout_episodes(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                                            uint64_t step0, uint32_t T, uint32_t flags, uint32_t *ep_length, int64_t *ep_return,
                                            uint64_t *stats);

/*
 * Episodes out of recorded step rows -- the device form of reinforce/training/self_play.py:359-392 (per-game lists,
 * "store only for still-active games", Trajectory(..., done, total_reward, max_tile)).  Input: the [*, row_stride]
 * row arrays env2048_step / env2048_step_sequence / env2048_step_many wrote, one row per step: `done` (required),
 * `reward` (int32 scores, or float32 with ENV2048_EPISODES_REWARD_F32; may be NULL) and `max_exp` (may be NULL).
 * `rows` logical rows are considered, logical row r living in physical row (first_row + r) % ring_rows (a plain
 * [T, N] recording: first_row = 0, ring_rows = rows = T).  An episode of environment i is a maximal run of rows ending
 * with done = 1; a run cut by the end of the recording is reported with terminated = 0.
 *   ENV2048_EPISODES_FROZEN  the recording was made WITHOUT auto-reset (the reference's loop): a finished game stays
 *                            finished, so only its first episode exists and later rows belong to no episode
 * Two calls: env2048_count_episodes gives counts[i] = episodes of environment i; the caller forms the exclusive prefix
 * sum `offsets` (and its total = number of records) and env2048_segment_episodes writes the records of environment i at
 * episodes[offsets[i] ...] in time order -- so the output order is deterministic: by environment, then by start.
 * episode_of (may be NULL; [*, row_stride] int32, same physical layout as the rows): index of the record a row belongs
 * to, -1 for rows after the end of a frozen game.
 */
#define ENV2048_EPISODES_FROZEN 1u
#define ENV2048_EPISODES_REWARD_F32 2u
typedef struct env2048_episode {
    uint32_t env;       /* column of the recording (environment index within the batch) */
    uint32_t start;     /* first logical row */
    uint32_t length;    /* rows = moves of the episode, invalid ones included (env.py:116 counts them too) */
    uint8_t terminated; /* 1: ended by done (Trajectory.done); 0: cut by the end of the recording */
    uint8_t max_exp;    /* log2(max tile) reached -- max over the episode's max_exp rows; 0 without them */
    uint16_t reserved;
    double total_reward; /* sum of the episode's reward rows (exact for int32 scores) */
} env2048_episode;
ENV2048_EXPORT int env2048_count_episodes(int device, void *stream, uint64_t n, uint64_t row_stride, uint32_t rows, uint32_t first_row,
                                          uint32_t ring_rows, const uint8_t *done, uint32_t flags, uint32_t *counts);
ENV2048_EXPORT int env2048_segment_episodes(int device, void *stream, uint64_t n, uint64_t row_stride, uint32_t rows,
                                            uint32_t first_row, uint32_t ring_rows, const uint8_t *done, const void *reward,
                                            const uint8_t *max_exp, uint32_t flags, const uint64_t *offsets,
                                            env2048_episode *episodes, int32_t *episode_of);

/*
 * Afterstate only (deterministic half of a move): boards_out = slide+merge(boards_in, action),
 * reward = merge score, changed = (boards_out != boards_in).  boards_out may alias boards_in.
 * [replaces: twentyfortyeight/core/gameboard.py:23-129 merge_column/slide_and_merge/latent_state;
 *            twentyfortyeight/core/gamemove.py:132-164 can_move (== changed);
 *            reinforce/game/core.py:32-170 merge_row/slide_and_merge/latent_state]
 */
ENV2048_EXPORT int env2048_afterstate(int device, void *stream, uint64_t n, const uint8_t *boards_in, const uint8_t *actions,
                       uint8_t *boards_out, int32_t *reward, uint8_t *changed);

/*
 * Afterstates of all four actions of every board in one launch -- the expansion step of a tree search over
 * the true environment: out_boards[4*i + a] = slide+merge(boards[i], a), out_reward[4*i + a] its merge score,
 * out_legal[i] bit a set <=> action a changes the board (out_reward / out_legal may be NULL).  Together with
 * env2048_chance_outcomes on the 4n afterstates this enumerates a full expectimax ply.
 * [replaces: one latent_state call per (board, action), twentyfortyeight/core/gameboard.py:105-129, and
 *            legal_actions_mask, gamemove.py:45-83]
 */
ENV2048_EXPORT int env2048_afterstates4(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *out_boards,
                                        int32_t *out_reward, uint8_t *out_legal);

/*
 * Chance half of a move with the outcome given by the caller: put exponent spawn_e[i] into the
 * spawn_k[i]-th empty cell (row-major).  placed[i] (may be NULL) = flat cell index, or 0xFF when
 * spawn_k[i] >= number of empty cells (board unchanged).
 * [replaces: twentyfortyeight/core/gameboard.py:174-244 after_state_lazy + generate_outcome,
 *            :247-288 fill_cells (cell order); reinforce/game/core.py:173-231 fill_cells]
 */
ENV2048_EXPORT int env2048_spawn(int device, void *stream, uint64_t n, uint8_t *boards, const uint8_t *spawn_k,
                  const uint8_t *spawn_e, uint8_t *placed);

/*
 * Chance half of a move with the outcome drawn on device: `count` sequential single spawns per
 * board, spawn t of environment env_base + i using the step word of (env, step_index + t)
 * exactly as env2048_step does (uniform empty cell, tile 4 with probability 0.1).  A full board is
 * left unchanged.
 * [replaces: twentyfortyeight/core/gameboard.py:247-288 fill_cells(state, number_tile, seed);
 *            reinforce/game/core.py:173-231 fill_cells(state, key)]
 */
ENV2048_EXPORT int env2048_spawn_random(int device, void *stream, uint64_t n, uint64_t env_base, uint8_t *boards,
                                        uint64_t seed, uint64_t step_index, uint32_t count);

/*
 * The actions ENV2048_FLAG_RANDOM_ACTIONS makes env2048_step / env2048_rollout take at (seed, step_index), written out:
 * actions[i] = low two bits of the step word of environment env_base + i.  A step played with this array as `actions`
 * (without the flag) is bit for bit the step played with the flag -- recorders use it so that the action row of a
 * random-play step says what was played (Trajectory.actions, reinforce/training/self_play.py:361).
 */
ENV2048_EXPORT int env2048_random_actions(int device, void *stream, uint64_t n, uint64_t env_base, uint64_t seed, uint64_t step_index,
                                          uint8_t *actions);

/*
 * All chance outcomes of an afterstate: for every board the 32 slots s = 2*k + v (k = rank of an
 * empty cell in row-major order, v = 0 for tile 2, 1 for tile 4) hold the board with that tile
 * placed and its probability P(tile) / n_empty (0.9 / 0.1); slots with k >= n_empty hold a zero
 * board and probability 0.  A full board yields slot 0 = the board itself with probability 1,
 * like the reference.  out_boards: n x 32 x 16 bytes, out_prob: n x 32 floats, out_count (may be
 * NULL): number of valid slots per board (2 * n_empty, or 1 for a full board).
 * [replaces: twentyfortyeight/core/gameboard.py:132-171 after_state (same enumeration order),
 *            :174-244 after_state_lazy + generate_outcome]
 */
ENV2048_EXPORT int env2048_chance_outcomes(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *out_boards,
                                           float *out_prob, uint8_t *out_count);

/*
 * One full expectimax ply of the true environment, fused: for every board i and action a, the afterstate's merge score
 * after_reward[4*i + a] and -- if a is legal -- ALL chance outcomes of that afterstate in the 32 slots s = 2*k + v of
 * env2048_chance_outcomes (k = rank of the empty cell in row-major order, v = 0 tile 2 / 1 tile 4):
 *     succ_boards[((4*i + a)*32 + s) * 16 ..]  the successor board (zero board for an unused slot)
 *     succ_prob  [ (4*i + a)*32 + s ]          P(tile) / n_empty (0 for an unused slot)
 *     succ_status[ (4*i + a)*32 + s ]          bits 0-3: legal-action mask of the successor, bit 4: successor is terminal
 *     succ_count [ 4*i + a ]                   used slots = 2 * n_empty(afterstate); 0 for an ILLEGAL action (the board
 *                                              does not change, so there is nothing to enumerate: all 32 slots are zero)
 *     legal[i]                                 bit a set <=> action a legal on board i
 * = env2048_afterstates4 -> env2048_chance_outcomes (on the legal afterstates) -> env2048_status of every successor in one
 * launch, without the 4n afterstates ever being written to memory.  succ_status, succ_count, after_reward, legal may be NULL.
 * [replaces: per node of a search over the true environment, twentyfortyeight/core/gameboard.py:105-129 latent_state x 4,
 *            :132-171 after_state, :328-348 is_done and gamemove.py:45-83 legal_actions_mask on every outcome]
 */
ENV2048_EXPORT int env2048_expectimax_ply(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *succ_boards,
                                          float *succ_prob, uint8_t *succ_status, uint8_t *succ_count, int32_t *after_reward,
                                          uint8_t *legal);

/*
 * Depth-2 expectimax of the true environment -- the whole two-ply tree of every board in one launch, nothing of it in
 * memory (the search a caller would build from latent_state + after_state + legal_actions, gameboard.py:105-171,
 * gamemove.py:45-83):
 *     Q2(s, a) = r(s, a) + sum over chance outcomes o of afterstate(s, a):  P(o) * max over legal a' of r(o, a')
 *     V2(s)    = max over legal a of Q2(s, a) (0 for a terminal board);  best = the smallest maximising a (255: terminal)
 * with P(o) = {0.9, 0.1}[tile] / num_empty (after_state) and r the merge score of latent_state; a successor with no legal
 * move contributes 0.  Outputs (each may be NULL, at least one must not be):
 *     terms        n x 4 x 4 int32, per (board, action): {r(s,a), S2, S4, num_empty} -- S2 / S4 = the sum over the empty
 *                  cells of the afterstate of max_a' r(o, a') with a 2 / a 4 spawned there; all 0 for an illegal action.
 *                  EXACT integers; Q2 = r + (0.9 * S2 + 0.1 * S4) / num_empty = (10 n r + 9 S2 + S4) / (10 n)
 *     q            n x 4 float32: that fraction as ONE double division, rounded to float32 (-inf for an illegal action);
 *                  the arg-max compares the fractions exactly (integers), so ties are decided by the action index alone
 *     best_action  n bytes        value  n float32 (V2)
 * [no single reference counterpart: composes gameboard.py:105-129 latent_state, :132-171 after_state, gamemove.py:45-83]
 */
ENV2048_EXPORT int env2048_expectimax_depth2(int device, void *stream, uint64_t n, const uint8_t *boards, int32_t *terms, float *q,
                                             uint8_t *best_action, float *value);

/*
 * Depth-d expectimax, d >= 3: the building blocks of a search tree that stays on the device as one DENSE frontier per level.
 *     V_d(s) = max over legal a of Q_d(s, a);   Q_d(s, a) = r(s, a) + (0.9 * S2 + 0.1 * S4) / num_empty,
 *     S2 / S4 = the sums of V_{d-1} over the chance outcomes of afterstate(s, a) that spawned a 2 / a 4  (0 for a terminal board)
 * [composes gameboard.py:105-129 latent_state, :132-171 after_state, gamemove.py:132-164 can_move, like the depth-2 call].
 * The CHILDREN of a board are those outcomes for its legal actions, in the order (action, empty cell in row-major rank, tile 2
 * then 4): on mid-game boards about 34 of the 128 possible.
 *   env2048_expand_count   counts[i] = number of children of board i.
 *   env2048_expand_offsets the offsets env2048_expand needs, computed on the device: offsets[i] = number of children of the
 *                          boards before i (count, block scan, scan of the block totals, add -- three small launches enqueued
 *                          by this call).  `scratch` is caller-owned device memory of env2048_expand_scratch_words(n) =
 *                          ceil(n / ENV2048_EXPAND_OFFSETS_BLOCK) + 1 uint64; when the call has completed its LAST word holds the
 *                          TOTAL number of children.  A host that must size
 *                          the next frontier passes `host_total`: two uint64 of PINNED (device-accessible) host memory --
 *                          the device writes the total into word 0 and then `sequence_value` into word 1 (release, system
 *                          scope), so env2048_wait_sequence(host_total + 1, sequence_value, ...) replaces a stream
 *                          synchronisation + device-to-host copy per level of the tree.  host_total may be NULL.
 *   env2048_expand         writes them densely: offsets = the exclusive prefix sums of counts (caller-computed), children =
 *                          (sum of counts) x 16 bytes, plus per (board, action) the merge score reward[4i + a] (0 if illegal) and
 *                          n_empty[4i + a] = empty cells of the afterstate (0 = illegal action; a legal move always leaves one).
 *                          The children of (i, a) start at offsets[i] + 2 * (n_empty of the actions before a), cell k at + 2k.
 *                          (reward must be 16-byte, n_empty 4-byte aligned: a board's four entries are stored as one word.)
 *   env2048_expectimax_depth2_f64   V_2 of every board as float64 (the deepest frontier's values).
 *   env2048_expectimax_backup       folds child values into the values of a level: value[i] = V (float64), optionally q (n x 4
 *                          float64, -inf for an illegal action) and best_action (255: terminal).  Sums run in child order and
 *                          every operation is a separate correctly rounded double operation: results do not depend on the
 *                          device's contraction or scheduling.
 * Depth d = expand x (d - 2), depth2_f64 on the last frontier, backup x (d - 2) (simulate_2048_amd.ops.expectimax does that).
 */
ENV2048_EXPORT int env2048_expand_count(int device, void *stream, uint64_t n, const uint8_t *boards, uint32_t *counts);
#define ENV2048_EXPAND_OFFSETS_BLOCK 512
ENV2048_EXPORT uint64_t env2048_expand_scratch_words(uint64_t n);
ENV2048_EXPORT int env2048_expand_offsets(int device, void *stream, uint64_t n, const uint8_t *boards, uint64_t *offsets, uint64_t *scratch,
                                          uint64_t *host_total, uint64_t sequence_value);
ENV2048_EXPORT int env2048_expand(int device, void *stream, uint64_t n, const uint8_t *boards, const uint64_t *offsets, uint8_t *children,
                                  int32_t *reward, uint8_t *n_empty);
ENV2048_EXPORT int env2048_expectimax_depth2_f64(int device, void *stream, uint64_t n, const uint8_t *boards, double *value);
ENV2048_EXPORT int env2048_expectimax_backup(int device, void *stream, uint64_t n, const int32_t *reward, const uint8_t *n_empty,
                                             const uint64_t *offsets, const double *child_value, double *value, double *q,
                                             uint8_t *best_action);

/*
 * Boards of any size: S x S cells, 2 <= S <= ENV2048_NXN_MAX_SIZE, ONE BYTE PER CELL (the same log2 code), row-major,
 * n * S * S bytes per batch, no alignment requirement.  The reference's core functions take boards of any shape and its env
 * has a `size` argument (twentyfortyeight/envs/twentyfortyeight.py:21; its timing script runs sizes 4, 6 and 8,
 * tests/test_perf_utils.py:47); these entry points are what the API mirrors call for S != 4.  Same semantics, flags
 * (AUTO_RESET, RANDOM_ACTIONS only), action coding and RNG contract as the packed 4 x 4 entry points of the same name --
 * the step word of (seed, env_base + i, step_index) picks the spawn (cell rank over the board's empty cells in row-major
 * order), a fresh board takes two sequential spawns from the reset block with cell ranks floor(r0 * S*S / 2^32) and
 * floor(r2 * (S*S - 1) / 2^32) -- so for S = 4 they return exactly what env2048_afterstate / _step / _status / _reset /
 * _spawn_random / _chance_outcomes return.  One board per lane, the board in registers, the size a template parameter of
 * the kernels (csrc/nxn_ops.hpp); batches whose base address is 16-byte aligned move whole waves of boards as 16-byte vectors,
 * any other address works through a slower per-lane byte path (as does the ragged last wave of a batch).  Scores are int32:
 * exponents above 30 are outside the domain, as for the packed path.
 *     env2048_nxn_afterstate     latent_state: slide + merge, no spawn; reward, changed (== can_move(rot90(board, action)))
 *     env2048_nxn_step           next_state + is_done (+ legal mask of the board written back), in place
 *     env2048_nxn_status         is_done, legal-action bitmask, number of empty cells, log2(max tile)
 *     env2048_nxn_reset          zeros + two spawned tiles
 *     env2048_nxn_spawn_random   fill_cells(number_tile = count): `count` sequential single spawns (steps step_index ...)
 *     env2048_nxn_chance_outcomes  after_state: out_boards = n x (2 S S) slots of S S bytes, (cell in row-major rank, tile 2
 *                                then 4); out_count[i] = outcomes (1 = the board itself when it is full); unused slots zero
 * [replaces: twentyfortyeight/core/gameboard.py:23-129,132-171,247-348, gamemove.py:45-164,
 *            twentyfortyeight/envs/twentyfortyeight.py:84-133 for size != 4]
 */
#define ENV2048_NXN_MAX_SIZE 8
ENV2048_EXPORT int env2048_nxn_afterstate(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards_in, const uint8_t *actions,
                                          uint8_t *boards_out, int32_t *reward, uint8_t *changed);
ENV2048_EXPORT int env2048_nxn_step(int device, void *stream, uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards,
                                    const uint8_t *actions, int32_t *reward, uint8_t *done, uint8_t *legal, uint64_t seed, uint64_t step_index,
                                    uint32_t flags);
ENV2048_EXPORT int env2048_nxn_status(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *done, uint8_t *legal,
                                      uint8_t *n_empty, uint8_t *max_exp);
ENV2048_EXPORT int env2048_nxn_reset(int device, void *stream, uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, uint8_t *done,
                                     uint64_t seed, uint64_t step_index);
ENV2048_EXPORT int env2048_nxn_spawn_random(int device, void *stream, uint64_t n, uint32_t size, uint64_t env_base, uint8_t *boards, uint64_t seed,
                                            uint64_t step_index, uint32_t count);
ENV2048_EXPORT int env2048_nxn_chance_outcomes(int device, void *stream, uint64_t n, uint32_t size, const uint8_t *boards, uint8_t *out_boards,
                                               uint8_t *out_count);

/*
 * Read-only board status; every output may be NULL.
 *     done / legal / obs as in env2048_step;  max_exp[i] = log2(max tile);  n_empty[i] = #zeros
 * [replaces: twentyfortyeight/core/gameboard.py:328-348 is_done;
 *            twentyfortyeight/core/gamemove.py:45-129 legal_actions_mask/legal/illegal_actions;
 *            reinforce/game/core.py:275-321, :346-377, :427-460 max_tile/count_empty]
 */
ENV2048_EXPORT int env2048_status(int device, void *stream, uint64_t n, const uint8_t *boards, uint8_t *done, uint8_t *legal,
                   float *obs, uint8_t *max_exp, uint8_t *n_empty);

/*
 * One-hot observation: out[(16*i + cell) * width + e] = 1 for the exponent e of that cell (empty
 * cells light index 0), every other byte 0.  width must be in [1, 255]; exponents >= width are
 * undefined input.  width = 31 reproduces encode_flatten(state, encodage_size=31) as uint8.
 * [replaces: twentyfortyeight/utils/binary.py:11-86 encode/encode_flatten;
 *            twentyfortyeight/envs/twentyfortyeight.py:66-67 observation (encoded=True)]
 */
ENV2048_EXPORT int env2048_encode_onehot(int device, void *stream, uint64_t n, const uint8_t *boards, uint32_t width,
                                         uint8_t *out);

/*
 * Tile values <-> packed exponents, n boards of 16 cells.  pack: value v (0 or a power of two)
 * -> e = log2(v) (0 for 0); unpack: e -> 2^e (0 for 0).  int32 matches the reinforce/game board
 * dtype (reinforce/game/core.py:340), int64 the NumPy env's (twentyfortyeight.py:39,101).
 */
ENV2048_EXPORT int env2048_pack_i32(int device, void *stream, uint64_t n, const int32_t *values, uint8_t *boards);
ENV2048_EXPORT int env2048_unpack_i32(int device, void *stream, uint64_t n, const uint8_t *boards, int32_t *values);
ENV2048_EXPORT int env2048_pack_i64(int device, void *stream, uint64_t n, const int64_t *values, uint8_t *boards);
ENV2048_EXPORT int env2048_unpack_i64(int device, void *stream, uint64_t n, const uint8_t *boards, int64_t *values);

#ifdef __cplusplus
}
#endif
#endif /* ENV2048_H */
