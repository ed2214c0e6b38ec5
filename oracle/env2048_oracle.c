/*
 * env2048_oracle.c -- CPU ORACLE for the batched 2048 environment step.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT.  It is a scalar, reference-shaped
 * restatement of the algorithm in the reference's NumPy engine
 * (/root/reference/twentyfortyeight/core/gameboard.py, gamemove.py) in the packed
 * log2 domain used by the HIP path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; simulate_2048_amd/ never does.
 *
 * Parity pin: the functions below are checked (tests/test_oracle_golden.py, -m "not gpu")
 * against fixtures under tests/golden/ that oracle/gen_golden.py produced by importing the
 * reference NumPy engine in the build container, and against the known-answer vectors of
 * the reference's own tests.  The Philox4x32-10 generator is pinned by
 * the Random123 known-answer vectors.
 *
 * Board layout (build contract): 16 bytes, byte i = 4*row + col,
 * value e: 0 = empty, e >= 1 means tile 2^e.
 *
 * Every function cites the reference lines it follows as  [ref: file:lines].
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>

#define ORACLE_API __attribute__((visibility("default")))

#define FLAG_AUTO_RESET 1u
#define FLAG_RANDOM_ACTIONS 2u

/* ------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; Random123).  Build-defined RNG contract.        */
/* ------------------------------------------------------------------------------------ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
/* P(tile 4) = 0.1  [ref: gameboard.py:13,16-17]  ->  r < floor(0.1 * 2^32) */
#define SPAWN4_THRESHOLD 429496729u

ORACLE_API void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Counter scheme (DESIGN.md "RNG contract"): key = seed, counter = (index_lo, step_lo, step_hi,
 * stream + 2*index_hi).
 *   stream 0 ("step words"):  index = env >> 2; environment env owns word (env & 3) of the block,
 *                             so one Philox block serves four neighbouring environments.
 *   stream 1 ("reset block"): index = env; all four words belong to that environment. */
static void env_block(uint64_t seed, uint64_t index, uint64_t step, uint32_t stream, uint32_t out[4])
{
    uint32_t ctr[4] = { (uint32_t)index, (uint32_t)step, (uint32_t)(step >> 32),
                        stream + 2u * (uint32_t)(index >> 32) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    oracle_philox4x32_10(ctr, key, out);
}

/* The one 32-bit word environment `env` draws at `step`. */
static uint32_t step_word(uint64_t seed, uint64_t env, uint64_t step)
{
    uint32_t blk[4];
    env_block(seed, env >> 2, step, 0u, blk);
    return blk[env & 3u];
}

ORACLE_API uint32_t oracle_step_word(uint64_t seed, uint64_t env, uint64_t step)
{
    return step_word(seed, env, step);
}

ORACLE_API void oracle_env_block(uint64_t seed, uint64_t env, uint64_t step, uint32_t stream, uint32_t out[4])
{
    env_block(seed, env, step, stream, out);
}

/* ------------------------------------------------------------------------------------ */
/* Deterministic engine                                                                  */
/* ------------------------------------------------------------------------------------ */

/* [ref: gameboard.py:23-69 merge_column]  Drop zeros, merge equal neighbours left to right,
 * each tile merging at most once; score = sum of merged VALUES (2^(e+1)).  Returns the
 * unpadded result length through *out_len, like the reference's variable-length array. */
ORACLE_API int64_t oracle_merge_column(const uint8_t column[4], uint8_t out[4], int *out_len)
{
    uint8_t non_zero[4];
    int n = 0;
    for (int i = 0; i < 4; ++i)
        if (column[i] != 0) non_zero[n++] = column[i];          /* :46 column[column != 0] */
    if (n <= 1) {                                                /* :47-48 */
        for (int i = 0; i < n; ++i) out[i] = non_zero[i];
        *out_len = n;
        return 0;
    }
    int64_t score = 0;
    int m = 0;
    int i = 0;
    while (i < n - 1) {                                          /* :56 */
        if (non_zero[i] == non_zero[i + 1]) {                    /* :57 */
            uint8_t merged = (uint8_t)(non_zero[i] + 1);         /* value*2 == exponent+1 */
            out[m++] = merged;
            score += (int64_t)1 << merged;                       /* :60 score += merged */
            i += 2;
        } else {
            out[m++] = non_zero[i];
            i += 1;
        }
    }
    if (i == n - 1) out[m++] = non_zero[n - 1];                  /* :66-67 */
    *out_len = m;
    return score;
}

/* [ref: gameboard.py:72-102 slide_and_merge]  merge_column on every row, left-padded into zeros. */
ORACLE_API int64_t oracle_slide_and_merge(const uint8_t board[16], uint8_t out[16])
{
    int64_t score = 0;
    memset(out, 0, 16);                                          /* :94 zeros_like */
    for (int r = 0; r < 4; ++r) {                                /* :97 */
        uint8_t merged[4];
        int len = 0;
        score += oracle_merge_column(board + 4 * r, merged, &len);
        for (int c = 0; c < len; ++c) out[4 * r + c] = merged[c]; /* :100 */
    }
    return score;
}

/* numpy.rot90(m, k) for a 4x4 array: k counter-clockwise quarter turns, k taken mod 4
 * (negative k allowed), as used at gameboard.py:127,129,316,320. */
ORACLE_API void oracle_rot90(const uint8_t in[16], int k, uint8_t out[16])
{
    uint8_t cur[16], nxt[16];
    memcpy(cur, in, 16);
    k = ((k % 4) + 4) % 4;
    for (int t = 0; t < k; ++t) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                nxt[4 * i + j] = cur[4 * j + (3 - i)];           /* one CCW turn */
        memcpy(cur, nxt, 16);
    }
    memcpy(out, cur, 16);
}

/* [ref: gameboard.py:105-129 latent_state]  rot90(k=action) -> slide_and_merge -> rot90(k=-action).
 * No spawn, no validity check. */
ORACLE_API int64_t oracle_latent_state(const uint8_t board[16], int action, uint8_t out[16])
{
    uint8_t rotated[16], updated[16];
    oracle_rot90(board, action, rotated);                        /* :127 */
    int64_t reward = oracle_slide_and_merge(rotated, updated);   /* :128 */
    oracle_rot90(updated, -action, out);                         /* :129 */
    return reward;
}

/* [ref: gamemove.py:132-164 can_move]  "would a left move change this board". */
ORACLE_API int oracle_can_move(const uint8_t board[16])
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c) {
            uint8_t left = board[4 * r + c], right = board[4 * r + c + 1];
            if (left == 0 && right != 0) return 1;               /* :158-160 can_slide */
            if (left != 0 && left == right) return 1;            /* :163-164 can_merge */
        }
    return 0;
}

/* [ref: gamemove.py:45-83 legal_actions_mask]  bit a set <=> action a legal
 * (0 left, 1 up, 2 right, 3 down), from horizontal / vertical adjacency without rotation. */
ORACLE_API uint8_t oracle_legal_actions_mask(const uint8_t board[16])
{
    int h_merge = 0, v_merge = 0, left = 0, right = 0, up = 0, down = 0;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c) {
            uint8_t l = board[4 * r + c], rr = board[4 * r + c + 1];
            if (l != 0 && l == rr) h_merge = 1;                  /* :64-65 */
            if (l == 0 && rr != 0) left = 1;                     /* :72 */
            if (rr == 0 && l != 0) right = 1;                    /* :73 */
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            uint8_t t = board[4 * r + c], b = board[4 * (r + 1) + c];
            if (t != 0 && t == b) v_merge = 1;                   /* :68-69 */
            if (t == 0 && b != 0) up = 1;                        /* :74 */
            if (b == 0 && t != 0) down = 1;                      /* :75 */
        }
    return (uint8_t)(((left | h_merge) << 0) | ((up | v_merge) << 1) |
                     ((right | h_merge) << 2) | ((down | v_merge) << 3)); /* :77-82 */
}

/* [ref: gameboard.py:328-348 is_done]  all cells non-zero, no equal vertical neighbours,
 * no equal horizontal neighbours.  (The all-zero board is NOT done here -- NumPy semantics,
 * SURVEY.md App. A.5.) */
ORACLE_API int oracle_is_done(const uint8_t board[16])
{
    for (int i = 0; i < 16; ++i)
        if (board[i] == 0) return 0;                             /* np_all(state != 0) */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            if (board[4 * r + c] == board[4 * (r + 1) + c]) return 0; /* state[:-1] == state[1:] */
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c)
            if (board[4 * r + c] == board[4 * r + c + 1]) return 0;   /* state[:, :-1] == state[:, 1:] */
    return 1;
}

ORACLE_API int oracle_count_empty(const uint8_t board[16])           /* [ref: reinforce/game/core.py:445-460] */
{
    int n = 0;
    for (int i = 0; i < 16; ++i) n += (board[i] == 0);
    return n;
}

ORACLE_API int oracle_max_exponent(const uint8_t board[16])          /* [ref: reinforce/game/core.py:427-442] log2 of max_tile */
{
    int m = 0;
    for (int i = 0; i < 16; ++i) if (board[i] > m) m = board[i];
    return m;
}

/* Place exponent e in the k-th empty cell, cells enumerated row-major exactly like
 * argwhere(state == 0) [ref: gameboard.py:275,287; after_state_lazy :200-202;
 * reinforce/game/core.py:206-213 cumsum/argmax].  Returns the flat index, or -1 if k is out
 * of range (board unchanged) -- the full-board case of gameboard.py:277-278. */
ORACLE_API int oracle_fill_cell_kv(uint8_t board[16], int k, uint8_t e)
{
    int seen = 0;
    for (int i = 0; i < 16; ++i) {
        if (board[i] == 0) {
            if (seen == k) { board[i] = e; return i; }
            ++seen;
        }
    }
    return -1;
}

/* Build contract for turning Philox words into a spawn.
 * Two-word form (reset block): k = floor(r_cell * n_empty / 2^32), exponent 2 iff
 * r_val < SPAWN4_THRESHOLD else 1. */
static inline int spawn_rank(uint32_t r_cell, int n_empty)
{
    return (int)(((uint64_t)r_cell * (uint64_t)n_empty) >> 32);
}
static inline uint8_t spawn_exponent(uint32_t r_val)
{
    return (r_val < SPAWN4_THRESHOLD) ? 2 : 1;
}

ORACLE_API void oracle_spawn_from_words(uint32_t r_cell, uint32_t r_val, int n_empty, int *k, int *e)
{
    *k = spawn_rank(r_cell, n_empty);
    *e = spawn_exponent(r_val);
}

/* One-word form (step word w): the low two bits are the on-device random action; with
 * m = w & ~3 and the 64-bit product P = m * n_empty, the cell rank is k = P >> 32 (uniform over
 * the empties) and the residual P mod 2^32 -- uniform again, the arithmetic-decoding remainder --
 * decides the tile: exponent 2 iff (P mod 2^32) < SPAWN4_THRESHOLD. */
ORACLE_API void oracle_spawn_from_word(uint32_t w, int n_empty, int *k, int *e, int *action)
{
    uint64_t prod = (uint64_t)(w & ~3u) * (uint64_t)n_empty;
    *k = (int)(prod >> 32);
    *e = ((uint32_t)prod < SPAWN4_THRESHOLD) ? 2 : 1;
    *action = (int)(w & 3u);
}

/* Fresh board: two sequential single spawns on the empty board
 * [ref: twentyfortyeight.py:100-101 zeros + fill_cells(2); reinforce/game/core.py:339-342],
 * drawn from ONE stream-1 block: (r0,r1) among 16 empties, (r2,r3) among the remaining 15. */
ORACLE_API void oracle_reset_board(uint64_t seed, uint64_t env, uint64_t step, uint8_t board[16])
{
    uint32_t r[4];
    env_block(seed, env, step, 1u, r);
    memset(board, 0, 16);
    oracle_fill_cell_kv(board, spawn_rank(r[0], 16), spawn_exponent(r[1]));
    oracle_fill_cell_kv(board, spawn_rank(r[2], 15), spawn_exponent(r[3]));
}

/* [ref: gameboard.py:291-325 next_state + twentyfortyeight.py:132-133 step;
 *  reinforce/game/env.py:103-119 step]  One transition of one board, in place.
 *  rotated = rot90(state,k=action); if can_move(rotated): slide, un-rotate, spawn ONE tile,
 *  reward = score; else: board unchanged, reward 0, no spawn.  done = is_done(new board).
 *  With FLAG_AUTO_RESET a terminal board is replaced by a fresh one (done still reports 1).
 *  action is taken mod 4, as rot90 does (gameboard.py:316). */
static int64_t step_one_ex(uint8_t board[16], int action, uint64_t seed, uint64_t env, uint64_t step,
                           uint32_t flags, uint32_t word, int *done_out, int *changed_out, int *max_exp_out);

static int64_t step_one(uint8_t board[16], int action, uint64_t seed, uint64_t env, uint64_t step,
                        uint32_t flags, uint32_t word, int *done_out, int *changed_out)
{
    return step_one_ex(board, action, seed, env, step, flags, word, done_out, changed_out, NULL);
}

/* max_exp_out: log2(max tile) of the board after move + spawn, BEFORE an auto-reset replaces it -- for a game that ends
 * here, the max tile it reached [ref: reinforce/game/env.py:127-131 info.max_tile; self_play.py:389]. */
static int64_t step_one_ex(uint8_t board[16], int action, uint64_t seed, uint64_t env, uint64_t step,
                           uint32_t flags, uint32_t word, int *done_out, int *changed_out, int *max_exp_out)
{
    uint8_t rotated[16];
    int64_t reward = 0;
    int changed = 0;
    action &= 3;
    oracle_rot90(board, action, rotated);                        /* :316 */
    if (oracle_can_move(rotated)) {                              /* :317 */
        uint8_t updated[16];
        reward = oracle_slide_and_merge(rotated, updated);       /* :319 */
        oracle_rot90(updated, -action, board);                   /* :320 */
        int k, e, unused;
        oracle_spawn_from_word(word, oracle_count_empty(board), &k, &e, &unused);
        oracle_fill_cell_kv(board, k, (uint8_t)e);               /* :323 fill_cells(number_tile=1) */
        changed = 1;
    }                                                            /* :325 else: state, 0 */
    int done = oracle_is_done(board);                            /* twentyfortyeight.py:133 */
    if (max_exp_out) *max_exp_out = oracle_max_exponent(board);
    if (done && (flags & FLAG_AUTO_RESET))
        oracle_reset_board(seed, env, step, board);
    *done_out = done;
    if (changed_out) *changed_out = changed;
    return reward;
}

/* [ref: reinforce/game/core.py:346-377 encode_observation]  log2(tile)/16, 0 for empty. */
static void encode_obs(const uint8_t board[16], float obs[16])
{
    for (int i = 0; i < 16; ++i) obs[i] = (float)board[i] * 0.0625f;
}

/* ------------------------------------------------------------------------------------ */
/* Batched drivers: same argument meaning as include/env2048.h                           */
/* ------------------------------------------------------------------------------------ */

ORACLE_API int oracle_step_batch(uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions,
                                 int32_t *reward, uint8_t *done, uint8_t *legal, float *obs,
                                 uint64_t seed, uint64_t step_index, uint32_t flags, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint8_t *b = boards + 16 * (size_t)i;
        uint32_t w = step_word(seed, env_base + (uint64_t)i, step_index);
        int action = (flags & FLAG_RANDOM_ACTIONS) ? (int)(w & 3u) : (int)actions[i];
        int d = 0;
        int64_t r = step_one(b, action, seed, env_base + (uint64_t)i, step_index, flags, w, &d, NULL);
        if (reward) reward[i] = (int32_t)r;
        if (done) done[i] = (uint8_t)d;
        if (legal) legal[i] = oracle_legal_actions_mask(b);
        if (obs) encode_obs(b, obs + 16 * (size_t)i);
    }
    return 0;
}

/* [ref: twentyfortyeight/utils/normalize.py:6-33 normalize_reward, default max_tile = 2^16]
 * 0 -> 0.0, else log2(reward) / log2(65536), computed in double like the reference and rounded to float32. */
ORACLE_API float oracle_normalize_reward(int64_t reward)
{
    if (reward == 0) return 0.0f;                                 /* :31-32 */
    return (float)(log2((double)reward) / 16.0);                  /* :33 */
}

/* oracle_step_batch plus the two recording outputs of the step kernel: max_exp (see step_one_ex) and the
 * log-normalised reward as float32 (ENV2048_FLAG_REWARD_LOG2_F32). */
ORACLE_API int oracle_step_batch_ex(uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions,
                                    int32_t *reward, uint8_t *done, uint8_t *legal, float *obs, uint8_t *max_exp,
                                    float *reward_norm, uint64_t seed, uint64_t step_index, uint32_t flags)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t *b = boards + 16 * (size_t)i;
        uint32_t w = step_word(seed, env_base + i, step_index);
        int action = (flags & FLAG_RANDOM_ACTIONS) ? (int)(w & 3u) : (int)actions[i];
        int d = 0, m = 0;
        int64_t r = step_one_ex(b, action, seed, env_base + i, step_index, flags, w, &d, NULL, &m);
        if (reward) reward[i] = (int32_t)r;
        if (reward_norm) reward_norm[i] = oracle_normalize_reward(r);
        if (done) done[i] = (uint8_t)d;
        if (legal) legal[i] = oracle_legal_actions_mask(b);
        if (obs) encode_obs(b, obs + 16 * (size_t)i);
        if (max_exp) max_exp[i] = (uint8_t)m;
    }
    return 0;
}

ORACLE_API int oracle_reset_batch(uint64_t n, uint64_t env_base, uint8_t *boards, uint8_t *done,
                                  uint64_t seed, uint64_t step_index, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        oracle_reset_board(seed, env_base + (uint64_t)i, step_index, boards + 16 * (size_t)i);
        if (done) done[i] = 0;
    }
    return 0;
}

/* T consecutive steps with on-device-style random actions (low two bits of each step word). */
ORACLE_API int oracle_rollout_batch(uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                                    uint64_t step0, uint32_t T, int64_t *return_sum, uint32_t *episodes,
                                    uint32_t *valid_moves, uint32_t flags, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint8_t *b = boards + 16 * (size_t)i;
        int64_t ret = 0;
        uint32_t eps = 0, moves = 0;
        for (uint32_t t = 0; t < T; ++t) {
            uint64_t step = step0 + t;
            uint32_t w = step_word(seed, env_base + (uint64_t)i, step);
            int d = 0, ch = 0;
            ret += step_one(b, (int)(w & 3u), seed, env_base + (uint64_t)i, step, flags, w, &d, &ch);
            eps += (uint32_t)d;
            moves += (uint32_t)ch;
        }
        if (return_sum) return_sum[i] = ret;
        if (episodes) episodes[i] = eps;
        if (valid_moves) valid_moves[i] = moves;
    }
    return 0;
}

/* Rollout with per-EPISODE statistics [ref: reinforce/training/self_play.py:503-521 evaluate_games: rewards, max tiles and
 * lengths of finished games].  stats[40] is ADDED to: [0] games, [1] sum length, [2] sum return, [3] sum return^2,
 * [4] min return, [5] max return, [6] sum length^2, [8+e] games whose largest tile was 2^e.  ep_length / ep_return carry
 * the game in progress across calls (may be NULL). */
ORACLE_API int oracle_rollout_episodes(uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed, uint64_t step0,
                                       uint32_t T, uint32_t flags, uint32_t *ep_length, int64_t *ep_return, uint64_t *stats)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t *b = boards + 16 * (size_t)i;
        uint64_t len = ep_length ? ep_length[i] : 0, sum = ep_return ? (uint64_t)ep_return[i] : 0;
        for (uint32_t t = 0; t < T; ++t) {
            uint64_t step = step0 + t;
            uint32_t w = step_word(seed, env_base + i, step);
            int d = 0, m = 0;
            uint64_t r = (uint64_t)step_one_ex(b, (int)(w & 3u), seed, env_base + i, step, flags, w, &d, NULL, &m);
            if (len == 0xFFFFFFFFu) continue;            /* no auto-reset: this game is over and was counted */
            sum += r;
            len += 1;
            if (d) {
                stats[0] += 1; stats[1] += len; stats[2] += sum; stats[3] += sum * sum;
                if (sum < stats[4]) stats[4] = sum;
                if (sum > stats[5]) stats[5] = sum;
                stats[6] += len * len;
                stats[8 + (m & 31)] += 1;
                if (flags & FLAG_AUTO_RESET) { len = 0; sum = 0; } else { len = 0xFFFFFFFFu; }
            }
        }
        if (ep_length) ep_length[i] = (uint32_t)len;
        if (ep_return) ep_return[i] = (int64_t)sum;
    }
    return 0;
}

/* Episode records out of recorded step rows [ref: reinforce/training/self_play.py:359-392]: per game, the rows recorded
 * while it was active (:359-365; active_mask &= ~done, :368), then Trajectory(done, total_reward = sum(rewards),
 * max_tile) (:381-390).  Rows are [*, row_stride]; logical row r is physical row (first_row + r) % ring_rows.
 * flags bit 0 (FROZEN): no auto-reset, only the first episode of a game exists; bit 1: reward rows are float32.
 * Records come out by environment, then by start.  Layout of one record = env2048_episode (24 bytes). */
typedef struct {
    uint32_t env, start, length;
    uint8_t terminated, max_exp;
    uint16_t reserved;
    double total_reward;
} oracle_episode;

ORACLE_API uint64_t oracle_segment_episodes(uint64_t n, uint64_t row_stride, uint32_t rows, uint32_t first_row, uint32_t ring_rows,
                                            const uint8_t *done, const void *reward, const uint8_t *max_exp, uint32_t flags,
                                            oracle_episode *episodes, int32_t *episode_of)
{
    uint64_t count = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t start = 0, length = 0, best = 0, r = 0;
        double total = 0.0;
        for (; r < rows; ++r) {
            uint64_t at = (uint64_t)((first_row + r) % ring_rows) * row_stride + i;
            if (reward) total += (flags & 2u) ? (double)((const float *)reward)[at] : (double)((const int32_t *)reward)[at];
            if (max_exp && max_exp[at] > best) best = max_exp[at];
            ++length;
            if (episode_of) episode_of[at] = (int32_t)count;
            if (done[at]) {
                if (episodes) episodes[count] = (oracle_episode){(uint32_t)i, start, length, 1, (uint8_t)best, 0, total};
                ++count;
                start = r + 1; length = 0; best = 0; total = 0.0;
                if (flags & 1u) { ++r; break; }
            }
        }
        if (length) {
            if (episodes) episodes[count] = (oracle_episode){(uint32_t)i, start, length, 0, (uint8_t)best, 0, total};
            ++count;
        }
        if (episode_of)
            for (; r < rows; ++r) episode_of[(uint64_t)((first_row + r) % ring_rows) * row_stride + i] = -1;
    }
    return count;
}

/* [ref: reinforce/game/env.py:79-133 step]  the JAX-shaped transition on int32 value boards:
 * done-guard (:103-104), next_state (:108), is_done = no legal action (:111 -> core.py:320-321),
 * step_count + 1 also on invalid moves (:116), total_reward += reward (:118), info.max_tile (:128). */
ORACLE_API int oracle_step_state_i32(uint64_t n, uint64_t env_base, int32_t *boards, const int32_t *actions,
                                     int32_t *step_count, uint8_t *done_io, float *total_reward, float *reward,
                                     int32_t *max_tile, uint64_t seed, uint64_t step_index)
{
    for (uint64_t i = 0; i < n; ++i) {
        int32_t *v = boards + 16 * i;
        uint8_t b[16];
        for (int c = 0; c < 16; ++c) {
            int e = 0;
            for (int32_t x = v[c]; x > 1; x >>= 1) ++e;
            b[c] = (uint8_t)(v[c] > 0 ? e : 0);
        }
        float r = 0.0f;
        if (!done_io[i]) {                                                   /* :124 lax.cond(state.done, ...) */
            int d = 0, ch = 0;
            uint32_t w = step_word(seed, env_base + i, step_index);
            r = (float)step_one(b, (int)((uint32_t)actions[i] & 3u), seed, env_base + i, step_index, 0u, w, &d, &ch);
            for (int c = 0; c < 16; ++c) v[c] = b[c] ? (int32_t)1 << b[c] : 0;
            done_io[i] = (uint8_t)(oracle_legal_actions_mask(b) == 0);      /* :111 */
            if (step_count) step_count[i] += 1;                             /* :116 */
            if (total_reward) total_reward[i] += r;                         /* :118 */
        }
        reward[i] = r;
        if (max_tile) {
            int m = oracle_max_exponent(b);
            max_tile[i] = m ? (int32_t)1 << m : 0;                          /* :128 */
        }
    }
    return 0;
}

/* The same under jax.vmap with one key per environment [ref: reinforce/game/env.py:193-211 batched_step =
 * vmap(step); the caller hands every environment its own key, reinforce/training/self_play.py:351-354]: row i is BY
 * DEFINITION the single-environment step with seed (keys[i][0] << 32) | keys[i][1] -- the composition below. */
ORACLE_API int oracle_step_state_i32_keyed(uint64_t n, const uint32_t *keys, int32_t *boards, const int32_t *actions,
                                           int32_t *step_count, uint8_t *done_io, float *total_reward, float *reward,
                                           int32_t *max_tile)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t seed = ((uint64_t)keys[2 * i] << 32) | keys[2 * i + 1];
        oracle_step_state_i32(1, 0, boards + 16 * i, actions + i, step_count ? step_count + i : NULL, done_io + i,
                              total_reward ? total_reward + i : NULL, reward + i, max_tile ? max_tile + i : NULL, seed, 0);
    }
    return 0;
}

/* [ref: reinforce/game/core.py:324-343 create_initial_board under vmap; env.py:176-190 batched_reset]  fresh int32
 * value boards, board i from the reset block of environment 0, step 0 of the stream seeded with keys[i]. */
ORACLE_API int oracle_reset_keyed_i32(uint64_t n, const uint32_t *keys, int32_t *boards)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t b[16];
        oracle_reset_board(((uint64_t)keys[2 * i] << 32) | keys[2 * i + 1], 0, 0, b);
        for (int c = 0; c < 16; ++c) boards[16 * i + c] = b[c] ? (int32_t)1 << b[c] : 0;
    }
    return 0;
}

/* Afterstate only [ref: gameboard.py:105-129 latent_state; reinforce/game/core.py:129-170]:
 * out = slide/merge of in under action; reward = score (NOT zeroed for invalid moves: an
 * invalid move merges nothing, so it is 0 anyway); changed = (out != in)
 * [ref: reinforce/game/core.py:258]. */
ORACLE_API int oracle_afterstate_batch(uint64_t n, const uint8_t *boards_in, const uint8_t *actions,
                                       uint8_t *boards_out, int32_t *reward, uint8_t *changed)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t out[16];
        int64_t r = oracle_latent_state(boards_in + 16 * i, actions[i] & 3, out);
        if (changed) changed[i] = (uint8_t)(memcmp(out, boards_in + 16 * i, 16) != 0);
        if (reward) reward[i] = (int32_t)r;
        memcpy(boards_out + 16 * i, out, 16);
    }
    return 0;
}

/* Injected spawn [ref: gameboard.py:174-244 after_state_lazy + generate_outcome]:
 * place exponent spawn_e[i] at the spawn_k[i]-th empty cell (row-major). placed[i] = flat
 * index or 0xFF when k is out of range (board unchanged). */
ORACLE_API int oracle_spawn_batch(uint64_t n, uint8_t *boards, const uint8_t *spawn_k,
                                  const uint8_t *spawn_e, uint8_t *placed)
{
    for (uint64_t i = 0; i < n; ++i) {
        int idx = oracle_fill_cell_kv(boards + 16 * i, spawn_k[i], spawn_e[i]);
        if (placed) placed[i] = (uint8_t)(idx < 0 ? 0xFF : idx);
    }
    return 0;
}

/* `count` sequential random spawns per board from the step words of (env, step_index + t)
 * [ref: gameboard.py:247-288 fill_cells]; a full board stays unchanged (:277-278). */
ORACLE_API int oracle_spawn_random_batch(uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed,
                                         uint64_t step_index, uint32_t count)
{
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t t = 0; t < count; ++t) {
            uint8_t *b = boards + 16 * i;
            int n_empty = oracle_count_empty(b);
            if (n_empty == 0) continue;
            int k, e, unused;
            oracle_spawn_from_word(step_word(seed, env_base + i, step_index + t), n_empty, &k, &e, &unused);
            oracle_fill_cell_kv(b, k, (uint8_t)e);
        }
    return 0;
}

/* [ref: gameboard.py:132-171 after_state]  every (empty cell row-major, value 2 then 4) outcome with
 * probability P(value)/n_empty, in the reference's enumeration order; 32 slots per board. */
ORACLE_API int oracle_chance_outcomes(uint64_t n, const uint8_t *boards, uint8_t *out_boards, float *out_prob,
                                      uint8_t *out_count)
{
    memset(out_boards, 0, (size_t)n * 32 * 16);
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + 16 * i;
        int n_empty = oracle_count_empty(b);
        for (int s = 0; s < 32; ++s) out_prob[32 * i + s] = 0.0f;
        if (n_empty == 0) {                                      /* :156-157 */
            memcpy(out_boards + (32 * i) * 16, b, 16);
            out_prob[32 * i] = 1.0f;
        } else {
            for (int k = 0; k < n_empty; ++k)                    /* :161 for cell in empty_cells */
                for (int v = 0; v < 2; ++v) {                    /* :162 for new_value in [2, 4] */
                    uint8_t *o = out_boards + (32 * i + 2 * k + v) * 16;
                    memcpy(o, b, 16);
                    oracle_fill_cell_kv(o, k, (uint8_t)(v + 1));
                    out_prob[32 * i + 2 * k + v] = (v ? 0.1f : 0.9f) / (float)n_empty;   /* :166 */
                }
        }
        if (out_count) out_count[i] = (uint8_t)(n_empty ? 2 * n_empty : 1);
    }
    return 0;
}

/* One expectimax ply [ref: gameboard.py:105-129 latent_state for each action; :132-171 after_state on every LEGAL
 * afterstate (legality: the move changes the board, gamemove.py:132-164 / gameboard.py:317); per outcome
 * gamemove.py:45-83 legal_actions_mask and gameboard.py:328-348 is_done].  Same slot layout as oracle_chance_outcomes;
 * an illegal action has count 0 and all-zero slots. */
ORACLE_API int oracle_expectimax_ply(uint64_t n, const uint8_t *boards, uint8_t *succ_boards, float *succ_prob,
                                     uint8_t *succ_status, uint8_t *succ_count, int32_t *after_reward, uint8_t *legal)
{
    memset(succ_boards, 0, (size_t)n * 128 * 16);
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + 16 * i;
        uint8_t mask = 0;
        for (int a = 0; a < 4; ++a) {
            uint8_t after[16];
            int64_t r = oracle_latent_state(b, a, after);
            int changed = memcmp(after, b, 16) != 0;
            int n_empty = oracle_count_empty(after);
            uint64_t base = (4 * i + (uint64_t)a) * 32;
            if (after_reward) after_reward[4 * i + a] = (int32_t)r;
            if (succ_count) succ_count[4 * i + a] = (uint8_t)(changed ? 2 * n_empty : 0);
            for (int s = 0; s < 32; ++s) {
                succ_prob[base + s] = 0.0f;
                if (succ_status) succ_status[base + s] = 0;
            }
            if (!changed) continue;
            mask |= (uint8_t)(1u << a);
            for (int k = 0; k < n_empty; ++k)
                for (int v = 0; v < 2; ++v) {
                    uint8_t *o = succ_boards + (base + 2 * k + v) * 16;
                    memcpy(o, after, 16);
                    oracle_fill_cell_kv(o, k, (uint8_t)(v + 1));
                    succ_prob[base + 2 * k + v] = (v ? 0.1f : 0.9f) / (float)n_empty;
                    if (succ_status)
                        succ_status[base + 2 * k + v] = (uint8_t)(oracle_legal_actions_mask(o) | (oracle_is_done(o) ? 16 : 0));
                }
        }
        if (legal) legal[i] = mask;
    }
    return 0;
}

/* [ref: twentyfortyeight/utils/binary.py:11-86 encode_flatten]  eye(width)[log2(cell)], empty -> index 0. */
/* Depth-2 expectimax composed from the reference's building blocks:
 *   Q2(s, a) = r(s, a) + sum_o P(o) * max_{legal a'} r(o, a')   with  latent_state [ref: gameboard.py:105-129],
 *   after_state's outcomes and P(o) = TILE_SPAWN_PROBS[v] / num_empty [ref: gameboard.py:132-171], legality = the move changes
 *   the board [ref: gamemove.py:45-83, 132-164]; a successor with no legal move contributes 0; V2 = max over legal a, best =
 *   the smallest maximising action (255 for a terminal board).  terms = {r, S2, S4, num_empty} per (board, action) with
 *   S2 / S4 the integer sums over the empty cells for a spawned 2 / 4; q = r + (0.9 S2 + 0.1 S4) / num_empty evaluated as the
 *   exact fraction (10 n r + 9 S2 + S4) / (10 n): one division in double, arg-max by exact integer comparison. */
ORACLE_API int oracle_expectimax_depth2(uint64_t n, const uint8_t *boards, int32_t *terms, double *q, uint8_t *best_action, double *value)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + 16 * i;
        int64_t best_x = 0, best_d = 1; /* Q2 = X / D with X = 10 n r + 9 S2 + S4, D = 10 n: compared exactly */
        int best_a = 255;
        for (int a = 0; a < 4; ++a) {
            uint8_t after[16];
            int64_t r = oracle_latent_state(b, a, after);
            int changed = memcmp(after, b, 16) != 0;
            int32_t *t = terms + (4 * i + (uint64_t)a) * 4;
            t[0] = t[1] = t[2] = t[3] = 0;
            q[4 * i + a] = -INFINITY;
            if (!changed) continue;
            int n_empty = oracle_count_empty(after);
            int64_t s[2] = {0, 0};
            for (int k = 0; k < n_empty; ++k)
                for (int v = 0; v < 2; ++v) {
                    uint8_t o[16], tmp[16];
                    memcpy(o, after, 16);
                    oracle_fill_cell_kv(o, k, (uint8_t)(v + 1));
                    int64_t best = 0;
                    for (int a2 = 0; a2 < 4; ++a2) {
                        int64_t r2 = oracle_latent_state(o, a2, tmp);
                        if (memcmp(tmp, o, 16) != 0 && r2 > best) best = r2;
                    }
                    s[v] += best;
                }
            t[0] = (int32_t)r;
            t[1] = (int32_t)s[0];
            t[2] = (int32_t)s[1];
            t[3] = n_empty;
            const int64_t d = 10 * (int64_t)n_empty, x = d * r + 9 * s[0] + s[1];
            q[4 * i + a] = (double)x / (double)d; /* = r + (0.9 S2 + 0.1 S4) / n, one correctly rounded division */
            if (best_a == 255 || x * best_d > best_x * d) { /* exact comparison of the fractions; a tie keeps the smaller action */
                best_x = x;
                best_d = d;
                best_a = a;
            }
        }
        best_action[i] = (uint8_t)best_a;
        value[i] = (double)best_x / (double)best_d;
    }
    return 0;
}

/* Depth-d expectimax (d >= 2) by plain recursion over the same building blocks:
 *   V_2(s) = the value oracle_expectimax_depth2 reports (exact fraction, one division);
 *   Q_d(s, a) = r(s, a) + (0.9 * S2 + 0.1 * S4) / num_empty  for d >= 3, with S2 / S4 = the sums of V_{d-1} over the outcomes
 *   of after_state(afterstate) that spawned a 2 / a 4, added in after_state's order (empty cells row-major) [ref:
 *   gameboard.py:105-171 composed as in the depth-2 case]; V_d = max over the legal a (the first maximum), 0 for a terminal board.
 * Every operation is a separate IEEE double operation in the order written (ISO C mode: no contraction), which the HIP back-up
 * kernel repeats under `#pragma clang fp contract(off)`, so the two agree bit for bit. */
static double oracle_value_depth(const uint8_t *b, int depth)
{
    if (depth <= 2) {
        int32_t terms[16];
        double q[4], value;
        uint8_t best;
        oracle_expectimax_depth2(1, b, terms, q, &best, &value);
        return best == 255 ? 0.0 : value;
    }
    double best_q = 0.0;
    int best_a = 255;
    for (int a = 0; a < 4; ++a) {
        uint8_t after[16];
        const int64_t r = oracle_latent_state(b, a, after);
        if (memcmp(after, b, 16) == 0) continue;
        const int n_empty = oracle_count_empty(after);
        double s2 = 0.0, s4 = 0.0;
        for (int k = 0; k < n_empty; ++k) {
            uint8_t o[16];
            memcpy(o, after, 16);
            oracle_fill_cell_kv(o, k, 1);
            s2 = s2 + oracle_value_depth(o, depth - 1);
            memcpy(o, after, 16);
            oracle_fill_cell_kv(o, k, 2);
            s4 = s4 + oracle_value_depth(o, depth - 1);
        }
        const double t2 = 0.9 * s2, t4 = 0.1 * s4, mix = t2 + t4, mean = mix / (double)n_empty, q = (double)r + mean;
        if (best_a == 255 || q > best_q) {
            best_q = q;
            best_a = a;
        }
    }
    return best_a == 255 ? 0.0 : best_q;
}

ORACLE_API int oracle_expectimax_depth(uint64_t n, const uint8_t *boards, int depth, double *q, uint8_t *best_action, double *value)
{
    if (depth < 2) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
        const uint64_t i = (uint64_t)ii;
        const uint8_t *b = boards + 16 * i;
        if (depth == 2) {
            int32_t terms[16];
            oracle_expectimax_depth2(1, b, terms, q + 4 * i, best_action + i, value + i);
            if (best_action[i] == 255) value[i] = 0.0;
            continue;
        }
        double best_q = 0.0;
        int best_a = 255;
        for (int a = 0; a < 4; ++a) {
            uint8_t after[16];
            const int64_t r = oracle_latent_state(b, a, after);
            q[4 * i + a] = -INFINITY;
            if (memcmp(after, b, 16) == 0) continue;
            const int n_empty = oracle_count_empty(after);
            double s2 = 0.0, s4 = 0.0;
            for (int k = 0; k < n_empty; ++k) {
                uint8_t o[16];
                memcpy(o, after, 16);
                oracle_fill_cell_kv(o, k, 1);
                s2 = s2 + oracle_value_depth(o, depth - 1);
                memcpy(o, after, 16);
                oracle_fill_cell_kv(o, k, 2);
                s4 = s4 + oracle_value_depth(o, depth - 1);
            }
            const double t2 = 0.9 * s2, t4 = 0.1 * s4, mix = t2 + t4, mean = mix / (double)n_empty;
            q[4 * i + a] = (double)r + mean;
            if (best_a == 255 || q[4 * i + a] > best_q) {
                best_q = q[4 * i + a];
                best_a = a;
            }
        }
        best_action[i] = (uint8_t)best_a;
        value[i] = best_a == 255 ? 0.0 : best_q;
    }
    return 0;
}

ORACLE_API int oracle_encode_onehot(uint64_t n, const uint8_t *boards, uint32_t width, uint8_t *out)
{
    memset(out, 0, (size_t)n * 16 * width);
    for (uint64_t c = 0; c < 16 * n; ++c)
        if (boards[c] < width) out[c * width + boards[c]] = 1;
    return 0;
}

/* Board status: done / legal bitmask / f32 obs / max exponent / empty count. */
ORACLE_API int oracle_status_batch(uint64_t n, const uint8_t *boards, uint8_t *done, uint8_t *legal,
                                   float *obs, uint8_t *max_exp, uint8_t *n_empty)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + 16 * i;
        if (done) done[i] = (uint8_t)oracle_is_done(b);
        if (legal) legal[i] = oracle_legal_actions_mask(b);
        if (obs) encode_obs(b, obs + 16 * i);
        if (max_exp) max_exp[i] = (uint8_t)oracle_max_exponent(b);
        if (n_empty) n_empty[i] = (uint8_t)oracle_count_empty(b);
    }
    return 0;
}

ORACLE_API int oracle_openmp_enabled(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

/* ==================================================================================== */
/* Boards of any size S x S, 2 <= S <= 8  [ref: twentyfortyeight/envs/twentyfortyeight.py:21  */
/* `size`; the reference's core functions are written for any shape and its own timing script */
/* runs 4 / 6 / 8, tests/test_perf_utils.py:47].  One byte per cell, row-major, exponents.     */
/* The same restatement as above with the length as a parameter -- and the same RNG contract:  */
/* the step word of (seed, env, step), two sequential spawns from the reset block.  For S = 4   */
/* every function here equals its 4x4 counterpart (tests/test_nxn.py).                          */
/* ==================================================================================== */
#define NXN_MAX 8

/* [ref: gameboard.py:23-69 merge_column] for a line of `len` cells. */
ORACLE_API int64_t oracle_nxn_merge_column(const uint8_t *column, int len, uint8_t *out, int *out_len)
{
    uint8_t non_zero[NXN_MAX];
    int n = 0;
    for (int i = 0; i < len; ++i)
        if (column[i] != 0) non_zero[n++] = column[i];          /* :46 */
    int64_t score = 0;
    int m = 0, i = 0;
    if (n <= 1) {                                                /* :47-48 */
        for (i = 0; i < n; ++i) out[i] = non_zero[i];
        *out_len = n;
        return 0;
    }
    while (i < n - 1) {                                          /* :56 */
        if (non_zero[i] == non_zero[i + 1]) {                    /* :57 */
            const uint8_t merged = (uint8_t)(non_zero[i] + 1);
            out[m++] = merged;
            score += (int64_t)1 << merged;                       /* :60 */
            i += 2;
        } else {
            out[m++] = non_zero[i];
            i += 1;
        }
    }
    if (i == n - 1) out[m++] = non_zero[n - 1];                  /* :66-67 */
    *out_len = m;
    return score;
}

/* [ref: gameboard.py:72-102 slide_and_merge] */
ORACLE_API int64_t oracle_nxn_slide_and_merge(const uint8_t *board, int S, uint8_t *out)
{
    int64_t score = 0;
    memset(out, 0, (size_t)(S * S));
    for (int r = 0; r < S; ++r) {
        uint8_t merged[NXN_MAX];
        int len = 0;
        score += oracle_nxn_merge_column(board + S * r, S, merged, &len);
        for (int c = 0; c < len; ++c) out[S * r + c] = merged[c];
    }
    return score;
}

/* numpy.rot90(m, k) for an S x S array. */
ORACLE_API void oracle_nxn_rot90(const uint8_t *in, int S, int k, uint8_t *out)
{
    uint8_t cur[NXN_MAX * NXN_MAX], nxt[NXN_MAX * NXN_MAX];
    memcpy(cur, in, (size_t)(S * S));
    k = ((k % 4) + 4) % 4;
    for (int t = 0; t < k; ++t) {
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j)
                nxt[S * i + j] = cur[S * j + (S - 1 - i)];       /* one CCW turn */
        memcpy(cur, nxt, (size_t)(S * S));
    }
    memcpy(out, cur, (size_t)(S * S));
}

/* [ref: gameboard.py:105-129 latent_state] */
ORACLE_API int64_t oracle_nxn_latent_state(const uint8_t *board, int S, int action, uint8_t *out)
{
    uint8_t rotated[NXN_MAX * NXN_MAX], updated[NXN_MAX * NXN_MAX];
    oracle_nxn_rot90(board, S, action, rotated);
    const int64_t reward = oracle_nxn_slide_and_merge(rotated, S, updated);
    oracle_nxn_rot90(updated, S, -action, out);
    return reward;
}

/* [ref: gamemove.py:132-164 can_move] */
ORACLE_API int oracle_nxn_can_move(const uint8_t *board, int S)
{
    for (int r = 0; r < S; ++r)
        for (int c = 0; c < S - 1; ++c) {
            const uint8_t left = board[S * r + c], right = board[S * r + c + 1];
            if (left == 0 && right != 0) return 1;
            if (left != 0 && left == right) return 1;
        }
    return 0;
}

/* [ref: gamemove.py:45-83 legal_actions_mask] */
ORACLE_API uint8_t oracle_nxn_legal_actions_mask(const uint8_t *board, int S)
{
    int h_merge = 0, v_merge = 0, left = 0, right = 0, up = 0, down = 0;
    for (int r = 0; r < S; ++r)
        for (int c = 0; c < S - 1; ++c) {
            const uint8_t l = board[S * r + c], rr = board[S * r + c + 1];
            if (l != 0 && l == rr) h_merge = 1;
            if (l == 0 && rr != 0) left = 1;
            if (rr == 0 && l != 0) right = 1;
        }
    for (int r = 0; r < S - 1; ++r)
        for (int c = 0; c < S; ++c) {
            const uint8_t t = board[S * r + c], b = board[S * (r + 1) + c];
            if (t != 0 && t == b) v_merge = 1;
            if (t == 0 && b != 0) up = 1;
            if (b == 0 && t != 0) down = 1;
        }
    return (uint8_t)(((left | h_merge) << 0) | ((up | v_merge) << 1) | ((right | h_merge) << 2) | ((down | v_merge) << 3));
}

/* [ref: gameboard.py:328-348 is_done] */
ORACLE_API int oracle_nxn_is_done(const uint8_t *board, int S)
{
    for (int i = 0; i < S * S; ++i)
        if (board[i] == 0) return 0;
    for (int r = 0; r < S - 1; ++r)
        for (int c = 0; c < S; ++c)
            if (board[S * r + c] == board[S * (r + 1) + c]) return 0;
    for (int r = 0; r < S; ++r)
        for (int c = 0; c < S - 1; ++c)
            if (board[S * r + c] == board[S * r + c + 1]) return 0;
    return 1;
}

static int nxn_count_empty(const uint8_t *board, int S)
{
    int n = 0;
    for (int i = 0; i < S * S; ++i) n += (board[i] == 0);
    return n;
}

/* k-th empty cell in argwhere order [ref: gameboard.py:275,287]. */
ORACLE_API int oracle_nxn_fill_cell_kv(uint8_t *board, int S, int k, uint8_t e)
{
    int seen = 0;
    for (int i = 0; i < S * S; ++i)
        if (board[i] == 0) {
            if (seen == k) { board[i] = e; return i; }
            ++seen;
        }
    return -1;
}

/* Fresh board: two sequential single spawns from ONE reset block, (r0, r1) among the S*S empties, (r2, r3) among the
 * remaining S*S - 1  [ref: twentyfortyeight.py:100-101]. */
ORACLE_API void oracle_nxn_reset_board(uint64_t seed, uint64_t env, uint64_t step, int S, uint8_t *board)
{
    uint32_t r[4];
    env_block(seed, env, step, 1u, r);
    memset(board, 0, (size_t)(S * S));
    oracle_nxn_fill_cell_kv(board, S, spawn_rank(r[0], S * S), spawn_exponent(r[1]));
    oracle_nxn_fill_cell_kv(board, S, spawn_rank(r[2], S * S - 1), spawn_exponent(r[3]));
}

/* [ref: gameboard.py:291-325 next_state + twentyfortyeight.py:132-133] one transition of one board in place. */
static int64_t nxn_step_one(uint8_t *board, int S, int action, uint64_t seed, uint64_t env, uint64_t step, uint32_t flags,
                            uint32_t word, int *done_out)
{
    uint8_t rotated[NXN_MAX * NXN_MAX];
    int64_t reward = 0;
    action &= 3;
    oracle_nxn_rot90(board, S, action, rotated);                 /* :316 */
    if (oracle_nxn_can_move(rotated, S)) {                       /* :317 */
        uint8_t updated[NXN_MAX * NXN_MAX];
        reward = oracle_nxn_slide_and_merge(rotated, S, updated);
        oracle_nxn_rot90(updated, S, -action, board);
        int k, e, unused;
        oracle_spawn_from_word(word, nxn_count_empty(board, S), &k, &e, &unused);
        oracle_nxn_fill_cell_kv(board, S, k, (uint8_t)e);        /* :323 */
    }
    const int done = oracle_nxn_is_done(board, S);
    if (done && (flags & FLAG_AUTO_RESET)) oracle_nxn_reset_board(seed, env, step, S, board);
    *done_out = done;
    return reward;
}

ORACLE_API int oracle_nxn_step_batch(uint64_t n, int S, uint64_t env_base, uint8_t *boards, const uint8_t *actions, int32_t *reward,
                                     uint8_t *done, uint8_t *legal, uint64_t seed, uint64_t step, uint32_t flags)
{
    if (S < 2 || S > NXN_MAX) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t env = env_base + i;
        const uint32_t word = step_word(seed, env, step);
        const int action = (flags & FLAG_RANDOM_ACTIONS) ? (int)(word & 3u) : (int)actions[i];
        int d = 0;
        const int64_t r = nxn_step_one(boards + (size_t)S * S * i, S, action, seed, env, step, flags, word, &d);
        if (reward) reward[i] = (int32_t)r;
        done[i] = (uint8_t)d;
        if (legal) legal[i] = oracle_nxn_legal_actions_mask(boards + (size_t)S * S * i, S);
    }
    return 0;
}

ORACLE_API int oracle_nxn_afterstate_batch(uint64_t n, int S, const uint8_t *boards, const uint8_t *actions, uint8_t *out, int32_t *reward,
                                           uint8_t *changed)
{
    if (S < 2 || S > NXN_MAX) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + (size_t)S * S * i;
        uint8_t *o = out + (size_t)S * S * i;
        const int64_t r = oracle_nxn_latent_state(b, S, actions[i] & 3, o);
        if (reward) reward[i] = (int32_t)r;
        if (changed) changed[i] = (uint8_t)(memcmp(b, o, (size_t)(S * S)) != 0);
    }
    return 0;
}

ORACLE_API int oracle_nxn_status_batch(uint64_t n, int S, const uint8_t *boards, uint8_t *done, uint8_t *legal, uint8_t *n_empty,
                                       uint8_t *max_exp)
{
    if (S < 2 || S > NXN_MAX) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + (size_t)S * S * i;
        if (done) done[i] = (uint8_t)oracle_nxn_is_done(b, S);
        if (legal) legal[i] = oracle_nxn_legal_actions_mask(b, S);
        if (n_empty) n_empty[i] = (uint8_t)nxn_count_empty(b, S);
        if (max_exp) {
            uint8_t m = 0;
            for (int c = 0; c < S * S; ++c) m = b[c] > m ? b[c] : m;
            max_exp[i] = m;
        }
    }
    return 0;
}

ORACLE_API int oracle_nxn_reset_batch(uint64_t n, int S, uint64_t env_base, uint8_t *boards, uint64_t seed, uint64_t step)
{
    if (S < 2 || S > NXN_MAX) return -1;
    for (uint64_t i = 0; i < n; ++i) oracle_nxn_reset_board(seed, env_base + i, step, S, boards + (size_t)S * S * i);
    return 0;
}

/* `count` sequential single spawns per board from the step words of (seed, env, step .. step + count - 1)
 * [ref: gameboard.py:247-288 fill_cells: number_tile tiles on distinct empty cells]. */
ORACLE_API int oracle_nxn_spawn_random_batch(uint64_t n, int S, uint64_t env_base, uint8_t *boards, uint64_t seed, uint64_t step,
                                             uint32_t count)
{
    if (S < 2 || S > NXN_MAX) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t *b = boards + (size_t)S * S * i;
        for (uint32_t t = 0; t < count; ++t) {
            const int ne = nxn_count_empty(b, S);
            if (ne == 0) break;
            int k, e, unused;
            oracle_spawn_from_word(step_word(seed, env_base + i, step + t), ne, &k, &e, &unused);
            oracle_nxn_fill_cell_kv(b, S, k, (uint8_t)e);
        }
    }
    return 0;
}

/* [ref: gameboard.py:132-171 after_state] every (empty cell, value) outcome, cells in argwhere order, value 2 then 4;
 * out = n x (2 S S) boards, count[i] = number of outcomes (1 = the board itself when it is full). */
ORACLE_API int oracle_nxn_chance_outcomes(uint64_t n, int S, const uint8_t *boards, uint8_t *out, uint8_t *count)
{
    if (S < 2 || S > NXN_MAX) return -1;
    const size_t cells = (size_t)S * S;
    memset(out, 0, n * 2 * cells * cells);
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t *b = boards + cells * i;
        uint8_t *o = out + 2 * cells * cells * i;
        const int ne = nxn_count_empty(b, S);
        if (ne == 0) {                                           /* :156-157 */
            memcpy(o, b, cells);
            count[i] = 1;
            continue;
        }
        for (int k = 0; k < ne; ++k)
            for (int e = 1; e <= 2; ++e) {
                uint8_t *slot = o + cells * (size_t)(2 * k + (e - 1));
                memcpy(slot, b, cells);
                oracle_nxn_fill_cell_kv(slot, S, k, (uint8_t)e);
            }
        count[i] = (uint8_t)(2 * ne);
    }
    return 0;
}
