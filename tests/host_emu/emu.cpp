// tests/host_emu/emu.cpp -- CPU build of simulate_2048_amd/csrc/board_ops.hpp (ENV2048_HOST_EMULATION).
// TEST HARNESS ONLY: lets the per-board SWAR arithmetic of the HIP kernels be checked against the
// oracle under UBSan (the shared object, tests/test_host_emulation.py) and under ASan + UBSan (the standalone
// driver tests/host_emu/asan_main.cpp, which includes this file) without a GPU.  Never loaded by the product.
#define ENV2048_HOST_EMULATION 1
#include "board_ops.hpp"

#include <cstring>

using namespace env2048;

static Board load(const uint8_t *p)
{
    Board b;
    std::memcpy(&b, p, 16);
    return b;
}
static void store(uint8_t *p, const Board &b) { std::memcpy(p, &b, 16); }
static DirSelectors selectors(uint32_t action)
{
    DirSelectors s;
    std::memcpy(&s, DIR_SELECTORS + 8 * (action & 3u), sizeof s);
    return s;
}

extern "C" {

void emu_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out[0], out[1], out[2], out[3]);
}

// mode 0: cndmask afterstate(); mode 1: selector-table afterstate_sel()
void emu_afterstate(uint64_t n, const uint8_t *in, const uint8_t *actions, uint8_t *out, int32_t *reward,
                    uint8_t *changed, int mode)
{
    for (uint64_t i = 0; i < n; ++i) {
        Board b = load(in + 16 * i);
        uint32_t score;
        bool ch;
        if (mode == 0)
            afterstate(b, actions[i] & 3u, score, ch);
        else
            afterstate_sel(b, selectors(actions[i]), score, ch);
        store(out + 16 * i, b);
        reward[i] = (int32_t)score;
        changed[i] = ch;
    }
}

// full transition with caller-provided step words (and the reset block computed here)
void emu_step(uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions, const uint32_t *words,
              int32_t *reward, uint8_t *done, uint8_t *legal, uint64_t seed, uint64_t step, uint32_t flags)
{
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    for (uint64_t i = 0; i < n; ++i) {
        Board b = load(boards + 16 * i);
        const uint32_t w = words[i];
        const uint32_t action = (flags & 2u) ? (w & 3u) : (actions[i] & 3u);
        uint32_t r;
        bool ch, dn;
        transition(b, selectors(action), w, r, ch, dn);
        if ((flags & 1u) && dn) {
            uint32_t w0, w1, w2, w3;
            env_block(key, env_base + i, STREAM_RESET, w0, w1, w2, w3);
            b = fresh_board(w0, w1, w2, w3);
        }
        store(boards + 16 * i, b);
        reward[i] = (int32_t)r;
        done[i] = dn;
        if (legal) legal[i] = (uint8_t)legal_mask(b);
    }
}

// T transitions with every board held in the plane space of its last move (what the multi-step kernels do) and mapped
// back to rows once at the end: must equal T emu_step calls.  words / actions are [T][n].
void emu_steps_oriented(uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions, const uint32_t *words, uint32_t T,
                        int64_t *reward_sum, uint32_t *done_count, uint8_t *last_done, uint64_t seed, uint64_t step0, uint32_t flags)
{
    for (uint64_t i = 0; i < n; ++i) {
        Board pl = load(boards + 16 * i);
        uint32_t ori = ORIENT_BOARD;
        int64_t sum = 0;
        uint32_t dones = 0;
        bool dn = false;
        for (uint32_t t = 0; t < T; ++t) {
            const uint32_t w = words[(uint64_t)t * n + i];
            const uint32_t action = (flags & 2u) ? (w & 3u) : (actions[(uint64_t)t * n + i] & 3u);
            OrientStep s;
            std::memcpy(&s.sA, ORIENT_SELECTORS + 4 * (4 * ori + action), 16);
            std::memcpy(&s.m0, ORIENT_RANK_MASKS + 4 * action, 16);
            uint32_t r;
            bool ch;
            transition_oriented(pl, s, w, r, ch, dn);
            ori = action;
            sum += r;
            dones += dn;
            if ((flags & 1u) && dn) {
                const uint64_t step = step0 + t;
                const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
                uint32_t w0, w1, w2, w3;
                env_block(key, env_base + i, STREAM_RESET, w0, w1, w2, w3);
                pl = fresh_board(w0, w1, w2, w3);
                ori = ORIENT_BOARD;
            }
        }
        const uint32_t *back = ORIENT_SELECTORS + 4 * (4 * ori + ORIENT_BOARD);
        store(boards + 16 * i, orient_network(pl, back[0], back[1], back[2], back[3]));
        reward_sum[i] = sum;
        done_count[i] = dones;
        last_done[i] = dn;
    }
}

void emu_step_words(uint64_t n, uint64_t env_base, uint64_t seed, uint64_t step, uint32_t *words)
{
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t env = env_base + i;
        uint32_t w[4];
        env_block(key, env >> 2, STREAM_STEP, w[0], w[1], w[2], w[3]);
        words[i] = w[env & 3u];
    }
}

void emu_reset(uint64_t n, uint64_t env_base, uint8_t *boards, uint64_t seed, uint64_t step)
{
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t w0, w1, w2, w3;
        env_block(key, env_base + i, STREAM_RESET, w0, w1, w2, w3);
        store(boards + 16 * i, fresh_board(w0, w1, w2, w3));
    }
}

void emu_status(uint64_t n, const uint8_t *boards, uint8_t *done, uint8_t *legal, uint8_t *n_empty)
{
    for (uint64_t i = 0; i < n; ++i) {
        const Board b = load(boards + 16 * i);
        const uint32_t e = count_empty(b);
        done[i] = (e == 0u) && no_equal_neighbours(b);
        legal[i] = (uint8_t)legal_mask(b);
        n_empty[i] = (uint8_t)e;
    }
}

void emu_spawn(uint64_t n, uint8_t *boards, const uint8_t *k, const uint8_t *e, uint8_t *placed)
{
    for (uint64_t i = 0; i < n; ++i) {
        Board b = load(boards + 16 * i);
        const uint32_t empt = empty_mask(b);
        const bool ok = k[i] < ENV2048_POPC(empt);
        const uint32_t bit = ok ? kth_empty(empt, k[i]) : 0u;
        uint32_t p0, p1, p2, p3;
        cell_planes(bit, p0, p1, p2, p3);
        b.r0 |= p0 * e[i];
        b.r1 |= p1 * e[i];
        b.r2 |= p2 * e[i];
        b.r3 |= p3 * e[i];
        store(boards + 16 * i, b);
        placed[i] = ok ? (uint8_t)cell_index(bit) : (uint8_t)0xFF;
    }
}

void emu_normalized_reward(uint64_t n, const uint32_t *reward, float *out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = normalized_reward(reward[i]);
}

void emu_max_exponent(uint64_t n, const uint8_t *boards, uint8_t *out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = (uint8_t)max_exponent(load(boards + 16 * i));
}

// per-line merge scores (line_merge_terms / line_score): `lines` holds n_lines x 4 exponent bytes (element 0 nearest the wall),
// four consecutive lines are evaluated together (byte lane j of the planes = line j); scores[i] = score of line i.
void emu_line_scores(uint64_t n_lines, const uint8_t *lines, uint32_t *scores, uint32_t *group_sums, uint32_t *group_slide_scores)
{
    for (uint64_t g = 0; 4 * g < n_lines; ++g) {
        uint32_t plane[4] = {0u, 0u, 0u, 0u};
        for (uint32_t j = 0; j < 4 && 4 * g + j < n_lines; ++j)
            for (uint32_t k = 0; k < 4; ++k) plane[k] |= (uint32_t)lines[4 * (4 * g + j) + k] << (8 * j);
        uint32_t v1, v2, f1, f2, s4[4];
        line_merge_terms_flags(plane[0], plane[1], plane[2], plane[3], v1, v2, f1, f2);
        line_scores4(v1, v2, f1, f2, s4[0], s4[1], s4[2], s4[3]);  // the SDWA form ...
        for (uint32_t j = 0; j < 4 && 4 * g + j < n_lines; ++j) {
            scores[4 * g + j] = line_score(v1, v2, 8 * j);         // ... and the shift form of the same score must agree
            if (s4[j] != scores[4 * g + j]) scores[4 * g + j] = 0xFFFFFFFFu;
        }
        group_sums[g] = line_score_sum(v1, v2);
        uint32_t a = plane[0], b = plane[1], c = plane[2], d = plane[3], changed;
        group_slide_scores[g] = slide_merge_planes(a, b, c, d, changed);
    }
}

}  // extern "C"
