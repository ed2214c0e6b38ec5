// tests/host_emu/emu_nxn.cpp -- CPU build of simulate_2048_amd/csrc/nxn_ops.hpp (ENV2048_HOST_EMULATION).
// TEST HARNESS ONLY: lets the register-resident arithmetic of the general-size kernels (boards S x S, 2 <= S <= 8) be
// checked against the oracle for every size without a GPU (tests/test_host_emulation_nxn.py, UBSan on the shared object).
// Never loaded by the product.
#define ENV2048_HOST_EMULATION 1
#include "nxn_ops.hpp"

#include <cstring>

using namespace env2048;

namespace {

template <int S>
nxn::Rows<S> load(const uint8_t *p)
{
    uint32_t dense[nxn::Shape<S>::DENSE] = {};
    std::memcpy(dense, p, nxn::Shape<S>::CELLS);
    nxn::Rows<S> b;
    nxn::rows_from_dense<S>(dense, b);
    return b;
}

template <int S>
void store(uint8_t *p, const nxn::Rows<S> &b)
{
    uint32_t dense[nxn::Shape<S>::DENSE];
    nxn::dense_from_rows<S>(b, dense);
    std::memcpy(p, dense, nxn::Shape<S>::CELLS);
}

template <int S>
void afterstate_all(uint64_t n, const uint8_t *in, const uint8_t *actions, uint8_t *out, int32_t *reward, uint8_t *changed)
{
    for (uint64_t i = 0; i < n; ++i) {
        nxn::Rows<S> b = load<S>(in + i * S * S);
        uint32_t score;
        bool ch;
        nxn::afterstate<S>(b, actions[i] & 3u, score, ch);
        store<S>(out + i * S * S, b);
        reward[i] = (int32_t)score;
        changed[i] = ch;
    }
}

template <int S>
void step_all(uint64_t n, uint64_t env_base, uint8_t *boards, const uint8_t *actions, int32_t *reward, uint8_t *done, uint8_t *legal,
              uint64_t seed, uint64_t step, uint32_t flags)
{
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    for (uint64_t i = 0; i < n; ++i) {
        nxn::Rows<S> b = load<S>(boards + i * S * S);
        uint32_t r, lg;
        bool d;
        nxn::step_board<S>(b, key, env_base + i, actions ? actions[i] : 0u, (flags & 2u) != 0u, (flags & 1u) != 0u, r, d, lg);
        store<S>(boards + i * S * S, b);
        reward[i] = (int32_t)r;
        done[i] = d;
        legal[i] = (uint8_t)lg;
    }
}

template <int S>
void status_all(uint64_t n, const uint8_t *boards, uint8_t *done, uint8_t *legal, uint8_t *n_empty, uint8_t *max_exp)
{
    for (uint64_t i = 0; i < n; ++i) {
        const nxn::Rows<S> b = load<S>(boards + i * S * S);
        bool d;
        uint32_t lg, ne;
        nxn::status<S>(b, d, lg, ne);
        done[i] = d;
        legal[i] = (uint8_t)lg;
        n_empty[i] = (uint8_t)ne;
        max_exp[i] = (uint8_t)nxn::max_exponent<S>(b);
    }
}

template <int S>
void place_all(uint64_t n, uint8_t *boards, const uint8_t *k, const uint8_t *e)
{
    for (uint64_t i = 0; i < n; ++i) {
        nxn::Rows<S> b = load<S>(boards + i * S * S);
        nxn::place_kth<S>(b, k[i], e[i], true);
        store<S>(boards + i * S * S, b);
    }
}

template <int S>
void fresh_all(uint64_t n, uint64_t env_base, uint64_t seed, uint64_t step, uint8_t *boards)
{
    const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    for (uint64_t i = 0; i < n; ++i) {
        nxn::Rows<S> b;
        uint32_t w0, w1, w2, w3;
        env_block(key, env_base + i, STREAM_RESET, w0, w1, w2, w3);
        nxn::fresh<S>(b, w0, w1, w2, w3);
        store<S>(boards + i * S * S, b);
    }
}

}  // namespace

#define DISPATCH(call)         \
    switch (size) {            \
    case 2: call(2); break;    \
    case 3: call(3); break;    \
    case 4: call(4); break;    \
    case 5: call(5); break;    \
    case 6: call(6); break;    \
    case 7: call(7); break;    \
    case 8: call(8); break;    \
    default: return -1;        \
    }                          \
    return 0;

extern "C" {

int emu_nxn_afterstate(uint64_t n, int size, const uint8_t *in, const uint8_t *actions, uint8_t *out, int32_t *reward, uint8_t *changed)
{
#define CALL(S) afterstate_all<S>(n, in, actions, out, reward, changed)
    DISPATCH(CALL)
#undef CALL
}

int emu_nxn_step(uint64_t n, int size, uint64_t env_base, uint8_t *boards, const uint8_t *actions, int32_t *reward, uint8_t *done,
                 uint8_t *legal, uint64_t seed, uint64_t step, uint32_t flags)
{
#define CALL(S) step_all<S>(n, env_base, boards, actions, reward, done, legal, seed, step, flags)
    DISPATCH(CALL)
#undef CALL
}

int emu_nxn_status(uint64_t n, int size, const uint8_t *boards, uint8_t *done, uint8_t *legal, uint8_t *n_empty, uint8_t *max_exp)
{
#define CALL(S) status_all<S>(n, boards, done, legal, n_empty, max_exp)
    DISPATCH(CALL)
#undef CALL
}

int emu_nxn_place(uint64_t n, int size, uint8_t *boards, const uint8_t *k, const uint8_t *e)
{
#define CALL(S) place_all<S>(n, boards, k, e)
    DISPATCH(CALL)
#undef CALL
}

int emu_nxn_reset(uint64_t n, int size, uint64_t env_base, uint64_t seed, uint64_t step, uint8_t *boards)
{
#define CALL(S) fresh_all<S>(n, env_base, seed, step, boards)
    DISPATCH(CALL)
#undef CALL
}

}  // extern "C"
