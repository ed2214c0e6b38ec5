// tests/host_emu/asan_main.cpp -- standalone ASan + UBSan driver for the batch loops of emu.cpp (TEST HARNESS ONLY).
// A shared object loaded through ctypes cannot run under AddressSanitizer without preloading its runtime, so the
// batch functions are exercised here in an ordinary executable, on heap buffers of EXACTLY the required size: any
// out-of-bounds access of the loops over caller buffers (the code the HIP kernels share through board_ops.hpp) aborts.
// Build + run: tests/test_host_emulation.py::test_batch_loops_under_asan_and_ubsan.
#include "emu.cpp"

#include <cstdio>
#include <vector>

int main()
{
    const uint64_t n = 1031;  // odd: no accidental slack from rounding
    std::vector<uint8_t> boards(16 * n), out(16 * n), actions(n), changed(n), done(n), legal(n), n_empty(n), k(n), e(n), placed(n), mx(n);
    std::vector<int32_t> reward(n);
    std::vector<uint32_t> words(n), scores(n);
    std::vector<float> norm(n);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        return x;
    };
    for (uint64_t i = 0; i < 16 * n; ++i) boards[i] = (uint8_t)(next() % 100 < 25 ? 0 : next() % 18);
    for (uint64_t i = 0; i < n; ++i) {
        actions[i] = (uint8_t)next();
        k[i] = (uint8_t)(next() % 17);
        e[i] = (uint8_t)(1 + next() % 2);
        scores[i] = (uint32_t)(next() % (1u << 22));
    }
    emu_afterstate(n, boards.data(), actions.data(), out.data(), reward.data(), changed.data(), 0);
    emu_afterstate(n, boards.data(), actions.data(), out.data(), reward.data(), changed.data(), 1);
    emu_step_words(n, 5, 77, 3, words.data());
    emu_step(n, 5, boards.data(), actions.data(), words.data(), reward.data(), done.data(), legal.data(), 77, 3, 1u);
    emu_step(n, 5, boards.data(), actions.data(), words.data(), reward.data(), done.data(), nullptr, 77, 4, 3u);
    {
        const uint32_t T = 3;
        std::vector<uint8_t> acts(T * n);
        std::vector<uint32_t> w3(T * n), dcount(n);
        std::vector<int64_t> rsum(n);
        for (auto &a : acts) a = (uint8_t)next();
        for (uint32_t t = 0; t < T; ++t) emu_step_words(n, 5, 77, 10 + t, w3.data() + t * n);
        emu_steps_oriented(n, 5, boards.data(), acts.data(), w3.data(), T, rsum.data(), dcount.data(), done.data(), 77, 10, 1u);
        emu_steps_oriented(n, 5, boards.data(), nullptr, w3.data(), T, rsum.data(), dcount.data(), done.data(), 77, 10, 3u);
    }
    emu_status(n, boards.data(), done.data(), legal.data(), n_empty.data());
    emu_spawn(n, boards.data(), k.data(), e.data(), placed.data());
    emu_reset(n, (1ull << 33) + 1, out.data(), 9, 9);
    emu_normalized_reward(n, scores.data(), norm.data());
    emu_max_exponent(n, boards.data(), mx.data());
    uint64_t sum = 0;
    for (uint64_t i = 0; i < 16 * n; ++i) sum = sum * 31 + boards[i] + out[i];
    for (uint64_t i = 0; i < n; ++i) sum = sum * 31 + (uint64_t)reward[i] + done[i] + legal[i] + placed[i] + mx[i];
    std::printf("asan+ubsan ok %llu\n", (unsigned long long)sum);
    return 0;
}
