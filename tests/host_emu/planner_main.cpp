// tests/host_emu/planner_main.cpp -- CPU test driver (g++ -fsanitize=address,undefined) for the two pieces of pure host
// logic behind the C-ABI: the env2048_step_many launch planner (csrc/step_planner.hpp) and the helper thread of
// env2048_step_sequence (csrc/helper_thread.hpp).  TEST HARNESS ONLY -- the product builds the same headers with hipcc.
//
//   planner_main plan  SEED ROUNDS   random record arrays (overlapping arenas, long runs of empty records, all variants)
//                                    -> every plan is checked against a brute-force dependency test
//   planner_main helper              stub jobs: results come back in order, two callers are served one at a time, a forked
//                                    child gets its own helper, ENV2048_HELPER_SPIN_US is honoured
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>

#include <random>
#include <vector>

#include "helper_thread.hpp"
#include "step_planner.hpp"

using namespace env2048;
using namespace env2048::host;

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "%s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                 \
            fprintf(stderr, "\n");                        \
            exit(1);                                      \
        }                                                 \
    } while (0)

static bool overlap(const MemRange &a, const MemRange &b) { return a.lo < b.hi && b.lo < a.hi; }

static bool depend(const env2048_step_args &x, const env2048_step_args &y)
{
    MemRange rx[RANGES_PER_RECORD], ry[RANGES_PER_RECORD];
    const int cx = record_ranges(x, rx), cy = record_ranges(y, ry);
    for (int i = 0; i < cx; ++i)
        for (int j = 0; j < cy; ++j)
            if ((rx[i].write || ry[j].write) && overlap(rx[i], ry[j])) return true;
    return false;
}

static int fuzz_plan(uint64_t seed, int rounds)
{
    std::mt19937_64 rng(seed);
    auto pick = [&](uint64_t m) { return (uint64_t)(rng() % m); };
    uint64_t total_records = 0, total_launches = 0, biggest = 0;
    for (int round = 0; round < rounds; ++round) {
        const uint64_t count = 1 + pick(round % 7 == 0 ? 260 : 90);
        // arenas: a few slots of board memory and output memory that records draw from (so that dependencies do occur)
        const uint64_t slots = 2 + pick(40);
        const bool uniform = round % 13 == 0;  // plain records on slots of their own: long independent runs, the 32-record cap
        const uint64_t max_n = (round % 3 == 0) ? 64 : 1 + pick(5000);
        const uintptr_t B = 0x100000000000ull, A = 0x200000000000ull, R = 0x300000000000ull, D = 0x400000000000ull,
                        Lg = 0x500000000000ull, O = 0x600000000000ull, M = 0x700000000000ull, C = 0x800000000000ull;
        const uint64_t stride = ((round % 5 == 0) ? max_n / 2 + 1 : max_n) * 64;  // (sometimes slots overlap their neighbours)
        std::vector<env2048_step_args> args(count);
        const int empty_mode = (int)pick(5);  // 0 none, 1 sprinkled, 2 long leading run, 3 long runs anywhere, 4 everything empty
        uint64_t run_left = empty_mode == 2 ? 30 + pick(120) : 0;
        for (uint64_t k = 0; k < count; ++k) {
            env2048_step_args a;
            memset(&a, 0, sizeof a);
            const uint64_t s = uniform ? k : pick(slots);
            a.n = 1 + pick(max_n);
            if (round % 4 == 1 || uniform) a.n = max_n;  // equal sizes: nothing is ragged
            bool empty = false;
            if (empty_mode == 1) empty = pick(4) == 0;
            if (empty_mode == 2 || empty_mode == 3) {
                if (run_left) {
                    empty = true;
                    --run_left;
                } else if (empty_mode == 3 && pick(20) == 0) {
                    run_left = 1 + pick(100);
                }
            }
            if (empty_mode == 4) empty = true;
            if (empty) a.n = 0;
            a.env_base = pick(1000);
            a.boards = reinterpret_cast<uint8_t *>(B + s * stride * 16);
            a.actions = reinterpret_cast<const uint8_t *>(A + (uniform || pick(3) ? s : pick(slots)) * stride);
            a.reward = pick(5) ? reinterpret_cast<int32_t *>(R + s * stride * 4) : nullptr;
            a.done = reinterpret_cast<uint8_t *>(D + (uniform || pick(10) ? s : pick(slots)) * stride);
            a.flags = (pick(2) ? ENV2048_FLAG_AUTO_RESET : 0u);
            if (uniform) {
                CHECK(validate_step(a, "fuzz") == ENV2048_OK, "record %llu did not validate: %s", (unsigned long long)k, g_error);
                args[k] = a;
                continue;
            }
            if (pick(6) == 0) a.flags |= ENV2048_FLAG_RANDOM_ACTIONS;
            if (pick(6) == 0) a.legal = reinterpret_cast<uint8_t *>(Lg + s * stride);
            if (pick(8) == 0) {
                a.obs = reinterpret_cast<void *>(O + s * stride * 64);
                if (pick(2)) a.flags |= ENV2048_FLAG_OBS_BF16;
            }
            if (pick(12) == 0) a.max_exp = reinterpret_cast<uint8_t *>(M + s * stride);
            if (pick(15) == 0) a.flags |= pick(2) ? ENV2048_FLAG_TILE_SMALL : ENV2048_FLAG_TILE_LARGE;
            if (pick(25) == 0) a.step_counter = reinterpret_cast<uint64_t *>(C);
            if (pick(30) == 0) a.boards = reinterpret_cast<uint8_t *>(R + s * stride * 4);  // boards aliasing another record's rewards
            CHECK(validate_step(a, "fuzz") == ENV2048_OK, "record %llu did not validate: %s", (unsigned long long)k, g_error);
            args[k] = a;
        }
        std::vector<uint32_t> launch_of(count, 0xFFFFFFFFu);
        std::vector<int32_t> bpl(count, -1), through(count, -1);
        const uint32_t launches = plan_step_many(count, args.data(), launch_of.data(), bpl.data(), through.data());
        // 1. every record got a launch id; ids are non-decreasing, start at 0 and grow by at most one
        uint64_t live_total = 0;
        for (uint64_t k = 0; k < count; ++k) {
            CHECK(launch_of[k] != 0xFFFFFFFFu, "record %llu has no launch", (unsigned long long)k);
            CHECK(bpl[k] == 1 || bpl[k] == BOARDS_PER_LANE, "record %llu: boards_per_lane %d", (unsigned long long)k, bpl[k]);
            CHECK(through[k] == 0 || through[k] == 1, "record %llu: write_through %d", (unsigned long long)k, through[k]);
            if (k) CHECK(launch_of[k] >= launch_of[k - 1] && launch_of[k] <= launch_of[k - 1] + 1, "launch ids jump at %llu", (unsigned long long)k);
            live_total += args[k].n ? 1 : 0;
        }
        CHECK(launch_of[0] == 0, "first launch id %u", launch_of[0]);
        // 2. per launch: at most MAX_FUSED_RECORDS live records, one variant / hint, no counter records in company, no
        //    dependent pair; the number of launches = number of ids that hold a live record
        uint32_t with_live = 0;
        for (uint64_t first = 0; first < count;) {
            uint64_t last = first;
            while (last < count && launch_of[last] == launch_of[first]) ++last;
            std::vector<uint64_t> members;
            for (uint64_t k = first; k < last; ++k)
                if (args[k].n) members.push_back(k);
            CHECK(members.size() <= (size_t)MAX_FUSED_RECORDS, "launch %u has %zu live records", launch_of[first], members.size());
            biggest = members.size() > biggest ? members.size() : biggest;
            if (!members.empty()) ++with_live;
            CHECK(!members.empty() || launches == 0, "a launch id without a live record although the call has work");
            for (size_t i = 0; i < members.size(); ++i) {
                const env2048_step_args &x = args[members[i]];
                if (members.size() > 1) {
                    CHECK(!x.step_counter, "a step_counter record shares launch %u", launch_of[first]);
                    CHECK(variant_of(x) == variant_of(args[members[0]]), "two kernel variants in launch %u", launch_of[first]);
                    CHECK((x.flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE)) ==
                              (args[members[0]].flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE)),
                          "two tiling hints in launch %u", launch_of[first]);
                }
                for (size_t j = i + 1; j < members.size(); ++j)
                    CHECK(!depend(x, args[members[j]]), "records %llu and %llu depend on each other but share launch %u",
                          (unsigned long long)members[i], (unsigned long long)members[j], launch_of[first]);
            }
            first = last;
        }
        CHECK(with_live == launches, "%u launches reported, %u ids hold live records", launches, with_live);
        CHECK(launches <= live_total && (launches > 0) == (live_total > 0), "launch count %u for %llu live records", launches,
              (unsigned long long)live_total);
        total_records += count;
        total_launches += launches;
    }
    printf("plan fuzz ok: %d rounds, %llu records, %llu launches, largest launch %llu records\n", rounds,
           (unsigned long long)total_records, (unsigned long long)total_launches, (unsigned long long)biggest);
    return 0;
}

static int test_interval_set()
{
    IntervalSet s;
    CHECK(s.insert(100, 200) && s.insert(200, 300) && s.n == 1 && s.lo[0] == 100 && s.hi[0] == 300, "touching ranges merge");
    CHECK(!s.meets(300, 400) && !s.meets(0, 100) && s.meets(299, 300) && s.meets(0, 101) && s.meets(150, 160), "meets() is byte overlap");
    CHECK(s.insert(400, 500) && s.insert(600, 700) && s.n == 3, "disjoint ranges stay apart");
    CHECK(s.insert(250, 650) && s.n == 1 && s.lo[0] == 100 && s.hi[0] == 700, "a bridging range swallows its neighbours");
    IntervalSet full;
    for (int i = 0; i < IntervalSet::CAP; ++i) CHECK(full.insert(1000 + 10 * (uintptr_t)i, 1005 + 10 * (uintptr_t)i), "fill %d", i);
    CHECK(full.n == IntervalSet::CAP && !full.insert(5, 6) && full.n == IntervalSet::CAP, "a full set refuses instead of overflowing");
    CHECK(full.insert(1005, 1010) && full.n == IntervalSet::CAP - 1, "a merging insertion still fits a full set");
    printf("interval set ok\n");
    return 0;
}

static int test_helper()
{
    // 1. results come back to the caller that posted them, in order
    // (helpers are never destroyed -- their detached thread lives as long as the process -- so the test keeps them in statics)
    static HelperThread &h = *new HelperThread(nullptr, 50);
    std::vector<int> seen;
    for (int k = 0; k < 200; ++k) {
        std::lock_guard<std::mutex> call(h.call_mutex);
        h.post([&seen, k]() -> int {
            seen.push_back(k);
            return 1000 + k;
        });
        CHECK(h.wait() == 1000 + k, "job %d returned someone else's result", k);
    }
    CHECK(seen.size() == 200 && h.jobs() == 200, "jobs served: %zu / %llu", seen.size(), (unsigned long long)h.jobs());
    for (int k = 0; k < 200; ++k) CHECK(seen[k] == k, "job order");
    // 2. two callers at once: each holds call_mutex across post .. wait, so jobs never interleave and nobody sees the other's result
    std::atomic<int> inside{0}, overlaps{0};
    auto caller = [&](int id) {
        for (int k = 0; k < 300; ++k) {
            std::lock_guard<std::mutex> call(h.call_mutex);
            h.post([&, id, k]() -> int {
                if (inside.fetch_add(1) != 0) overlaps.fetch_add(1);
                std::this_thread::sleep_for(std::chrono::microseconds(k % 7));
                inside.fetch_sub(1);
                return id * 100000 + k;
            });
            CHECK(h.wait() == id * 100000 + k, "caller %d got a foreign result at job %d", id, k);
        }
    };
    std::thread t1(caller, 1), t2(caller, 2);
    t1.join();
    t2.join();
    CHECK(overlaps.load() == 0 && h.jobs() == 800, "jobs overlapped (%d) or were lost (%llu)", overlaps.load(), (unsigned long long)h.jobs());
    // 3. the poll window: with spin 0 the thread sleeps at once and burns (almost) nothing while idle; with a long window it burns the window
    static HelperThread &sleeper = *new HelperThread(nullptr, 0), &spinner = *new HelperThread(nullptr, 20000);
    for (HelperThread *t : {&sleeper, &spinner}) {
        std::lock_guard<std::mutex> call(t->call_mutex);
        t->post([]() -> int { return 0; });
        CHECK(t->wait() == 0, "warm-up job");
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(60));
    {
        std::lock_guard<std::mutex> call(sleeper.call_mutex);
        sleeper.post([]() -> int { return 0; });
        sleeper.wait();
    }
    {
        std::lock_guard<std::mutex> call(spinner.call_mutex);
        spinner.post([]() -> int { return 0; });
        spinner.wait();
    }
    CHECK(sleeper.spin_us() == 0 && spinner.spin_us() == 20000, "spin windows");
    CHECK(sleeper.cpu_ns() < 5000000ull, "a helper with spin 0 used %llu ns of CPU for two empty jobs", (unsigned long long)sleeper.cpu_ns());
    // (the window is wall-clock: on a loaded machine the spinning thread is preempted and burns less than the 20 ms it spans)
    CHECK(spinner.cpu_ns() > 2000000ull && spinner.cpu_ns() > 2 * sleeper.cpu_ns(), "a helper with a 20 ms window only used %llu ns (spin 0: %llu ns)",
          (unsigned long long)spinner.cpu_ns(), (unsigned long long)sleeper.cpu_ns());
    // 4. fork: the child's registry slot is rebuilt (the parent's thread does not exist there) and works
    static PerDeviceRegistry<HelperThread> registry;
    int made = 0;
    auto make = [&] {
        ++made;
        return new HelperThread(nullptr, 0);
    };
    HelperThread *parent = registry.get(3, make);
    CHECK(parent && registry.get(3, make) == parent && made == 1 && registry.peek(3) == parent && !registry.peek(4), "registry caches per slot");
    CHECK(!registry.get(-1, make) && !registry.get(64, make), "slot range");
    {
        std::lock_guard<std::mutex> call(parent->call_mutex);
        parent->post([]() -> int { return 7; });
        CHECK(parent->wait() == 7, "parent helper");
    }
    fflush(stdout);
    const pid_t pid = fork();
    CHECK(pid >= 0, "fork failed");
    if (pid == 0) {
        if (registry.peek(3) != nullptr) _exit(11);  // the parent's helper must not be visible in the child
        HelperThread *mine = registry.get(3, make);
        if (!mine || mine == parent || mine->owner() != getpid()) _exit(12);
        std::lock_guard<std::mutex> call(mine->call_mutex);
        mine->post([]() -> int { return 42; });
        _exit(mine->wait() == 42 ? 0 : 13);
    }
    int status = 0;
    CHECK(waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0, "forked child failed: status 0x%x", status);
    CHECK(registry.get(3, make) == parent, "the parent keeps its helper after a fork");
    printf("helper thread ok: cpu_ns idle spin0 %llu, spin20ms %llu\n", (unsigned long long)sleeper.cpu_ns(), (unsigned long long)spinner.cpu_ns());
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "plan")) {
        if (test_interval_set()) return 1;
        return fuzz_plan(argc > 2 ? strtoull(argv[2], nullptr, 0) : 1, argc > 3 ? atoi(argv[3]) : 2000);
    }
    if (argc >= 2 && !strcmp(argv[1], "helper")) return test_helper();
    fprintf(stderr, "usage: planner_main plan [SEED [ROUNDS]] | helper\n");
    return 2;
}
