// step_planner.hpp -- host-side launch planning of env2048_step_many: which consecutive records may share one launch.
// [step-kernel-source-begin]
// Plain C++ (no HIP): the same header is compiled by hipcc into libenv2048.so and by g++ into the CPU planner tests
// (tests/host_emu), where it is fuzzed against a brute-force dependency check.
//
// The call's contract is "the records, in order".  Two records depend on each other iff one WRITES memory the other reads or
// writes (boards are read and written; actions are read; reward / done / legal / obs / max_exp are written); records that
// touch disjoint memory give the same bytes in any order, so a run of consecutive, pairwise independent records may execute
// concurrently -- as one step_many_kernel launch.  The policy is a function of the argument records alone (addresses, sizes,
// flags): no caller hint, no environment variable, no timing.
#pragma once
#include <stdint.h>

#include "errors.hpp"

namespace env2048 {
namespace host {

inline int validate_step(const env2048_step_args &a, const char *who)
{
    if (a.flags & ~STEP_FLAGS) return fail(ENV2048_ERR_INVALID_ARG, "%s: unknown flag bits 0x%x", who, a.flags);
    if ((a.flags & ENV2048_FLAG_OBS_F16) && (a.flags & ENV2048_FLAG_OBS_BF16))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: OBS_F16 and OBS_BF16 exclude each other", who);
    if ((a.flags & ENV2048_FLAG_STORES_WRITE_THROUGH) && (a.flags & ENV2048_FLAG_STORES_WRITE_BACK))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: STORES_WRITE_THROUGH and STORES_WRITE_BACK exclude each other", who);
    if ((a.flags & ENV2048_FLAG_TILE_SMALL) && (a.flags & ENV2048_FLAG_TILE_LARGE))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: TILE_SMALL and TILE_LARGE exclude each other", who);
    if (a.reserved) return fail(ENV2048_ERR_INVALID_ARG, "%s: reserved must be 0", who);
    if ((a.flags & ENV2048_FLAG_REWARD_LOG2_F32) && !a.reward)
        return fail(ENV2048_ERR_INVALID_ARG, "%s: REWARD_LOG2_F32 without a reward array", who);
    if (a.n == 0) return ENV2048_OK;
    if (!a.boards || !a.done) return fail(ENV2048_ERR_INVALID_ARG, "%s: boards and done are required", who);
    if (!a.actions && !(a.flags & ENV2048_FLAG_RANDOM_ACTIONS))
        return fail(ENV2048_ERR_INVALID_ARG, "%s: actions is NULL without ENV2048_FLAG_RANDOM_ACTIONS", who);
    if (misaligned(a.boards, 16)) return fail(ENV2048_ERR_ALIGNMENT, "%s: boards must be 16-byte aligned", who);
    if (misaligned(a.reward, 4)) return fail(ENV2048_ERR_ALIGNMENT, "%s: reward must be 4-byte aligned", who);
    if (misaligned(a.obs, 16)) return fail(ENV2048_ERR_ALIGNMENT, "%s: obs must be 16-byte aligned", who);
    if (misaligned(a.step_counter, 8)) return fail(ENV2048_ERR_ALIGNMENT, "%s: step_counter must be 8-byte aligned", who);
    if (int rc = check_range(a.n, a.env_base)) return rc;
    return check_count(a.n);
}

struct MemRange {
    uintptr_t lo, hi;
    bool write;
};
constexpr int RANGES_PER_RECORD = 7;

// The byte ranges one record touches (empty arrays and absent outputs contribute nothing); returns their number.
inline int record_ranges(const env2048_step_args &a, MemRange *out)
{
    int c = 0;
    auto add = [&](const void *p, uint64_t bytes, bool write) {
        if (p && bytes) out[c++] = MemRange{reinterpret_cast<uintptr_t>(p), reinterpret_cast<uintptr_t>(p) + bytes, write};
    };
    add(a.boards, 16 * a.n, true);
    if (!(a.flags & ENV2048_FLAG_RANDOM_ACTIONS)) add(a.actions, a.n, false);
    add(a.reward, 4 * a.n, true);
    add(a.done, a.n, true);
    add(a.legal, a.n, true);
    add(a.obs, a.n * 16u * (obs_format(a.flags) == OBS_F32 ? 4u : 2u), true);
    add(a.max_exp, a.n, true);
    return c;
}

// A set of byte ranges kept as sorted, pairwise disjoint, non-touching intervals (overlapping AND touching insertions are
// merged, so the contiguous sub-buffers of an arena collapse into one entry), which makes "does [lo, hi) meet the set" one
// binary search.  Fixed capacity: a group has at most MAX_FUSED_RECORDS records of RANGES_PER_RECORD ranges; insert()
// refuses (returns false, set unchanged) rather than write past the arrays.
struct IntervalSet {
    static constexpr int CAP = MAX_FUSED_RECORDS * RANGES_PER_RECORD;
    uintptr_t lo[CAP], hi[CAP];
    int n = 0;
    int first_ending_at_or_after(uintptr_t x) const  // smallest i with hi[i] >= x (hi is sorted because the intervals are disjoint)
    {
        int a = 0, b = n;
        while (a < b) {
            const int m = (a + b) / 2;
            if (hi[m] >= x)
                b = m;
            else
                a = m + 1;
        }
        return a;
    }
    bool meets(uintptr_t l, uintptr_t h) const  // shares at least one byte with [l, h)
    {
        int i = first_ending_at_or_after(l);
        if (i < n && hi[i] == l) ++i;  // merely touching from below
        return i < n && lo[i] < h;
    }
    bool insert(uintptr_t l, uintptr_t h)
    {
        const int i = first_ending_at_or_after(l);
        int j = i;  // intervals [i, j) overlap or touch [l, h): swallow them
        while (j < n && lo[j] <= h) {
            l = lo[j] < l ? lo[j] : l;
            h = hi[j] > h ? hi[j] : h;
            ++j;
        }
        if (j == i) {  // nothing swallowed: open a slot
            if (n == CAP) return false;
            for (int k = n; k > i; --k) {
                lo[k] = lo[k - 1];
                hi[k] = hi[k - 1];
            }
            ++n;
        } else if (j > i + 1) {  // several swallowed: close the gap
            const int drop = j - i - 1;
            for (int k = i + 1; k + drop < n; ++k) {
                lo[k] = lo[k + drop];
                hi[k] = hi[k + drop];
            }
            n -= drop;
        }
        lo[i] = l;
        hi[i] = h;
        return true;
    }
};

struct StepVariant {
    bool extras, random_actions;
    bool operator==(const StepVariant &o) const { return extras == o.extras && random_actions == o.random_actions; }
};

inline StepVariant variant_of(const env2048_step_args &a)
{
    return StepVariant{a.legal || a.obs || a.max_exp || (a.flags & ENV2048_FLAG_REWARD_LOG2_F32) != 0u, (a.flags & ENV2048_FLAG_RANDOM_ACTIONS) != 0u};
}

// The launches of an env2048_step_many call, in order: fn(first, last, lead, live, total) for every run of records
// [first, last) that goes out together; live = its records with n > 0 (0 only when the whole call holds no work: then the
// single run covers everything and nothing is launched), lead = the first of them, total = their boards.  The runs tile
// [0, count) without gaps.  Greedy, order-preserving, one pass with O(1) state per record, so it terminates on any input:
// a live record joins the open run iff the run has fewer than MAX_FUSED_RECORDS live records, neither reads its step index
// from device memory, both ask for the same kernel variant and tiling hint, the run stays within 2x of its smallest member's
// size (surplus blocks of a ragged 2-D grid are cheap, not free; sizes within one block never count as ragged), and the
// record is independent of every member.  Empty records (n == 0) enqueue nothing and decide nothing: they stay in whatever
// run is open when they come by (at the very start: the first run).
template <typename Fn>
void for_each_launch(uint64_t count, const env2048_step_args *args, Fn fn)
{
    IntervalSet written, read;  // what the open run's members write / only read
    MemRange ranges[RANGES_PER_RECORD];
    auto join = [&](const env2048_step_args &a, bool check) {
        const int c = record_ranges(a, ranges);
        if (check)
            for (int i = 0; i < c; ++i)
                if (written.meets(ranges[i].lo, ranges[i].hi) || (ranges[i].write && read.meets(ranges[i].lo, ranges[i].hi))) return false;
        // (cannot run out of slots: <= MAX_FUSED_RECORDS members x RANGES_PER_RECORD ranges = CAP; checked all the same)
        for (int i = 0; i < c; ++i)
            if (!(ranges[i].write ? written : read).insert(ranges[i].lo, ranges[i].hi)) return false;
        return true;
    };
    uint64_t first = 0, lead = 0, live = 0, total = 0, lo_n = 0, hi_n = 0;
    bool fusable = false;
    StepVariant variant{false, false};
    uint32_t tile_hint = 0;
    for (uint64_t k = 0; k < count; ++k) {
        const env2048_step_args &a = args[k];
        if (a.n == 0) continue;
        if (live) {
            const uint64_t new_lo = a.n < lo_n ? a.n : lo_n, new_hi = a.n > hi_n ? a.n : hi_n;
            const bool joins = fusable && live < (uint64_t)MAX_FUSED_RECORDS && !a.step_counter && variant_of(a) == variant &&
                               (a.flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE)) == tile_hint &&
                               !(new_hi > 2 * new_lo && new_hi > (uint64_t)block_boards(1)) && join(a, true);
            if (joins) {
                lo_n = new_lo;
                hi_n = new_hi;
                ++live;
                total += a.n;
                continue;
            }
            fn(first, k, lead, live, total);
            first = k;
        }
        lead = k;  // this record opens a run
        live = 1;
        total = lo_n = hi_n = a.n;
        variant = variant_of(a);
        tile_hint = a.flags & (ENV2048_FLAG_TILE_SMALL | ENV2048_FLAG_TILE_LARGE);
        fusable = !a.step_counter;
        written.n = read.n = 0;
        if (fusable) join(a, false);
    }
    if (first < count) fn(first, count, live ? lead : first, live, total);
}

// env2048_step_many_plan without its argument validation: launch_of[k] = index of the launch record k rides in (for an empty
// record: the launch of the run it sits in), the tiling / store policy that launch gets, and the number of launches.
inline uint32_t plan_step_many(uint64_t count, const env2048_step_args *args, uint32_t *launch_of, int32_t *boards_per_lane,
                               int32_t *write_through)
{
    uint32_t launches = 0;
    for_each_launch(count, args, [&](uint64_t first, uint64_t last, uint64_t lead, uint64_t live, uint64_t total) {
        for (uint64_t r = first; r < last; ++r) {
            launch_of[r] = launches;
            if (boards_per_lane) boards_per_lane[r] = plan_small_tiling(total, args[lead].flags) ? 1 : BOARDS_PER_LANE;
            if (write_through) write_through[r] = plan_write_through(total, args[r].n ? args[r].flags : args[lead].flags) ? 1 : 0;
        }
        if (live) ++launches;
    });
    return launches;
}

}  // namespace host
}  // namespace env2048
// [step-kernel-source-end]
