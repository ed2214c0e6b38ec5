// helper_thread.hpp -- the one extra host thread per device that enqueues the second chain of env2048_step_sequence.
// Plain C++ (no HIP): the thread runs whatever job it is handed, so the CPU tests drive it with stub jobs.
//
// Life cycle: created on the first split call for its device, detached, never destroyed (it may be blocked in wait() while
// static destructors run); a forked child does not inherit threads, so the owner pid is kept and the child creates its own.
// One job at a time: the caller holds `call_mutex` from posting a job until it has waited for it, so concurrent callers of the
// same device are served one after the other, each seeing only its own job's result.
//
// Poll window: calls come in bursts (a stepping loop), and a futex wake-up costs more than a whole 2^20-board step, so after a
// job the helper keeps polling for the next one for `spin_us` microseconds before it goes to sleep on the condition variable.
// ENV2048_HELPER_SPIN_US (read once, when the first helper of the process is created) sets that window: default 300, 0 =
// sleep at once (a CPU-pinned caller that cannot spare the core), at most 100 000.  While polling the thread executes `pause`
// and burns one hardware thread; `cpu_ns()` reports what it has consumed so far (CLOCK_THREAD_CPUTIME_ID).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace env2048 {
namespace host {

inline uint32_t helper_spin_us_from_env()
{
    static const uint32_t value = [] {
        const char *s = getenv("ENV2048_HELPER_SPIN_US");
        if (!s || !*s) return 300u;
        char *end = nullptr;
        const long v = strtol(s, &end, 10);
        if (end == s || v < 0) return 300u;
        return v > 100000 ? 100000u : (uint32_t)v;
    }();
    return value;
}

class HelperThread {
public:
    std::mutex call_mutex;  // held by a caller across post() .. wait(): one job at a time

    // `on_start` runs once in the new thread before the first job (hipSetDevice there).
    explicit HelperThread(std::function<void()> on_start, uint32_t spin_us = helper_spin_us_from_env())
        : spin_us_(spin_us), owner_(getpid())
    {
        std::thread([this, on_start = std::move(on_start)] {
            if (on_start) on_start();
            loop();
        }).detach();
    }
    HelperThread(const HelperThread &) = delete;
    HelperThread &operator=(const HelperThread &) = delete;

    pid_t owner() const { return owner_; }
    uint32_t spin_us() const { return spin_us_; }
    uint64_t jobs() const { return jobs_.load(std::memory_order_relaxed); }
    uint64_t cpu_ns() const { return cpu_ns_.load(std::memory_order_relaxed); }  // as of the end of the last job / poll window

    // job() returns an int the caller gets back from wait() (a hipError_t in the library).  The job may refer to the caller's
    // stack: the caller does not return before wait().
    void post(std::function<int()> job)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = std::move(job);
            has_job_ = true;
            job_done_ = false;
            posted_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return job_done_; });
        return job_result_;
    }

private:
    static uint64_t thread_cpu_ns()
    {
        timespec ts;
        if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) != 0) return 0;
        return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
    }
    void loop()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            if (!has_job_ && spin_us_) {
                lk.unlock();
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us_);
                while (!posted_.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) cpu_relax();
                cpu_ns_.store(thread_cpu_ns(), std::memory_order_relaxed);
                lk.lock();
            }
            cv_.wait(lk, [this] { return has_job_; });
            posted_.store(false, std::memory_order_relaxed);
            std::function<int()> job = std::move(job_);
            has_job_ = false;
            lk.unlock();
            const int result = job();
            jobs_.fetch_add(1, std::memory_order_relaxed);
            cpu_ns_.store(thread_cpu_ns(), std::memory_order_relaxed);
            lk.lock();
            job_result_ = result;
            job_done_ = true;
            cv_.notify_all();
        }
    }
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

    const uint32_t spin_us_;
    const pid_t owner_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool has_job_ = false, job_done_ = false;
    int job_result_ = 0;
    std::atomic<bool> posted_{false};  // set with has_job_: what the helper polls while it is still warm
    std::atomic<uint64_t> jobs_{0}, cpu_ns_{0};
};

// Per-slot registry (slot = device ordinal): the helper of `slot`, created by make() on first use and again in a forked child.
template <typename T, int SLOTS = 64>
class PerDeviceRegistry {
public:
    template <typename Make>
    T *get(int slot, Make make)
    {
        if (slot < 0 || slot >= SLOTS) return nullptr;
        std::lock_guard<std::mutex> lock(m_);
        if (table_[slot] && owner_[slot] != getpid()) table_[slot] = nullptr;  // forked child: the parent's helper thread does not exist here
        if (!table_[slot]) {
            table_[slot] = make();
            owner_[slot] = getpid();
        }
        return table_[slot];
    }
    T *peek(int slot)
    {
        if (slot < 0 || slot >= SLOTS) return nullptr;
        std::lock_guard<std::mutex> lock(m_);
        return table_[slot] && owner_[slot] == getpid() ? table_[slot] : nullptr;
    }

private:
    std::mutex m_;
    T *table_[SLOTS] = {};
    pid_t owner_[SLOTS] = {};
};

}  // namespace host
}  // namespace env2048
