// Synchronisation of the per-device host threads of a multi-GPU call (run_multi in engine.hip; one thread + one
// engine per device where the reference has one OpenMP team, src/pairwise.cpp:199-203).  Host-only C++17: also
// compiled by the sanitizer build, whose check program injects a failure at every stage (tests/test_host_hardening_cpu.py).
#ifndef KSPIDER_HOST_SYNC_H
#define KSPIDER_HOST_SYNC_H
#include <atomic>
#include <condition_variable>
#include <mutex>

namespace ksp {

// A barrier that also carries the "somebody failed" decision.  The decision is taken ONCE per barrier generation, by
// the thread that arrives last, under the mutex; every thread of that generation returns that same value.  (Reading a
// shared flag after leaving a plain barrier is a race: a thread released early can run on, fail in the next stage and
// raise the flag before a slower thread of the previous generation samples it — the slow thread then leaves while the
// fast one waits at the next barrier for an arrival that never comes.)
class FailBarrier {
public:
    explicit FailBarrier(int n) : n_(n) {}
    void fail() { failed_.store(1, std::memory_order_release); }
    // (stage-local hint, e.g. to skip work that would only fail again; never a reason to skip a sync())
    bool failed_hint() const { return failed_.load(std::memory_order_acquire) != 0; }
    // arrive and wait for the others; true = some thread had failed before the last one arrived: every thread of
    // this generation sees true and returns from its body
    bool sync() {
        std::unique_lock<std::mutex> l(mu_);
        const unsigned g = gen_;
        if (++waiting_ == n_) {
            waiting_ = 0;
            latched_ = failed_.load(std::memory_order_acquire) != 0;
            ++gen_;
            cv_.notify_all();
            return latched_;
        }
        cv_.wait(l, [&] { return gen_ != g; });
        // (latched_ cannot be overwritten before this thread reads it: the next generation completes only after
        //  every thread, this one included, has arrived again)
        return latched_;
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    const int n_;
    int waiting_ = 0;
    unsigned gen_ = 0;
    bool latched_ = false;
    std::atomic<int> failed_{0};
};

}  // namespace ksp
#endif
