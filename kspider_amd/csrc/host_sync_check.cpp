// Check program of host_sync.h (make asan): the thread / barrier skeleton of run_multi (engine.hip) — every worker
// runs the same sequence of stages with a sync point behind each — with a failure injected into one worker at one
// stage, the others delayed at random so that fast threads run ahead of slow ones.  Every run must end with all
// workers returned (no thread left waiting at a barrier) and with the failure reported; a watchdog turns a hang into
// a non-zero exit.  tests/test_host_hardening_cpu.py runs it.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "host_sync.h"

static int run_once(int nd, int stages, int fail_thread, int fail_stage, unsigned seed) {
    ksp::FailBarrier bar(nd);
    std::vector<int> rc((size_t)nd, 0), reached((size_t)nd, 0);
    auto body = [&](int i) {
        std::mt19937 rng(seed * 131u + (unsigned)i);
        for (int s = 0; s < stages; ++s) {
            // the stage's work: skipped when somebody is already known to have failed (as run_multi does)
            if (!bar.failed_hint()) {
                if ((rng() & 3u) == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));
                if (i == fail_thread && s == fail_stage) { rc[(size_t)i] = 7; bar.fail(); }
            }
            reached[(size_t)i] = s + 1;
            if (bar.sync()) return;
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < nd; ++i) th.emplace_back(body, i);
    for (auto& t : th) t.join();
    int any = 0;
    for (int i = 0; i < nd; ++i) any |= rc[(size_t)i];
    // everybody leaves at the same sync point: the one behind the failing stage
    for (int i = 0; i < nd; ++i) {
        const int want = fail_thread < 0 ? stages : fail_stage + 1;
        if (reached[(size_t)i] != want) { std::printf("worker %d left after stage %d, expected %d\n", i, reached[(size_t)i], want); return -1; }
    }
    return any;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20;
    std::atomic<int> done{0};
    std::thread watchdog([&] {
        for (int t = 0; t < 600 && !done.load(); ++t) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!done.load()) { std::printf("HANG: a worker is still waiting at a barrier\n"); std::fflush(stdout); std::_Exit(3); }
    });
    int bad = 0, runs = 0;
    const int stages = 12;   // (run_multi has up to 12 sync points on the sketch path)
    for (int r = 0; r < rounds && !bad; ++r)
        for (int nd : {1, 2, 3, 5, 8}) {
            if (run_once(nd, stages, -1, -1, (unsigned)r) != 0) { bad = 1; break; }   // no failure: all stages run
            ++runs;
            for (int s = 0; s < stages && !bad; ++s) {
                const int ft = (r + s) % nd;
                const int got = run_once(nd, stages, ft, s, (unsigned)(r * 97 + s));
                if (got != 7) { std::printf("nd %d stage %d thread %d: rc %d\n", nd, s, ft, got); bad = 1; }
                ++runs;
            }
        }
    done.store(1);
    watchdog.join();
    std::printf("host_sync %s: %d runs\n", bad ? "FAILED" : "ok", runs);
    return bad;
}
