// pool_tsan.cpp -- sbe_host::StepPool (sbayes_amd/csrc/sbe_pool.h, the host worker pool of sbe_step_batch) under
// ThreadSanitizer on the CPU.  No HIP, no GPU:
//     g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tests/c/pool_tsan.cpp -o pool_tsan && ./pool_tsan 10000
// N generations of 1..64 items from one caller thread, with and without the caller's poll hook, jobs of uneven length,
// idle gaps of every kind between generations (none: the workers are still polling; short: inside their 400 us polling
// window; long: they sleep on the condition variable and wake late), and pools of 1..8 threads destroyed both while
// their workers poll and after they fell asleep.  Every item's result is checked; ThreadSanitizer reports go to stderr
// and make the exit code non-zero (TSAN_OPTIONS=halt_on_error=1 exitcode=66 by the test).
#include "../../sbayes_amd/csrc/sbe_pool.h"

#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
    const int generations = argc > 1 ? std::atoi(argv[1]) : 10000;
    const int rounds = 25;
    std::mt19937 rng(12345);
    long items = 0, polls = 0;
    for (int round = 0; round < rounds; ++round) {
        const int n_workers = (int)(rng() % 8);                      // 0 workers: the caller does everything
        sbe_host::StepPool pool(n_workers);
        std::vector<long> out(64);
        for (int g = 1; g <= generations / rounds; ++g) {
            const int n = 1 + (int)(rng() % 64);
            std::fill(out.begin(), out.end(), -1L);
            long poll_calls = 0;
            const std::function<void()> poll = [&] { ++poll_calls; };    // (called by the calling thread only)
            const bool use_poll = (rng() & 1u) != 0;
            pool.run(n, [&](int i) {
                volatile int x = 0;
                const int spin = (i * 7919 + g) % 300;
                for (int k = 0; k < spin; ++k) x = x + k;
                if ((i + g) % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
                out[i] = (long)i * g + round;
            }, use_poll ? &poll : nullptr);
            for (int i = 0; i < n; ++i)
                if (out[i] != (long)i * g + round) { std::fprintf(stderr, "pool_tsan: item %d of generation %d not done\n", i, g); return 1; }
            for (int i = n; i < 64; ++i)
                if (out[i] != -1L) { std::fprintf(stderr, "pool_tsan: item %d beyond n = %d was run\n", i, n); return 1; }
            items += n; polls += poll_calls;
            switch (rng() % 10) {
                case 0: std::this_thread::sleep_for(std::chrono::microseconds(60)); break;     // inside the polling window
                case 1: std::this_thread::sleep_for(std::chrono::microseconds(900)); break;    // workers asleep: late wake-ups
                default: break;                                                                 // back to back
            }
        }
        if (round & 1) std::this_thread::sleep_for(std::chrono::milliseconds(2));               // destroy mid-idle (asleep)
    }                                                                                           // else: destroy while they poll
    std::printf("pool_tsan: %d generations in %d pools, %ld items, %ld poll calls: ok\n", generations, rounds, items, polls);
    return 0;
}
