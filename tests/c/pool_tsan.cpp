// pool_tsan.cpp -- sbe_host::StepPool (sbayes_amd/csrc/sbe_pool.h, the host worker pool of sbe_step_batch) under
// ThreadSanitizer on the CPU.  No HIP, no GPU:
//     g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tests/c/pool_tsan.cpp -o pool_tsan && ./pool_tsan 10000
// N generations of 1..64 items from one caller thread, with and without the caller's poll hook, jobs of uneven length,
// idle gaps of every kind between generations (none: the workers are still polling; short: inside their 400 us polling
// window; long: they sleep on the condition variable and wake late), and pools of 1..8 threads destroyed both while
// their workers poll and after they fell asleep.  Every item's result is checked; ThreadSanitizer reports go to stderr
// and make the exit code non-zero (TSAN_OPTIONS=halt_on_error=1 exitcode=66 by the test).
#include "../../sbayes_amd/csrc/sbe_pool.h"

#include <algorithm>
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
    // ---- run_as_chunks_land: the result lands chunk by chunk, each chunk with its own flag, in ANY order (a kernel storing
    // into host-mapped memory); every thread reads the flags itself.  Producer threads stand in for the device: each fills
    // its chunks (in a shuffled order) and raises their flags; one flag in a while is never raised and the caller's tick
    // -- the engine's stream synchronisation -- declares everything landed instead (after joining the producers).
    long chunk_jobs = 0, ticks = 0;
    for (int round = 0; round < rounds; ++round) {
        const int n_workers = (int)(rng() % 8);
        sbe_host::StepPool pool(n_workers);
        for (int g = 1; g <= std::max(4, generations / rounds / 8); ++g) {
            const int n_chunks = 1 + (int)(rng() % 16);
            const size_t chunk = 128 * (1 + rng() % 16), bytes = n_chunks * chunk - rng() % 100, job = 64 * (1 + rng() % 12);
            const int n_jobs = (int)((bytes + job - 1) / job);
            std::vector<unsigned char> stage(bytes, 0xEE), dst(bytes, 0);
            std::vector<std::atomic<unsigned long long>> flags(n_chunks);
            for (auto& f : flags) f.store(0);
            const unsigned long long seq = (unsigned long long)g + 1000ull * round + 1;
            const unsigned char tag = (unsigned char)(1 + (g + round) % 200);
            std::vector<int> order(n_chunks);
            for (int k = 0; k < n_chunks; ++k) order[k] = k;
            std::shuffle(order.begin(), order.end(), rng);
            const int lost = (g % 7 == 0) ? order[n_chunks / 2] : -1;           // this chunk's flag never comes
            std::atomic<bool> all_landed{false};
            std::thread device([&] {
                for (int k : order) {
                    if ((k + g) % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(10));
                    const size_t lo = k * chunk, hi = std::min(bytes, lo + chunk);
                    for (size_t b = lo; b < hi; ++b) stage[b] = (unsigned char)(tag + b % 7);
                    if (k != lost) flags[k].store(seq, std::memory_order_release);
                }
            });
            bool joined = false;
            long my_ticks = 0;
            sbe_host::run_as_chunks_land((g + round) % 5 == 0 ? nullptr : &pool, n_jobs,
                [&](int j) { return (int)((size_t)j * job / chunk); },
                [&](int j) { return (int)((std::min(bytes, (size_t)(j + 1) * job) - 1) / chunk); },
                [&](int k) { return flags[k].load(std::memory_order_acquire) == seq || all_landed.load(std::memory_order_acquire); },
                [&] { if (++my_ticks > 2000 && !joined) { device.join(); joined = true; all_landed.store(true, std::memory_order_release); } },
                [&](int j) { const size_t o = (size_t)j * job; std::copy(stage.begin() + o, stage.begin() + std::min(bytes, o + job), dst.begin() + o); });
            if (!joined) device.join();
            for (size_t b = 0; b < bytes; ++b)
                if (dst[b] != (unsigned char)(tag + b % 7)) { std::fprintf(stderr, "pool_tsan: run_as_chunks_land copied byte %zu before it landed\n", b); return 1; }
            chunk_jobs += n_jobs; ticks += my_ticks;
        }
    }
    std::printf("pool_tsan: %d generations in %d pools, %ld items, %ld run_as_chunks_land jobs, %ld poll calls, %ld ticks: ok\n",
                generations, rounds, items, chunk_jobs, polls, ticks);
    return 0;
}
