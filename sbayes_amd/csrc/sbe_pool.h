// sbe_pool.h -- host worker threads of sbe_step_batch: run job(0..n-1) on the workers and the calling thread.
//
// Plain C++17 (no HIP): included by sbe_engine.hip and, on its own, by tests/c/pool_tsan.cpp, which runs it under
// ThreadSanitizer on the CPU (log: profiles/r3/tsan_pool.log).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>

namespace sbe_host {

// One turn of a spin-wait.  Several engines share a host (one process per chain: sbayes/mcmc_setup.py:271-282), and their
// waiting threads together can outnumber the cores the processes may use; a waiter that only pauses then burns the time slice
// of a thread that has work (measured: six single-chain processes on one card moved half the streamed [N, F, C] results one
// process moves alone, tests/test_gpu_processes.py).  So every 64th turn gives the core away -- sched_yield returns at once
// when nobody else is runnable, i.e. it costs nothing on an idle host.
struct SpinWait {
    unsigned n = 0;
    void turn() { if ((++n & 63u) == 0u) sched_yield(); else __builtin_ia32_pause(); }
};

struct StepPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_idle;
    std::function<void(int)> job;      // job(i) for i in [0, n_items)
    int n_items = 0;
    std::atomic<int> next{0}, done{0}; // items are claimed and counted without the lock (16 threads on one mutex
                                       // cost more than the 5 us jobs they were handing out)
    int active = 0;                    // workers inside the claim loop of the current generation (under m)
    uint64_t generation = 0;
    bool stop = false;
    // A sleeping worker needs tens of microseconds to wake -- as long as the whole host half of a 64-chain sweep.  So a
    // worker that has just finished keeps polling this counter for a short while (kSpinUs) before it blocks: while
    // sweeps follow each other (an MCMC loop) the pool stays hot, an idle engine costs nothing.
    std::atomic<uint64_t> gen_hint{0};
    std::atomic<bool> stop_hint{false};
    static constexpr int kSpinUs = 400;

    explicit StepPool(int n_threads) {
        for (int t = 0; t < n_threads; ++t) workers.emplace_back([this] { loop(); });
    }
    ~StepPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; stop_hint.store(true); }
        cv_work.notify_all();
        for (auto& w : workers) w.join();
    }
    void claim_loop(const std::function<void()>* poll = nullptr) {
        for (;;) {
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_items) break;
            job(i);
            done.fetch_add(1, std::memory_order_release);
            if (poll) (*poll)();
        }
    }
    void loop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            if (seen != 0) {                                         // (after the first job: poll before blocking)
                lk.unlock();
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(kSpinUs);
                SpinWait sw;
                while (gen_hint.load(std::memory_order_acquire) == seen && !stop_hint.load(std::memory_order_relaxed) &&
                       std::chrono::steady_clock::now() < t_end)
                    sw.turn();
                lk.lock();
            }
            cv_work.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            ++active;                                                // run() does not touch job / n_items while active > 0
            lk.unlock();
            claim_loop();
            lk.lock();
            if (--active == 0) cv_idle.notify_all();
        }
    }
    // run job(0..n-1) on the workers and the calling thread; returns when all are done.  `poll` (optional) is called by
    // the CALLING thread after each of its own items and while it waits for the others (e.g. to issue HIP copies for
    // the items that are finished: HIP calls stay on one thread)
    void run(int n, std::function<void(int)> f, const std::function<void()>* poll = nullptr) {
        {
            std::unique_lock<std::mutex> lk(m);
            cv_idle.wait(lk, [&] { return active == 0; });           // (a worker that woke late for the previous run)
            job = std::move(f); n_items = n;
            done.store(0, std::memory_order_relaxed); next.store(0, std::memory_order_relaxed);
            ++generation;
            gen_hint.store(generation, std::memory_order_release);
        }
        cv_work.notify_all();
        claim_loop(poll);
        SpinWait sw;
        while (done.load(std::memory_order_acquire) < n) {                          // (jobs are microseconds long)
            if (poll) (*poll)();
            sw.turn();
        }
    }
};

// Jobs that consume a large device-to-host result while it is still arriving (the literal a1 / a3 call surfaces hand back
// [N, F] / [N, F, C] float64 arrays).  The result lands CHUNK BY CHUNK with one completion flag per chunk, in any order (a kernel storing into
// host-mapped memory, sbe_kernels.hip.h signal_chunk): job j reads chunks first_chunk(j)..last_chunk(j) and every thread
// looks at the flags ITSELF -- `chunk_ready(k)` only reads host memory -- so no job waits for another thread to notice.
// `caller_tick` runs on the calling thread only (while it waits inside its own job, after each of its jobs, while it
// waits for the others): the one place that may talk to the HIP runtime -- the engine uses it to fall back to a stream
// synchronisation when a flag stays away, after which chunk_ready must return true for every chunk.
template <class First, class Last, class Ready, class Tick, class Work>
void run_as_chunks_land(StepPool* pool, int n, First first_chunk, Last last_chunk, Ready chunk_ready, Tick caller_tick, Work work) {
    auto wait_for = [&](int j, bool mine) {
        SpinWait sw;
        for (int k = first_chunk(j), k1 = last_chunk(j); k <= k1; ++k)
            while (!chunk_ready(k)) { if (mine) caller_tick(); else sw.turn(); }
        std::atomic_thread_fence(std::memory_order_acquire);
    };
    if (!pool) {
        for (int j = 0; j < n; ++j) { wait_for(j, true); work(j); }
        return;
    }
    const std::thread::id caller = std::this_thread::get_id();
    const std::function<void()> tick = [&] { caller_tick(); };
    pool->run(n, [&](int j) { wait_for(j, std::this_thread::get_id() == caller); work(j); }, &tick);
}

}  // namespace sbe_host
