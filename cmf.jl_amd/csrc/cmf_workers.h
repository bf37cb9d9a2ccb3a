// cmf_workers.h -- the enqueue workers of a T-sharded group (cmf_group.h), free of HIP so that the queue, the meeting point and
// the abort protocol can be exercised under ThreadSanitizer on a CPU (tests/worker_pool_stress.cpp, tests/test_worker_pool.py).
//
// One thread per local shard.  Single producer (the thread that calls the ABI -- a handle is used by one host thread at a
// time), single consumer per queue.  A worker spins briefly for the next job and then sleeps on a condition variable, so an
// idle group costs nothing.  A job that fails (returns non-zero) raises the pool's abort flag: the rest of every worker's batch
// is skipped (its kernels would run on half-made inputs), and workers waiting at the meeting point give up with `echo`.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#if defined(__x86_64__) || defined(__i386__)
#define CMF_CPU_PAUSE() __builtin_ia32_pause()
#else
#define CMF_CPU_PAUSE() std::this_thread::yield()
#endif

struct CmfWorker {
    static constexpr uint32_t QN = 64;
    std::thread th;
    std::function<int()> q[QN];
    std::atomic<uint32_t> head{0}, tail{0}; // consumer / producer positions (free running)
    std::atomic<bool> quit{false}, asleep{false};
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> rc{0};                 // code of the first failed job since the last collect
    std::string err;                        // its message (written before rc is published)
    std::atomic<int64_t> busy_ns{0};        // time spent inside jobs
};

// What a worker thread touches besides its own CmfWorker: co-owned by the pool and by every worker thread (shared_ptr), so
// that a worker that had to be ABANDONED inside a call that never returned (cmf_pool_stop(..., true)) and wakes up later --
// e.g. when its communicator is aborted -- never sees freed memory, whatever has happened to the pool's owner meanwhile.
struct CmfPoolShared {
    std::atomic<bool> abort{false};         // a job failed (or a wait ran out): meeting points give up, batches are skipped
    std::atomic<int> bar_count{0}, bar_gen{0};
    std::function<void(size_t)> on_start;   // runs first on worker i's thread (bind the device)
    std::function<std::string()> last_error; // the calling thread's error text after a failed job (thread-local in the library)
};

struct CmfWorkerPool {
    std::shared_ptr<CmfPoolShared> sh = std::make_shared<CmfPoolShared>();
    std::vector<std::unique_ptr<CmfWorker>> w;
    std::atomic<bool> &abort = sh->abort;
    std::atomic<int> &bar_count = sh->bar_count, &bar_gen = sh->bar_gen;
    std::function<void(size_t)> &on_start = sh->on_start;
    std::function<std::string()> &last_error = sh->last_error;
    CmfWorkerPool() = default;
    CmfWorkerPool(const CmfWorkerPool &) = delete;
    CmfWorkerPool &operator=(const CmfWorkerPool &) = delete;
    bool empty() const { return w.empty(); }
    size_t size() const { return w.size(); }
};

static void cmf_worker_main(std::shared_ptr<CmfPoolShared> pool, CmfWorker *w, size_t i)
{
    if (pool->on_start) pool->on_start(i);
    unsigned idle = 0;
    for (;;) {
        const uint32_t h = w->head.load(std::memory_order_relaxed);
        if (h == w->tail.load(std::memory_order_acquire)) {
            if (w->quit.load(std::memory_order_acquire)) return;
            if (++idle < 20000) { CMF_CPU_PAUSE(); continue; }
            std::unique_lock<std::mutex> lock(w->mu); // nothing for a while (~0.2 ms): sleep until the producer posts
            w->asleep.store(true, std::memory_order_seq_cst);
            // (seq_cst on both sides of the handshake with cmf_pool_post -- store asleep, load tail here; store tail, load asleep
            // there -- so that at least one side sees the other's store on every architecture, not only on x86)
            w->cv.wait(lock, [&] { return h != w->tail.load(std::memory_order_seq_cst) || w->quit.load(std::memory_order_acquire); });
            w->asleep.store(false, std::memory_order_seq_cst);
            idle = 0;
            continue;
        }
        idle = 0;
        std::function<int()> &job = w->q[h % CmfWorker::QN];
        if (w->rc.load(std::memory_order_relaxed) == 0 && !pool->abort.load(std::memory_order_acquire)) {
            const auto tj = std::chrono::steady_clock::now();
            const int rc = job();
            w->busy_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tj).count(), std::memory_order_relaxed);
            if (rc != 0) {
                if (pool->last_error) w->err = pool->last_error();
                w->rc.store(rc, std::memory_order_release);
                pool->abort.store(true, std::memory_order_release);
            }
        }
        job = nullptr;
        w->head.store(h + 1, std::memory_order_release);
    }
}

static void cmf_pool_start(CmfWorkerPool &pool, size_t n)
{
    for (size_t i = 0; i < n; ++i) pool.w.emplace_back(new CmfWorker());
    for (size_t i = 0; i < n; ++i) pool.w[i]->th = std::thread(cmf_worker_main, pool.sh, pool.w[i].get(), i);
}

static void cmf_pool_post(CmfWorkerPool &pool, size_t i, std::function<int()> job)
{
    CmfWorker *w = pool.w[i].get();
    const uint32_t t = w->tail.load(std::memory_order_relaxed);
    while (t - w->head.load(std::memory_order_acquire) >= CmfWorker::QN) CMF_CPU_PAUSE(); // queue full: the worker is behind
    w->q[t % CmfWorker::QN] = std::move(job);
    w->tail.store(t + 1, std::memory_order_seq_cst);
    if (w->asleep.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lock(w->mu);
        w->cv.notify_one();
    }
}

// have the workers taken everything that was posted?
static bool cmf_pool_idle(const CmfWorkerPool &pool)
{
    for (const auto &w : pool.w)
        if (w->head.load(std::memory_order_acquire) != w->tail.load(std::memory_order_acquire)) return false;
    return true;
}

// Wait until every posted job has run; false when `timeout_s` ran out first (the abort flag is then raised).
static bool cmf_pool_wait(CmfWorkerPool &pool, double timeout_s)
{
    const auto t_begin = std::chrono::steady_clock::now();
    for (unsigned spins = 1; !cmf_pool_idle(pool); ++spins) {
        if ((spins & 0xFFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > timeout_s) {
            pool.abort.store(true, std::memory_order_release);
            return false;
        }
        CMF_CPU_PAUSE();
    }
    return true;
}

// After cmf_pool_wait: the first real failure in worker order (0 = none; a worker that only gave up at a meeting point because
// ANOTHER one failed reports `echo`, which counts only if nothing else is there), its message in *err; clears the pool's failure state.
static int cmf_pool_collect(CmfWorkerPool &pool, int echo, std::string *err)
{
    int rc = 0;
    for (int pass = 0; pass < 2 && rc == 0; ++pass)
        for (auto &w : pool.w) {
            const int r = w->rc.load(std::memory_order_acquire);
            if (r != 0 && rc == 0 && (pass == 1 || r != echo)) {
                rc = r;
                if (err) *err = w->err;
            }
        }
    for (auto &w : pool.w) w->rc.store(0, std::memory_order_relaxed);
    pool.abort.store(false, std::memory_order_release);
    pool.bar_count.store(0, std::memory_order_relaxed); // (workers that left a meeting point on abort did not complete it)
    return rc;
}

// Meeting point of ALL workers inside a job (an event must have been recorded before another stream is told to wait for it).
// Returns 0, or `echo` when any job of the pool has failed meanwhile.
static int cmf_pool_barrier(CmfWorkerPool &pool, int echo)
{
    const int n = (int)pool.w.size();
    const int gen = pool.bar_gen.load(std::memory_order_acquire);
    if (pool.bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
        pool.bar_count.store(0, std::memory_order_relaxed);
        pool.bar_gen.fetch_add(1, std::memory_order_acq_rel);
        return 0;
    }
    while (pool.bar_gen.load(std::memory_order_acquire) == gen) {
        if (pool.abort.load(std::memory_order_acquire)) return echo;
        CMF_CPU_PAUSE();
    }
    return 0;
}

// Ends the threads.  abandon_busy: a worker that still holds a job (stuck inside a call that will never return) is detached and
// its queue leaked instead of joined.
// Returns how many workers had to be abandoned inside a job (abandon_busy): whatever their jobs captured must then stay alive.
static size_t cmf_pool_stop(CmfWorkerPool &pool, bool abandon_busy)
{
    size_t abandoned = 0;
    for (auto &w : pool.w) {
        w->quit.store(true, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lock(w->mu);
            w->cv.notify_one();
        }
        if (!w->th.joinable()) continue;
        if (abandon_busy && w->head.load(std::memory_order_acquire) != w->tail.load(std::memory_order_acquire)) {
            w->th.detach();
            (void)w.release();
            ++abandoned;
            continue;
        }
        w->th.join();
    }
    pool.w.clear();
    return abandoned;
}
