// Stress of the enqueue-worker pool (cmf.jl_amd/csrc/cmf_workers.h) on the CPU, meant to be built with -fsanitize=thread:
//   * many batches of jobs posted to R workers, every job passing several meeting points (cmf_pool_barrier) and writing
//     into shared arrays in the pattern of the event-ordered collectives (own slot before the meeting point, the others'
//     slots after it) -- a missing acquire / release in the queue or the barrier is a data race TSAN reports, and a lost
//     wake-up is a hang the test's timeout catches;
//   * workers going to sleep between batches (the producer must wake them);
//   * failures injected in random jobs: the batch's remaining jobs are skipped, workers at a meeting point give up with the
//     echo code, cmf_pool_collect reports the real failure (not an echo) with its message, and the pool works afterwards;
//   * a queue deeper than its capacity (the producer waits for room).
// Prints "ok <batches> <failures injected>" and exits 0, or a message and exits 1.
#include "cmf_workers.h"

#include <cstdio>
#include <cstdlib>
#include <random>

static thread_local std::string t_err;

int main(int argc, char **argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 4;
    const int batches = argc > 2 ? atoi(argv[2]) : 300;
    const int ECHO = -1000;
    CmfWorkerPool pool;
    std::vector<int> started(R, 0);
    pool.on_start = [&](size_t i) { started[i] = 1; };
    pool.last_error = []() { return t_err; };
    cmf_pool_start(pool, R);
    std::vector<long> slot(R, 0), seen(R, 0);
    std::mt19937 rng(12345);
    long expect = 0;
    int injected = 0;
    for (int b = 0; b < batches; ++b) {
        const int jobs = 1 + (int)(rng() % 5);
        const bool deep = b % 50 == 49;                       // now and then more jobs than the queue holds
        const int njobs = deep ? 3 * (int)CmfWorker::QN : jobs;
        const int fail_job = (rng() % 7 == 0) ? (int)(rng() % njobs) : -1;
        const int fail_rank = (int)(rng() % R);
        if (fail_job >= 0) ++injected;
        for (int j = 0; j < njobs; ++j) {
            const long v = ++expect;
            for (int i = 0; i < R; ++i)
                cmf_pool_post(pool, (size_t)i, [&, i, j, v, fail_job, fail_rank]() -> int {
                    slot[i] = v;                                          // "record my event"
                    if (j == fail_job && i == fail_rank) {
                        t_err = "injected failure in job " + std::to_string(j) + " of worker " + std::to_string(i);
                        return 7;
                    }
                    if (int rc = cmf_pool_barrier(pool, ECHO)) { t_err = "echo"; return rc; }
                    long s = 0;
                    for (int k = 0; k < R; ++k) s += slot[k];             // "wait for everybody's event": all slots hold v now
                    if (s != v * R) { t_err = "a worker ran ahead of the meeting point"; return 9; }
                    if (int rc = cmf_pool_barrier(pool, ECHO)) { t_err = "echo"; return rc; } // nobody overwrites its slot before all have read
                    seen[i] = v;
                    return 0;
                });
        }
        if (!cmf_pool_wait(pool, 60.0)) { fprintf(stderr, "batch %d: the pool did not drain\n", b); return 1; }
        std::string err;
        const int rc = cmf_pool_collect(pool, ECHO, &err);
        if (fail_job < 0) {
            if (rc != 0) { fprintf(stderr, "batch %d: unexpected failure %d: %s\n", b, rc, err.c_str()); return 1; }
            for (int i = 0; i < R; ++i)
                if (seen[i] != expect) { fprintf(stderr, "batch %d: worker %d finished at %ld, expected %ld\n", b, i, seen[i], expect); return 1; }
        } else {
            if (rc != 7 || err.find("injected failure") == std::string::npos) {
                fprintf(stderr, "batch %d: expected the injected failure, got %d: %s\n", b, rc, err.c_str());
                return 1;
            }
        }
        if (b % 40 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(2)); // let the workers fall asleep
    }
    cmf_pool_stop(pool, false);
    for (int i = 0; i < R; ++i)
        if (!started[i]) { fprintf(stderr, "worker %d never started\n", i); return 1; }
    printf("ok %d %d\n", batches, injected);
    return 0;
}
