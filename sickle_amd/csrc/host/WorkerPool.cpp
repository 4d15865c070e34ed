#include "WorkerPool.h"

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

WorkerPool &WorkerPool::instance()
{
    // never destroyed: the error paths leave through exit(1) like the reference's, with the other
    // pipeline threads still running, and a pool torn down by the static destructors under them
    // would be freed memory in use (ThreadSanitizer found exactly that)
    static WorkerPool &pool = *new WorkerPool([] {
        if (const char *e = getenv("SICKLE_HOST_THREADS")) {
            const int n = atoi(e);
            if (n >= 1) return n;
        }
        cpu_set_t set;
        int n = 0;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
        if (n < 1) n = (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        // a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) can be far below the CPUs it
        // may be scheduled on; threads beyond the quota only get throttled
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32];
            long period = 0;
            if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
                const long q = atol(quota) / period;
                if (q >= 1 && q < n) n = (int)q;
            }
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            long q = -1, period = 0;
            if (fscanf(g, "%ld", &q) == 1 && q > 0) {
                if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                    if (fscanf(h, "%ld", &period) == 1 && period > 0 && q / period >= 1 && q / period < n)
                        n = (int)(q / period);
                    fclose(h);
                }
            }
            fclose(g);
        }
        if (n > 32) n = 32; // the stages are memory-bound long before that
        return n;
    }());
    return pool;
}

WorkerPool::WorkerPool(int threads)
{
    for (int i = 1; i < threads; ++i) workers.emplace_back([this] { worker_loop(); });
}

WorkerPool::~WorkerPool()
{
    {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
    }
    cv.notify_all();
    for (std::thread &t : workers) t.join();
}

void WorkerPool::run(Job &job)
{
    for (;;) {
        const size_t p = job.next.fetch_add(1);
        if (p >= job.parts) return;
        const size_t b = job.n * p / job.parts, e = job.n * (p + 1) / job.parts;
        if (b < e) (*job.fn)(b, e, p);
        job.done.fetch_add(1);
    }
}

void WorkerPool::worker_loop()
{
    for (;;) {
        std::shared_ptr<Job> job;
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return stop || !jobs.empty(); });
            if (stop) return;
            job = jobs.front();
            if (job->next.load() >= job->parts) { // nothing left to claim: retire it
                jobs.pop_front();
                continue;
            }
        }
        run(*job);
    }
}

void WorkerPool::parallel_for(size_t n, size_t parts, const std::function<void(size_t, size_t, size_t)> &fn)
{
    if (n == 0) return;
    if (parts < 1) parts = 1;
    if (parts > n) parts = n;
    if (parts == 1 || workers.empty()) {
        for (size_t p = 0; p < parts; ++p) {
            const size_t b = n * p / parts, e = n * (p + 1) / parts;
            if (b < e) fn(b, e, p);
        }
        return;
    }
    auto job = std::make_shared<Job>();
    job->fn = &fn;
    job->n = n;
    job->parts = parts;
    {
        std::lock_guard<std::mutex> lk(m);
        jobs.push_back(job);
    }
    cv.notify_all();
    run(*job); // the caller works too
    while (job->done.load() < parts) std::this_thread::yield();
    {
        std::lock_guard<std::mutex> lk(m);
        for (auto it = jobs.begin(); it != jobs.end(); ++it)
            if (it->get() == job.get()) {
                jobs.erase(it);
                break;
            }
    }
}
