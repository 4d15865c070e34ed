// WorkerPool.h -- a small shared thread pool for the host side of the pipeline (newline
// indexing, record framing, packing into pinned buffers, output assembly).  The reference
// spends its -a threads on the scan itself (src/trim_single.cpp:323-333); here the scan is on
// the GPU and the host threads go to the parts that are left.  Several pipeline stages call
// parallel_for concurrently; the caller always takes part in its own loop, so a call never
// waits on a pool that is busy with another stage's work.
#ifndef SICKLE_WORKERPOOL_H
#define SICKLE_WORKERPOOL_H

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

class WorkerPool {
public:
    // the process-wide pool: SICKLE_HOST_THREADS or the CPUs this process may run on
    static WorkerPool &instance();
    explicit WorkerPool(int threads);
    ~WorkerPool();
    int size() const { return (int)workers.size() + 1; }

    // fn(begin, end, part) over [0, n) cut into `parts` contiguous ranges (part = range index)
    void parallel_for(size_t n, size_t parts, const std::function<void(size_t, size_t, size_t)> &fn);

private:
    struct Job {
        const std::function<void(size_t, size_t, size_t)> *fn;
        size_t n, parts;
        std::atomic<size_t> next{0}, done{0};
    };
    void run(Job &job);
    void worker_loop();
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Job>> jobs;
    bool stop = false;
};

#endif
