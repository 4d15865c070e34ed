// probe (host only): how fast can one process put N bytes into a NEW file in tmpfs?
//   a) one pwrite stream            b) T pwrite streams into disjoint ranges of the one file
//   c) ftruncate + mmap, T threads memcpy into their ranges (page faults in the threads)
//   d) as c with MAP_POPULATE (pages made by the mapping call, one thread), then T threads memcpy
//   e) three files, one pwrite stream each (what the CLI's writer stage does now)
// usage: write_paths <dir> <GiB> <threads>
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <fcntl.h>
#include <linux/falloc.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t bytes = (size_t)(atof(argc > 2 ? argv[2] : "4") * (1ull << 30));
    const int T = argc > 3 ? atoi(argv[3]) : 16;
    std::vector<char> src(64 << 20);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (char)(i * 131 >> 3);
    auto path = [&](const char *tag) { return dir + "/wp_" + tag; };
    auto fresh = [&](const char *tag) { std::string p = path(tag); unlink(p.c_str()); return open(p.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644); };
    auto pw_range = [&](int fd, size_t lo, size_t hi) {
        for (size_t at = lo; at < hi;) {
            const size_t n = std::min(src.size(), hi - at);
            ssize_t w = pwrite(fd, src.data(), n, (off_t)at);
            if (w <= 0) { perror("pwrite"); exit(1); }
            at += (size_t)w;
        }
    };
    auto report = [&](const char *what, double t) { printf("%-62s %.3f s  %.2f GB/s\n", what, t, bytes / t / 1e9); fflush(stdout); };
    {
        int fd = fresh("a");
        double t0 = now();
        pw_range(fd, 0, bytes);
        report("a) one pwrite stream", now() - t0);
        close(fd); unlink(path("a").c_str());
    }
    for (int th : {3, T}) {
        int fd = fresh("b");
        double t0 = now();
        std::vector<std::thread> ts;
        for (int i = 0; i < th; ++i) ts.emplace_back([&, i] { pw_range(fd, bytes / th * i, i == th - 1 ? bytes : bytes / th * (i + 1)); });
        for (auto &t : ts) t.join();
        char nm[96]; snprintf(nm, sizeof nm, "b) %d pwrite streams, disjoint ranges of one file", th);
        report(nm, now() - t0);
        close(fd); unlink(path("b").c_str());
    }
    for (int populate = 0; populate < 2; ++populate) {
        int fd = fresh("c");
        double t0 = now();
        if (ftruncate(fd, (off_t)bytes)) { perror("ftruncate"); return 1; }
        char *m = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        double t1 = now();
        std::vector<std::thread> ts;
        for (int i = 0; i < T; ++i)
            ts.emplace_back([&, i] {
                const size_t lo = bytes / T * i, hi = i == T - 1 ? bytes : bytes / T * (i + 1);
                for (size_t at = lo; at < hi; at += src.size()) memcpy(m + at, src.data(), std::min(src.size(), hi - at));
            });
        for (auto &t : ts) t.join();
        double t2 = now();
        munmap(m, bytes);
        char nm[128]; snprintf(nm, sizeof nm, "%s) mmap%s + %d threads memcpy (map %.3f s, copy %.3f s)", populate ? "d" : "c", populate ? " MAP_POPULATE" : "", T, t1 - t0, t2 - t1);
        report(nm, now() - t0);
        close(fd); unlink(path("c").c_str());
    }
    {
        int fds[3] = {fresh("e0"), fresh("e1"), fresh("e2")};
        double t0 = now();
        std::vector<std::thread> ts;
        for (int i = 0; i < 3; ++i) ts.emplace_back([&, i] { pw_range(fds[i], 0, bytes / 3); });
        for (auto &t : ts) t.join();
        report("e) three files, one pwrite stream each (a third of the bytes each)", now() - t0);
        for (int i = 0; i < 3; ++i) close(fds[i]);
        unlink(path("e0").c_str()); unlink(path("e1").c_str()); unlink(path("e2").c_str());
    }
    {   // f) three files, each: ftruncate + mmap + T/3 threads
        int fds[3] = {fresh("f0"), fresh("f1"), fresh("f2")};
        const size_t per = bytes / 3;
        double t0 = now();
        char *ms[3];
        for (int i = 0; i < 3; ++i) { if (ftruncate(fds[i], (off_t)per)) return 1; ms[i] = (char *)mmap(nullptr, per, PROT_READ | PROT_WRITE, MAP_SHARED, fds[i], 0); }
        std::vector<std::thread> ts;
        for (int i = 0; i < T; ++i)
            ts.emplace_back([&, i] {
                for (int f = 0; f < 3; ++f) {
                    const size_t lo = per / T * i, hi = i == T - 1 ? per : per / T * (i + 1);
                    for (size_t at = lo; at < hi; at += src.size()) memcpy(ms[f] + at, src.data(), std::min(src.size(), hi - at));
                }
            });
        for (auto &t : ts) t.join();
        for (int i = 0; i < 3; ++i) munmap(ms[i], per);
        char nm[128]; snprintf(nm, sizeof nm, "f) three files, mmap, %d threads each filling its slice of all three", T);
        report(nm, now() - t0);
        for (int i = 0; i < 3; ++i) close(fds[i]);
        unlink(path("f0").c_str()); unlink(path("f1").c_str()); unlink(path("f2").c_str());
    }
    {   // g) posix_fallocate of the whole file (pages made and zeroed, no data), then one / two pwrite streams into it
        for (int streams : {1, 2}) {
            int fd = fresh("g");
            double t0 = now();
            if (fallocate(fd, 0, 0, (off_t)bytes)) { perror("fallocate"); return 1; }
            double t1 = now();
            std::vector<std::thread> ts;
            for (int i = 0; i < streams; ++i) ts.emplace_back([&, i] { pw_range(fd, bytes / streams * i, i == streams - 1 ? bytes : bytes / streams * (i + 1)); });
            for (auto &t : ts) t.join();
            double t2 = now();
            char nm[160]; snprintf(nm, sizeof nm, "g) fallocate (%.3f s = %.2f GB/s) then %d pwrite stream(s) into the made pages (%.3f s = %.2f GB/s)", t1 - t0, bytes / (t1 - t0) / 1e9, streams, t2 - t1, bytes / (t2 - t1) / 1e9);
            report(nm, now() - t0);
            close(fd); unlink(path("g").c_str());
        }
    }
    {   // h) two files: a helper thread fallocates 256 MiB ahead of each file's single pwrite stream
        int fds[2] = {fresh("h0"), fresh("h1")};
        const size_t per = bytes / 2, step = 256u << 20;
        double t0 = now();
        std::vector<std::thread> ts;
        for (int f = 0; f < 2; ++f) {
            ts.emplace_back([&, f] { for (size_t at = 0; at < per; at += step) if (fallocate(fds[f], FALLOC_FL_KEEP_SIZE, (off_t)at, (off_t)std::min(step, per - at))) { perror("fallocate"); exit(1); } });
            ts.emplace_back([&, f] { pw_range(fds[f], 0, per); });
        }
        for (auto &t : ts) t.join();
        report("h) two files, each: one thread fallocating ahead + one pwrite stream (half of the bytes each)", now() - t0);
        for (int f = 0; f < 2; ++f) close(fds[f]);
        unlink(path("h0").c_str()); unlink(path("h1").c_str());
    }
    {   // i) two files, one pwrite stream each, nothing else (the reference point for h)
        int fds[2] = {fresh("i0"), fresh("i1")};
        const size_t per = bytes / 2;
        double t0 = now();
        std::vector<std::thread> ts;
        for (int f = 0; f < 2; ++f) ts.emplace_back([&, f] { pw_range(fds[f], 0, per); });
        for (auto &t : ts) t.join();
        report("i) two files, one pwrite stream each (half of the bytes each)", now() - t0);
        for (int f = 0; f < 2; ++f) close(fds[f]);
        unlink(path("i0").c_str()); unlink(path("i1").c_str());
    }
    return 0;
}
