// probe: GzParallel / GzInflater / zlib gzread decode rate of one gzip file (bytes discarded)
#include "GzInflater.h"
#include "GzParallel.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>
template <class Z> static void timeit(const char *name, Z &z, std::vector<char> &buf)
{
    auto t0 = std::chrono::steady_clock::now();
    size_t total = 0;
    for (;;) { size_t n = z.read(buf.data(), buf.size()); total += n; if (n < buf.size()) break; }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s %.1f MB/s (%zu bytes, %.3f s) %s\n", name, total / dt / 1e6, total, dt, z.error() ? z.error() : "");
}
int main(int argc, char **argv)
{
    int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    const unsigned char *p = (const unsigned char *)mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    std::vector<char> buf(32u << 20, 1);
    for (int rep = 0; rep < 3; ++rep) { GzParallel z(p, st.st_size); timeit("GzParallel", z, buf); printf("  rounds %llu used %llu dropped %llu\n", (unsigned long long)z.rounds, (unsigned long long)z.stretches_used, (unsigned long long)z.stretches_dropped); }
    for (int rep = 0; rep < 2; ++rep) { GzInflater z(p, st.st_size); timeit("GzInflater", z, buf); }
    { gzFile g = gzopen(argv[1], "r"); gzbuffer(g, 4u << 20); auto t0 = std::chrono::steady_clock::now(); size_t total = 0;
      for (;;) { int n = gzread(g, buf.data(), (unsigned)buf.size()); if (n <= 0) break; total += n; }
      double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); printf("zlib gzread %.1f MB/s\n", total / dt / 1e6); gzclose(g); }
}
