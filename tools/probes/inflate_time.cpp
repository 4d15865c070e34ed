#include "GzInflater.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>
int main(int argc, char **argv)
{
    int fd = open(argv[1], O_RDONLY);
    struct stat st; fstat(fd, &st);
    const unsigned char *p = (const unsigned char *)mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    size_t piece = 32u << 20;
    std::vector<char> buf(piece, 1);
    for (int rep = 0; rep < 3; ++rep) {
        GzInflater z(p, st.st_size);
        auto t0 = std::chrono::steady_clock::now();
        size_t total = 0;
        for (;;) { size_t n = z.read(buf.data(), piece); total += n; if (n < piece) break; }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("GzInflater %.1f MB/s (%zu bytes, %.3f s) %s\n", total / dt / 1e6, total, dt, z.error() ? z.error() : "");
    }
    for (int rep = 0; rep < 2; ++rep) {
        gzFile g = gzopen(argv[1], "r"); gzbuffer(g, 4u << 20);
        auto t0 = std::chrono::steady_clock::now();
        size_t total = 0;
        for (;;) { int n = gzread(g, buf.data(), piece); if (n <= 0) break; total += n; }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("zlib gzread %.1f MB/s\n", total / dt / 1e6);
        gzclose(g);
    }
}
