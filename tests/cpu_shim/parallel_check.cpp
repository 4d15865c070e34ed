// test-only: decodes a gzip file with GzParallel (chunk size and stretch count from the command
// line, so that small files still go through many rounds), writes the bytes to stdout
#include "GzParallel.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
int main(int argc, char **argv)
{
    int fd = open(argv[1], O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const unsigned char *p = st.st_size ? (const unsigned char *)mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : (const unsigned char *)"";
    const size_t chunk = argc > 2 ? strtoull(argv[2], nullptr, 10) : (2u << 20);
    const int width = argc > 3 ? atoi(argv[3]) : 0;
    const size_t piece = argc > 4 ? strtoull(argv[4], nullptr, 10) : (32u << 20);
    GzParallel z(p, st.st_size, chunk, width);
    std::vector<char> buf(piece);
    for (;;) {
        size_t n = z.read(buf.data(), piece);
        fwrite(buf.data(), 1, n, stdout);
        if (n < piece) break;
    }
    fprintf(stderr, "rounds %llu used %llu dropped %llu\n", (unsigned long long)z.rounds, (unsigned long long)z.stretches_used,
            (unsigned long long)z.stretches_dropped);
    if (z.error()) {
        fprintf(stderr, "error: %s\n", z.error());
        return 2;
    }
    return 0;
}
