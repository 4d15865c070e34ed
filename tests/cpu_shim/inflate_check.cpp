// test-only: decodes a gzip file with GzInflater in odd-sized reads, writes the bytes to stdout
#include "GzInflater.h"
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
    struct stat st; fstat(fd, &st);
    const unsigned char *p = st.st_size ? (const unsigned char *)mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : (const unsigned char *)"";
    size_t piece = argc > 2 ? strtoull(argv[2], nullptr, 10) : (32u << 20);
    GzInflater z(p, st.st_size);
    std::vector<char> buf(piece);
    for (;;) {
        size_t n = z.read(buf.data(), piece);
        fwrite(buf.data(), 1, n, stdout);
        if (n < piece) break;
    }
    if (z.error()) { fprintf(stderr, "error: %s\n", z.error()); return 2; }
    return 0;
}
