// test-only: any file -> BGZF on stdout through bgzf_append_block (level from argv[2], -1 = FqDeflate)
#include "FqDeflate.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::string data;
    std::vector<char> buf(1 << 20);
    for (size_t n; (n = fread(buf.data(), 1, buf.size(), f)) > 0;) data.append(buf.data(), n);
    fclose(f);
    const int level = argc > 2 ? atoi(argv[2]) : -1;
    std::string out;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t at = 0; at < data.size(); at += kBgzfInput) bgzf_append_block(data.data() + at, std::min(kBgzfInput, data.size() - at), level, out);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out.append((const char *)kBgzfEofBlock, sizeof kBgzfEofBlock);
    fwrite(out.data(), 1, out.size(), stdout);
    fprintf(stderr, "%zu -> %zu bytes (%.1f%%), %.1f MB/s\n", data.size(), out.size(), 100.0 * out.size() / (data.size() ? data.size() : 1), data.size() / dt / 1e6);
    return 0;
}
