// Order-independent fingerprint of a FASTQ file: number of 4-line records, sum and xor of a
// 64-bit FNV-1a hash per record.  Used by tools/e2e_bench.py when two outputs differ in md5:
// equal fingerprints = same records in a different order.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<char> buf(64u << 20);
    uint64_t h = 1469598103934665603ull, sum = 0, x = 0, records = 0;
    int lines = 0;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const unsigned char c = (unsigned char)buf[i];
            h = (h ^ c) * 1099511628211ull;
            if (c == '\n' && ++lines == 4) {
                sum += h;
                x ^= h;
                ++records;
                lines = 0;
                h = 1469598103934665603ull;
            }
        }
    }
    printf("%llu %016llx %016llx\n", (unsigned long long)records, (unsigned long long)sum, (unsigned long long)x);
    return 0;
}
