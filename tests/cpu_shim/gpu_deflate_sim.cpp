// test-only: runs the phases of sk_deflate_block.h (the GPU's BGZF block encoder) on the host, the
// 64 lanes one after the other with the barriers where the kernel has them; any file -> BGZF on stdout
#include "sk_deflate_block.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static size_t encode_block(const uint8_t *p, uint32_t n, uint32_t *out_words, skd_shared *sh, uint32_t *tok)
{
#define ALL_LANES(call) for (int lane = 0; lane < SKD_LANES; ++lane) { call; }
    ALL_LANES(skd_phase_clear(sh, out_words, lane));
    ALL_LANES(skd_phase_count_newlines(sh, p, n, lane));
    skd_phase_scan_segments(sh, n);
    ALL_LANES(skd_phase_line_starts(sh, p, n, lane));
    skd_phase_close_lines(sh, p, n);
    ALL_LANES(skd_phase_tokenize(sh, p, tok, lane));
    skd_phase_codes_and_header(sh, out_words);
    ALL_LANES(skd_phase_size_lines(sh, tok, lane));
    skd_phase_place_lines(sh, out_words);
    ALL_LANES(skd_phase_emit(sh, tok, out_words, lane));
    if (sh->total_bits > (SKD_OUT_WORDS - 2) * 32u) return 0;
    return (sh->total_bits + 7) / 8;
}

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::string data;
    std::vector<char> buf(1 << 20);
    for (size_t n; (n = fread(buf.data(), 1, buf.size(), f)) > 0;) data.append(buf.data(), n);
    fclose(f);
    std::vector<uint32_t> out_words(SKD_OUT_WORDS), tok(SKD_BLOCK_MAX + 8);
    skd_shared *sh = new skd_shared;
    std::string out;
    size_t stored = 0;
    for (size_t at = 0; at < data.size() || (at == 0 && data.empty()); at += SKD_BLOCK_MAX) {
        const uint32_t n = (uint32_t)std::min<size_t>(SKD_BLOCK_MAX, data.size() - at);
        const uint8_t *p = (const uint8_t *)data.data() + at;
        size_t clen = n ? encode_block(p, n, out_words.data(), sh, tok.data()) : 0;
        std::string body;
        if (clen == 0 || clen >= n + 5) { // empty, or not compressible: a stored block
            ++stored;
            body.push_back(1);
            body.push_back((char)(n & 0xff));
            body.push_back((char)(n >> 8));
            body.push_back((char)(~n & 0xff));
            body.push_back((char)((~n >> 8) & 0xff));
            body.append((const char *)p, n);
        } else {
            body.assign((const char *)out_words.data(), clen);
        }
        const uint32_t total = (uint32_t)(18 + body.size() + 8);
        const unsigned char head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)((total - 1) & 0xff), (unsigned char)((total - 1) >> 8)};
        out.append((const char *)head, 18);
        out.append(body);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, n);
        for (int i = 0; i < 4; ++i) out.push_back((char)(crc >> (8 * i)));
        for (int i = 0; i < 4; ++i) out.push_back((char)(n >> (8 * i)));
        if (data.empty()) break;
    }
    fwrite(out.data(), 1, out.size(), stdout);
    fprintf(stderr, "%zu -> %zu bytes (%.1f%%), %zu stored blocks\n", data.size(), out.size(), 100.0 * out.size() / (data.size() ? data.size() : 1), stored);
    return 0;
}
