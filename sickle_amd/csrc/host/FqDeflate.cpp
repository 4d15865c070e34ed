#include "FqDeflate.h"

#include <algorithm>
#include <cstring>
#include <queue>
#include <vector>

#include <zlib.h>

#include "Deflate.h"

using namespace deflate_detail;

namespace {
constexpr int kMinAligned = 6; // shortest copy from the line four lines up worth a length/distance pair
constexpr int kMinRun = 5;     // shortest run worth one

struct SymbolMaps {
    uint8_t len_sym[256];  // length - 3 -> code - 257
    uint8_t dist_sym[512]; // zlib's two-level map: d < 256 ? [d] : [256 + (d >> 7)], d = distance - 1
    SymbolMaps()
    {
        for (int s = 0; s < 29; ++s) {
            const int hi = s == 28 ? 258 : kLenBase[s] + (1 << kLenExtra[s]) - 1;
            for (int l = kLenBase[s]; l <= hi && l <= 258; ++l) len_sym[l - 3] = (uint8_t)s;
        }
        len_sym[255] = 28;
        for (int s = 0; s < 30; ++s)
            for (int d = kDistBase[s] - 1; d < kDistBase[s] - 1 + (1 << kDistExtra[s]); ++d) {
                if (d < 256) dist_sym[d] = (uint8_t)s;
                else dist_sym[256 + (d >> 7)] = (uint8_t)s;
            }
    }
    int dist(uint32_t d) const { return d < 256 ? dist_sym[d] : dist_sym[256 + (d >> 7)]; }
};
const SymbolMaps maps;

struct BitWriter {
    unsigned char *p, *end;
    uint64_t acc = 0;
    int n = 0;
    bool overflow = false;
    void put(uint32_t bits, int count)
    {
        acc |= (uint64_t)bits << n;
        n += count;
        if (n >= 32) {
            if (end - p < 4) {
                overflow = true;
                n -= 32;
                acc >>= 32;
                return;
            }
            const uint32_t w = (uint32_t)acc;
            memcpy(p, &w, 4);
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    void flush()
    {
        while (n > 0) {
            if (p >= end) {
                overflow = true;
                return;
            }
            *p++ = (unsigned char)acc;
            acc >>= 8;
            n -= 8;
        }
        n = 0;
    }
};

// Length-limited canonical Huffman code: lens[] and bit-reversed codes[] for n symbols.
void huffman(const uint32_t *freq, int n, int max_len, uint8_t *lens, uint16_t *codes)
{
    memset(lens, 0, (size_t)n);
    std::vector<int> used;
    for (int s = 0; s < n; ++s)
        if (freq[s]) used.push_back(s);
    if (used.empty()) return;
    if (used.size() == 1) { // one symbol: give it (and a neighbour) one bit, so the code is complete
        lens[used[0]] = 1;
        lens[used[0] == 0 ? 1 : 0] = 1;
    } else {
        // depth of every leaf of the Huffman tree
        struct Node {
            uint64_t f;
            int id;
            bool operator>(const Node &o) const { return f > o.f || (f == o.f && id > o.id); }
        };
        std::priority_queue<Node, std::vector<Node>, std::greater<Node>> heap;
        std::vector<int> parent(used.size() * 2, -1);
        for (size_t i = 0; i < used.size(); ++i) heap.push({freq[used[i]], (int)i});
        int next = (int)used.size();
        while (heap.size() > 1) {
            const Node a = heap.top();
            heap.pop();
            const Node b = heap.top();
            heap.pop();
            parent[(size_t)a.id] = parent[(size_t)b.id] = next;
            heap.push({a.f + b.f, next++});
        }
        std::vector<int> depth((size_t)next, 0);
        for (int i = next - 2; i >= 0; --i) depth[(size_t)i] = depth[(size_t)parent[(size_t)i]] + 1;
        // how many codes of each length; fold the too long ones back under max_len (Kraft sum kept at 1)
        int count[64] = {0};
        for (size_t i = 0; i < used.size(); ++i) count[std::min(depth[i], 63)]++;
        for (int l = max_len + 1; l < 64; ++l) {
            count[max_len] += count[l];
            count[l] = 0;
        }
        uint64_t total = 0;
        for (int l = max_len; l > 0; --l) total += (uint64_t)count[l] << (max_len - l);
        while (total != (1ull << max_len)) {
            count[max_len]--;
            for (int l = max_len - 1; l > 0; --l)
                if (count[l]) {
                    count[l]--;
                    count[l + 1] += 2;
                    break;
                }
            total--;
        }
        // shortest codes to the most frequent symbols
        std::vector<int> order(used.size());
        for (size_t i = 0; i < used.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            return freq[used[(size_t)a]] > freq[used[(size_t)b]] || (freq[used[(size_t)a]] == freq[used[(size_t)b]] && a < b);
        });
        size_t at = 0;
        for (int l = 1; l <= max_len; ++l)
            for (int k = 0; k < count[l]; ++k) lens[used[(size_t)order[at++]]] = (uint8_t)l;
    }
    uint32_t next_code[17] = {0};
    int bl_count[17] = {0};
    for (int s = 0; s < n; ++s) bl_count[lens[s]]++;
    bl_count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code + (uint32_t)bl_count[l - 1]) << 1;
        next_code[l] = code;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = next_code[l]++;
        uint32_t r = 0;
        for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
        codes[s] = (uint16_t)r;
    }
}

struct Scratch {
    std::vector<uint32_t> tokens; // literal: the byte; match: 1<<31 | (length-3)<<15 | (distance-1)
};

inline uint32_t match_token(uint32_t len, uint32_t dist) { return (1u << 31) | ((len - 3) << 15) | (dist - 1); }

// The text as literals, runs, and copies from the same column of the line four lines up.
void tokenize(const unsigned char *p, size_t n, std::vector<uint32_t> &out)
{
    out.clear();
    size_t line_start[4] = {SIZE_MAX, SIZE_MAX, SIZE_MAX, SIZE_MAX}; // of the last four lines, oldest first
    size_t i = 0;
    while (i < n) {
        const unsigned char *nl = (const unsigned char *)memchr(p + i, '\n', n - i);
        const size_t end = nl ? (size_t)(nl - p) + 1 : n; // the line with its newline
        const size_t ref_line = line_start[0];
        const size_t ref_end = line_start[1]; // the reference line ends where the next one starts
        long shift = 0;                       // reference column = own column + shift
        size_t j = i;
        // only lines that begin like the one four up are compared with it (headers, '+' lines);
        // bases and qualities begin differently and are not worth five probes per byte
        bool use_ref = false;
        if (ref_line != SIZE_MAX && j - ref_line <= 32768) {
            const size_t head = std::min({(size_t)kMinAligned, end - i, ref_end - ref_line});
            use_ref = head > 0 && memcmp(p + i, p + ref_line, head) == 0;
        }
        while (j < end) {
            // copy from the reference line (same column, or one or two to the side once a field
            // has changed width)
            size_t best = 0, best_ref = 0;
            if (use_ref) {
                static const long tries[5] = {0, 1, -1, 2, -2};
                for (long t : tries) {
                    const long col = (long)(j - i) + shift + t;
                    if (col < 0) continue;
                    const size_t r = ref_line + (size_t)col;
                    if (r >= ref_end || j - r > 32768) continue;
                    const size_t lim = std::min({end - j, ref_end - r, (size_t)258});
                    size_t m = 0;
                    while (m < lim && p[r + m] == p[j + m]) ++m;
                    if (m > best) {
                        best = m;
                        best_ref = r;
                    }
                    if (best >= (size_t)kMinAligned) break;
                }
            }
            // a run of one byte
            size_t run = 1;
            if (j > 0 && p[j] == p[j - 1]) {
                const size_t lim = std::min(end - j, (size_t)258);
                while (run < lim && p[j + run] == p[j]) ++run;
            } else {
                run = 0;
            }
            if (best >= (size_t)kMinAligned && best >= run) {
                out.push_back(match_token((uint32_t)best, (uint32_t)(j - best_ref)));
                shift = (long)best_ref - (long)ref_line - (long)(j - i);
                j += best;
            } else if (run >= (size_t)kMinRun) {
                out.push_back(match_token((uint32_t)run, 1));
                j += run;
            } else {
                out.push_back(p[j]);
                ++j;
            }
        }
        line_start[0] = line_start[1];
        line_start[1] = line_start[2];
        line_start[2] = line_start[3];
        line_start[3] = i;
        i = end;
    }
}

size_t stored(const unsigned char *p, size_t n, unsigned char *out, size_t cap)
{
    size_t at = 0;
    do {
        const size_t m = std::min(n, (size_t)65535);
        if (cap - at < m + 5) return 0;
        out[at] = m == n ? 1 : 0; // BFINAL on the last, BTYPE 00
        out[at + 1] = (unsigned char)(m & 0xff);
        out[at + 2] = (unsigned char)(m >> 8);
        out[at + 3] = (unsigned char)(~m & 0xff);
        out[at + 4] = (unsigned char)((~m >> 8) & 0xff);
        memcpy(out + at + 5, p, m);
        at += 5 + m;
        p += m;
        n -= m;
    } while (n);
    return at;
}
} // namespace

size_t fq_deflate(const char *text, size_t n, unsigned char *out, size_t cap)
{
    const unsigned char *p = (const unsigned char *)text;
    if (n == 0) {
        if (cap < 2) return 0;
        out[0] = 3; // final, fixed codes, end-of-block
        out[1] = 0;
        return 2;
    }
    static thread_local Scratch scratch;
    std::vector<uint32_t> &tok = scratch.tokens;
    tokenize(p, n, tok);

    uint32_t lfreq[286] = {0}, dfreq[30] = {0};
    for (uint32_t t : tok) {
        if (t >> 31) {
            lfreq[257 + maps.len_sym[(t >> 15) & 0xff]]++;
            dfreq[maps.dist(t & 0x7fff)]++;
        } else {
            lfreq[t]++;
        }
    }
    lfreq[256] = 1;
    int dused = 0;
    for (int s = 0; s < 30; ++s) dused += dfreq[s] != 0;
    if (dused < 2) { // always two distance codes, like zlib: the code is complete for every reader
        for (int s = 0; s < 30 && dused < 2; ++s)
            if (!dfreq[s]) {
                dfreq[s] = 1;
                ++dused;
            }
    }
    uint8_t llen[286], dlen[30];
    uint16_t lcode[286], dcode[30];
    huffman(lfreq, 286, 15, llen, lcode);
    huffman(dfreq, 30, 15, dlen, dcode);

    // the code lengths, run-length coded with the code length alphabet
    int hlit = 286, hdist = 30;
    while (hlit > 257 && llen[hlit - 1] == 0) --hlit;
    while (hdist > 1 && dlen[hdist - 1] == 0) --hdist;
    uint8_t all[316];
    memcpy(all, llen, (size_t)hlit);
    memcpy(all + hlit, dlen, (size_t)hdist);
    const int total = hlit + hdist;
    struct Cl {
        uint8_t sym, extra_bits, extra;
    };
    Cl cl[316];
    int ncl = 0;
    uint32_t clfreq[19] = {0};
    for (int i = 0; i < total;) {
        int run = 1;
        while (i + run < total && all[i + run] == all[i]) ++run;
        const uint8_t v = all[i];
        int left = run;
        if (v == 0) {
            while (left >= 11) {
                const int r = std::min(left, 138);
                cl[ncl++] = {18, 7, (uint8_t)(r - 11)};
                left -= r;
            }
            if (left >= 3) {
                cl[ncl++] = {17, 3, (uint8_t)(left - 3)};
                left = 0;
            }
        } else {
            cl[ncl++] = {v, 0, 0};
            --left;
            while (left >= 3) {
                const int r = std::min(left, 6);
                cl[ncl++] = {16, 2, (uint8_t)(r - 3)};
                left -= r;
            }
        }
        while (left-- > 0) cl[ncl++] = {v, 0, 0};
        i += run;
    }
    for (int i = 0; i < ncl; ++i) clfreq[cl[i].sym]++;
    uint8_t cllen[19];
    uint16_t clcode[19];
    huffman(clfreq, 19, 7, cllen, clcode);
    int hclen = 19;
    while (hclen > 4 && cllen[kClOrder[hclen - 1]] == 0) --hclen;

    BitWriter w{out, out + cap};
    w.put(1, 1); // BFINAL
    w.put(2, 2); // dynamic codes
    w.put((uint32_t)(hlit - 257), 5);
    w.put((uint32_t)(hdist - 1), 5);
    w.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) w.put(cllen[kClOrder[i]], 3);
    for (int i = 0; i < ncl; ++i) {
        w.put(clcode[cl[i].sym], cllen[cl[i].sym]);
        if (cl[i].extra_bits) w.put(cl[i].extra, cl[i].extra_bits);
    }
    for (uint32_t t : tok) {
        if (t >> 31) {
            const uint32_t l3 = (t >> 15) & 0xff, d1 = t & 0x7fff;
            const int ls = maps.len_sym[l3], ds = maps.dist(d1);
            w.put(lcode[257 + ls], llen[257 + ls]);
            if (kLenExtra[ls]) w.put(l3 + 3 - kLenBase[ls], kLenExtra[ls]);
            w.put(dcode[ds], dlen[ds]);
            if (kDistExtra[ds]) w.put(d1 + 1 - kDistBase[ds], kDistExtra[ds]);
        } else {
            w.put(lcode[t], llen[t]);
        }
    }
    w.put(lcode[256], llen[256]);
    w.flush();
    const size_t made = (size_t)(w.p - out);
    if (w.overflow || made >= n + 5 * ((n + 65534) / 65535)) return stored(p, n, out, cap); // not compressible
    return made;
}

// ---------------------------------------------------------------- BGZF
const unsigned char kBgzfEofBlock[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

namespace {
constexpr size_t kBgzfMaxBlock = 0x10000;

struct Deflater { // one raw-deflate state per worker thread and level, reset per block
    z_stream zs;
    int level = -100;
    bool live = false;
    ~Deflater()
    {
        if (live) deflateEnd(&zs);
    }
    bool prepare(int lvl)
    {
        if (live && lvl == level) return deflateReset(&zs) == Z_OK;
        if (live) deflateEnd(&zs);
        memset(&zs, 0, sizeof zs);
        live = deflateInit2(&zs, lvl, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        level = lvl;
        return live;
    }
};
} // namespace

void bgzf_append_block(const char *p, size_t n, int level, std::string &out)
{
    const size_t at = out.size();
    out.resize(at + kBgzfMaxBlock);
    unsigned char *blk = (unsigned char *)out.data() + at;
    memcpy(blk, kBgzfEofBlock, 16);
    const size_t room = kBgzfMaxBlock - 18 - 8;
    size_t clen = 0;
    if (level < 0) {
        clen = fq_deflate(p, n, blk + 18, room);
    } else {
        static thread_local Deflater d;
        if (d.prepare(level)) {
            d.zs.next_in = (Bytef *)p;
            d.zs.avail_in = (uInt)n;
            d.zs.next_out = blk + 18;
            d.zs.avail_out = (uInt)room;
            if (deflate(&d.zs, Z_FINISH) == Z_STREAM_END) clen = d.zs.total_out;
        }
    }
    if (clen == 0) clen = stored((const unsigned char *)p, n, blk + 18, room); // always fits: n <= kBgzfInput
    const uint32_t total = (uint32_t)(18 + clen + 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)p, (uInt)n);
    blk[16] = (unsigned char)((total - 1) & 0xff);
    blk[17] = (unsigned char)((total - 1) >> 8);
    unsigned char *tail = blk + 18 + clen;
    for (int i = 0; i < 4; ++i) tail[i] = (unsigned char)(crc >> (8 * i));
    for (int i = 0; i < 4; ++i) tail[4 + i] = (unsigned char)((uint32_t)n >> (8 * i));
    out.resize(at + total);
}
