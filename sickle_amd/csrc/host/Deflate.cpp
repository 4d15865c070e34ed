#include "Deflate.h"

#include <zlib.h> // crc32, crc32_combine only

#include <algorithm>
#include <vector>

#include "WorkerPool.h"

using namespace deflate_detail;

// CRC-32 (the gzip polynomial, reflected 0xedb88320), eight bytes per step through eight tables
// ("slicing by 8"); zlib 1.2.11's crc32 works four bytes at a time.
namespace {
struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
};
const CrcTables crc_tables;

uint32_t crc32_fast(const unsigned char *p, size_t n)
{
    const uint32_t(*t)[256] = crc_tables.t;
    uint32_t c = 0xffffffffu;
    while (n && ((uintptr_t)p & 7)) {
        c = (c >> 8) ^ t[0][(c ^ *p++) & 0xff];
        --n;
    }
    while (n >= 8) {
        const uint64_t v = load64(p) ^ c;
        c = t[7][v & 0xff] ^ t[6][(v >> 8) & 0xff] ^ t[5][(v >> 16) & 0xff] ^ t[4][(v >> 24) & 0xff] ^ t[3][(v >> 32) & 0xff] ^
            t[2][(v >> 40) & 0xff] ^ t[1][(v >> 48) & 0xff] ^ t[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ t[0][(c ^ *p++) & 0xff];
    return ~c;
}
} // namespace

uint32_t deflate_crc32(const unsigned char *p, size_t n) { return crc32_fast(p, n); }

// CRC-32 of a large buffer on all host threads (slices combined with crc32_combine)
uint32_t deflate_parallel_crc32(const unsigned char *p, size_t n)
{
    if (n < (4u << 20)) return crc32_fast(p, n);
    WorkerPool &pool = WorkerPool::instance();
    const size_t parts = std::min<size_t>((size_t)pool.size() * 2, n / (1u << 20));
    std::vector<uint32_t> crc(parts);
    std::vector<size_t> len(parts);
    pool.parallel_for(n, parts, [&](size_t b, size_t e, size_t part) {
        crc[part] = crc32_fast(p + b, e - b);
        len[part] = e - b;
    });
    uLong c = crc[0];
    for (size_t i = 1; i < parts; ++i) c = crc32_combine(c, crc[i], (z_off_t)len[i]);
    return (uint32_t)c;
}

// ---- careful bit reader (headers, and the decode loops near the end of the input)
bool DeflateStream::need_bits(int n)
{
    while (bitcnt <= 56 && in < in_end) {
        bitbuf |= (uint64_t)*in++ << bitcnt;
        bitcnt += 8;
    }
    return bitcnt >= n;
}

uint32_t DeflateStream::take_bits(int n)
{
    const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1));
    bitbuf >>= n;
    bitcnt -= n;
    return v;
}

// drops the rest of the current byte and hands the whole bytes still in the buffer back to `in`
void DeflateStream::align_to_byte()
{
    const int drop = bitcnt & 7;
    bitcnt -= drop;
    in -= bitcnt >> 3;
    bitbuf = 0;
    bitcnt = 0;
}

bool DeflateStream::gzip_header(bool first_member, bool *none)
{
    // between members the bit buffer is empty
    *none = false;
    const size_t left = (size_t)(in_end - in);
    if (left == 0 || (!first_member && (left < 2 || in[0] != 0x1f || in[1] != 0x8b))) { // zlib's reader: trailing bytes are ignored
        *none = true;
        return true;
    }
    if (left < 10 || in[0] != 0x1f || in[1] != 0x8b) return fail(first_member ? "not in gzip format" : "unexpected end of file");
    if (in[2] != 8) return fail("unknown compression method");
    const unsigned flg = in[3];
    if (flg & 0xe0) return fail("unknown header flags set");
    const unsigned char *p = in + 10;
    if (flg & 4) { // FEXTRA
        if (in_end - p < 2) return fail("unexpected end of file");
        const size_t xlen = (size_t)p[0] | ((size_t)p[1] << 8);
        p += 2;
        if ((size_t)(in_end - p) < xlen) return fail("unexpected end of file");
        p += xlen;
    }
    for (unsigned bit = 8; bit <= 16; bit <<= 1) // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            const unsigned char *z = (const unsigned char *)memchr(p, 0, (size_t)(in_end - p));
            if (!z) return fail("unexpected end of file");
            p = z + 1;
        }
    if (flg & 2) { // FHCRC
        if (in_end - p < 2) return fail("unexpected end of file");
        p += 2;
    }
    in = p;
    return true;
}

// Canonical Huffman decode table.  Codes of up to primary_bits index the table directly (an entry
// repeated for every value of the bits above its code); longer ones go through a subtable per
// primary prefix.
bool DeflateStream::build(const uint8_t *lens, int n, uint32_t *table, int primary_bits, bool dist)
{
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return fail("invalid code lengths set");
    }
    if (dist) {
        dist_complete = left == 0;
        dist_codes = 0;
        for (int l = 1; l <= 15; ++l) dist_codes += count[l];
    } else {
        lit_complete = left == 0;
    }
    uint32_t next_code[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next_code[l] = code;
    }
    const uint32_t psize = 1u << primary_bits;
    for (uint32_t i = 0; i < psize; ++i) table[i] = entry(0, INVALID, 0, 0);
    // bit-reversed code of every symbol, and the longest code under each primary prefix
    uint16_t rev[320];
    uint8_t longest[1 << kLitBits];
    memset(longest, 0, psize);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t c = next_code[l]++, r = 0;
        for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
        rev[s] = (uint16_t)r;
        if (l > primary_bits) {
            uint8_t &m = longest[r & (psize - 1)];
            if (l > m) m = (uint8_t)l;
        }
    }
    uint32_t sub_next = psize;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t e;
        if (!dist) {
            if (s < 256) e = entry(0, LIT, 0, (uint32_t)s);
            else if (s == 256) e = entry(0, EOB, 0, 0);
            else if (s < 286) e = entry(0, BASE, kLenExtra[s - 257], kLenBase[s - 257]);
            else e = entry(0, INVALID, 0, 0);
        } else {
            e = s < 30 ? entry(0, BASE, kDistExtra[s], kDistBase[s]) : entry(0, INVALID, 0, 0);
        }
        const uint32_t r = rev[s];
        if (l <= primary_bits) {
            e |= (uint32_t)l;
            for (uint32_t i = r; i < psize; i += 1u << l) table[i] = e;
        } else {
            const uint32_t prefix = r & (psize - 1);
            const int sub_bits = longest[prefix] - primary_bits;
            if (e_kind(table[prefix]) != SUB) {
                table[prefix] = entry((uint32_t)primary_bits, SUB, (uint32_t)sub_bits, sub_next);
                for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[sub_next + i] = entry(0, INVALID, 0, 0);
                sub_next += 1u << sub_bits;
            }
            const uint32_t base = e_value(table[prefix]);
            e |= (uint32_t)(l - primary_bits);
            for (uint32_t i = r >> primary_bits; i < (1u << sub_bits); i += 1u << (l - primary_bits)) table[base + i] = e;
        }
    }
    if (!dist) {
        // Two literals per lookup where both codes fit the primary index: FASTQ text is mostly
        // literals with short codes (bases 2-3 bits, qualities 4-6), and the decode loop is one
        // dependent table lookup per entry whatever the entry yields.  In place: an entry already
        // turned into a pair still says what its first literal and that literal's length are.
        for (uint32_t i = 0; i < psize; ++i) {
            const uint32_t e1 = table[i];
            if (e_kind(e1) != LIT) continue;
            const uint32_t l1 = e_nbits(e1);
            const uint32_t e2 = table[i >> l1];
            uint32_t l2, lit2;
            if (e_kind(e2) == LIT) {
                l2 = e_nbits(e2);
                lit2 = e_value(e2);
            } else if (e_kind(e2) == LIT2) {
                l2 = e_extra(e2);
                lit2 = e_value(e2) & 0xff;
            } else {
                continue;
            }
            if (l1 + l2 > (uint32_t)primary_bits) continue;
            table[i] = entry(l1 + l2, LIT2, l1, e_value(e1) | (lit2 << 8));
        }
    }
    return true;
}

void DeflateStream::fixed_tables()
{
    if (tables_are_fixed) return;
    uint8_t lens[288];
    for (int i = 0; i < 144; ++i) lens[i] = 8;
    for (int i = 144; i < 256; ++i) lens[i] = 9;
    for (int i = 256; i < 280; ++i) lens[i] = 7;
    for (int i = 280; i < 288; ++i) lens[i] = 8;
    build(lens, 288, lit_table, kLitBits, false);
    uint8_t dl[32];
    for (int i = 0; i < 32; ++i) dl[i] = 5;
    build(dl, 32, dist_table, kDistBits, true);
    tables_are_fixed = true;
}

bool DeflateStream::dynamic_tables()
{
    tables_are_fixed = false;
    if (!need_bits(14)) return fail("unexpected end of file");
    const int hlit = (int)take_bits(5) + 257, hdist = (int)take_bits(5) + 1, hclen = (int)take_bits(4) + 4;
    if (hlit > 286 || hdist > 30) return fail("too many length or distance symbols");
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
        if (!need_bits(3)) return fail("unexpected end of file");
        cl[kClOrder[i]] = (uint8_t)take_bits(3);
    }
    uint32_t cl_table[1 << 7]; // the code length code: at most 7 bits, one flat table of symbols
    {
        int count[8] = {0};
        for (int i = 0; i < 19; ++i) count[cl[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l <= 7; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return fail("invalid code lengths set");
        }
        uint32_t next_code[8], code = 0;
        for (int l = 1; l <= 7; ++l) {
            code = (code + (uint32_t)count[l - 1]) << 1;
            next_code[l] = code;
        }
        for (int i = 0; i < 128; ++i) cl_table[i] = entry(0, INVALID, 0, 0);
        for (int s = 0; s < 19; ++s) {
            const int l = cl[s];
            if (!l) continue;
            uint32_t c = next_code[l]++, r = 0;
            for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
            for (uint32_t i = r; i < 128; i += 1u << l) cl_table[i] = entry((uint32_t)l, LIT, 0, (uint32_t)s);
        }
    }
    uint8_t lens[320];
    int have = 0;
    const int total = hlit + hdist;
    while (have < total) {
        need_bits(14); // a code (<= 7 bits) and its repeat count (<= 7 bits); checked after use
        const uint32_t e = cl_table[bitbuf & 127];
        if (e_kind(e) != LIT) return fail("invalid code lengths set");
        bitbuf >>= e_nbits(e);
        bitcnt -= (int)e_nbits(e);
        if (bitcnt < 0) return fail("unexpected end of file");
        const uint32_t sym = e_value(e);
        if (sym < 16) {
            lens[have++] = (uint8_t)sym;
            continue;
        }
        int rep, extra, base;
        uint8_t fill = 0;
        if (sym == 16) {
            if (have == 0) return fail("invalid bit length repeat");
            fill = lens[have - 1];
            extra = 2;
            base = 3;
        } else if (sym == 17) {
            extra = 3;
            base = 3;
        } else {
            extra = 7;
            base = 11;
        }
        rep = base + (int)take_bits(extra);
        if (bitcnt < 0) return fail("unexpected end of file");
        if (have + rep > total) return fail("invalid bit length repeat");
        while (rep--) lens[have++] = fill;
    }
    if (lens[256] == 0) return fail("invalid code -- missing end-of-block");
    if (!build(lens, hlit, lit_table, kLitBits, false)) return false;
    if (!build(lens + hlit, hdist, dist_table, kDistBits, true)) return false;
    return true;
}

int DeflateStream::next_block()
{
    if (!need_bits(3)) return fail("unexpected end of file"), -1;
    last_block = take_bits(1) != 0;
    const uint32_t type = take_bits(2);
    if (type == 0) {
        align_to_byte();
        if (in_end - in < 4) return fail("unexpected end of file"), -1;
        const uint32_t len = (uint32_t)in[0] | ((uint32_t)in[1] << 8), nlen = (uint32_t)in[2] | ((uint32_t)in[3] << 8);
        if ((len ^ 0xffffu) != nlen) return fail("invalid stored block lengths"), -1;
        in += 4;
        stored_left = len;
        return 0;
    }
    if (type == 1) fixed_tables();
    else if (type == 2) {
        if (!dynamic_tables()) return -1;
    } else {
        return fail("invalid block type"), -1;
    }
    return 1;
}
