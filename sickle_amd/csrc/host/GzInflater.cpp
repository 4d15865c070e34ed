#include "GzInflater.h"

#include <zlib.h> // crc32, crc32_combine only

#include <algorithm>
#include <cstring>

#include "WorkerPool.h"

namespace {
// table entry: bits 0-5 code bits to drop (the shift count as the CPU takes it), 8-12 extra bits
// (subtable index bits for SUB, length of the first code for LIT2), 13-15 kind, 16-31 literal /
// two literals / base value / subtable offset
enum Kind : uint32_t { LIT = 0, LIT2 = 1, BASE = 2, EOB = 3, SUB = 4, INVALID = 5 };
inline uint32_t entry(uint32_t nbits, Kind k, uint32_t extra, uint32_t value) { return nbits | (extra << 8) | (k << 13) | (value << 16); }
inline uint32_t e_nbits(uint32_t e) { return e & 63; }
inline uint32_t e_kind(uint32_t e) { return (e >> 13) & 7; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint64_t load64(const unsigned char *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v; // little-endian host (x86-64)
}
inline void store64(unsigned char *p, uint64_t v) { memcpy(p, &v, 8); }
inline void store16(unsigned char *p, uint16_t v) { memcpy(p, &v, 2); }

uint32_t parallel_crc(const unsigned char *p, size_t n)
{
    if (n < (4u << 20)) return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
    WorkerPool &pool = WorkerPool::instance();
    const size_t parts = std::min<size_t>((size_t)pool.size() * 2, n / (1u << 20));
    std::vector<uint32_t> crc(parts);
    std::vector<size_t> len(parts);
    pool.parallel_for(n, parts, [&](size_t b, size_t e, size_t part) {
        uLong c = crc32(0L, Z_NULL, 0);
        for (size_t at = b; at < e;) { // crc32 takes a 32-bit length
            const size_t m = std::min<size_t>(e - at, 1u << 30);
            c = crc32(c, p + at, (uInt)m);
            at += m;
        }
        crc[part] = (uint32_t)c;
        len[part] = e - b;
    });
    uLong c = crc[0];
    for (size_t i = 1; i < parts; ++i) c = crc32_combine(c, crc[i], (z_off_t)len[i]);
    return (uint32_t)c;
}
} // namespace

GzInflater::GzInflater(const unsigned char *data, size_t size) : in(data), in_end(data + size), win(kWindow + kChunk + kSlack) {}

bool GzInflater::fail(const char *what)
{
    if (!err) err = what;
    state = FAILED;
    return false;
}

// ---- careful bit reader (headers, and the decode loops near the end of the input)
bool GzInflater::need_bits(int n)
{
    while (bitcnt <= 56 && in < in_end) {
        bitbuf |= (uint64_t)*in++ << bitcnt;
        bitcnt += 8;
    }
    return bitcnt >= n;
}

uint32_t GzInflater::take_bits(int n)
{
    const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1));
    bitbuf >>= n;
    bitcnt -= n;
    return v;
}

// drops the rest of the current byte and hands the whole bytes still in the buffer back to `in`
void GzInflater::align_to_byte()
{
    const int drop = bitcnt & 7;
    bitcnt -= drop;
    in -= bitcnt >> 3;
    bitbuf = 0;
    bitcnt = 0;
}

bool GzInflater::member_header()
{
    // between members the bit buffer is empty
    const size_t left = (size_t)(in_end - in);
    if (!first_member && (left < 2 || in[0] != 0x1f || in[1] != 0x8b)) { // zlib's reader: trailing bytes are ignored
        state = DONE;
        return true;
    }
    if (left == 0) {
        state = DONE;
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
    first_member = false;
    mstart = opos;
    state = BLOCK_HEADER;
    return true;
}

bool GzInflater::member_trailer()
{
    align_to_byte();
    if (in_end - in < 8) return fail("unexpected end of file");
    MemberEnd m;
    m.abs_off = abs_handed + (opos - rpos);
    m.crc = (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24);
    m.isize = (uint32_t)in[4] | ((uint32_t)in[5] << 8) | ((uint32_t)in[6] << 16) | ((uint32_t)in[7] << 24);
    in += 8;
    ends.push_back(m);
    state = MEMBER_HEADER;
    return true;
}

// Canonical Huffman decode table.  Codes of up to primary_bits index the table directly (an entry
// repeated for every value of the bits above its code); longer ones go through a subtable per
// primary prefix.
bool GzInflater::build(const uint8_t *lens, int n, uint32_t *table, int primary_bits, bool dist)
{
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return fail("invalid code lengths set");
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

void GzInflater::fixed_tables()
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

bool GzInflater::dynamic_tables()
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

bool GzInflater::block_header()
{
    if (!need_bits(3)) return fail("unexpected end of file");
    last_block = take_bits(1) != 0;
    const uint32_t type = take_bits(2);
    if (type == 0) {
        align_to_byte();
        if (in_end - in < 4) return fail("unexpected end of file");
        const uint32_t len = (uint32_t)in[0] | ((uint32_t)in[1] << 8), nlen = (uint32_t)in[2] | ((uint32_t)in[3] << 8);
        if ((len ^ 0xffffu) != nlen) return fail("invalid stored block lengths");
        in += 4;
        stored_left = len;
        state = STORED;
        return true;
    }
    if (type == 1) fixed_tables();
    else if (type == 2) {
        if (!dynamic_tables()) return false;
    } else {
        return fail("invalid block type");
    }
    state = HUFFMAN;
    return true;
}

bool GzInflater::run_stored(size_t olimit)
{
    const size_t n = std::min({stored_left, olimit - opos, (size_t)(in_end - in)});
    memcpy(&win[opos], in, n);
    in += n;
    opos += n;
    stored_left -= n;
    if (stored_left == 0) state = last_block ? MEMBER_TRAILER : BLOCK_HEADER;
    else if (in == in_end) return fail("unexpected end of file");
    return true;
}

// Decodes symbols until the block ends or opos reaches olimit (a match may run past olimit by
// less than kSlack; the buffer has that room).  FAST: at least 16 input bytes are ahead, so the
// refill is one unaligned load and a run of literals is decoded from one refill without looking at
// the bit count; otherwise (the last bytes of the input) one symbol per refill, every count checked.
template <bool FAST>
bool GzInflater::huffman_loop(size_t olimit, bool &block_done)
{
    unsigned char *const base = win.data();
    unsigned char *o = base + opos;
    unsigned char *const oend = base + olimit;
    const unsigned char *const lowest = base + mstart;
    const uint32_t *const lt = lit_table, *const dt = dist_table;
    constexpr uint64_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;
    const unsigned char *ip = in;
    uint64_t bb = bitbuf;
    int bc = bitcnt;
    const char *problem = nullptr;

#define SK_REFILL()                                 \
    do {                                            \
        bb |= load64(ip) << bc; /* to 56..63 bits */ \
        ip += (63 - bc) >> 3;                       \
        bc |= 56;                                   \
    } while (0)
#define SK_DROP(e)             \
    do {                       \
        bb >>= e_nbits(e);     \
        bc -= (int)e_nbits(e); \
    } while (0)
#define SK_PUT_LITERALS(e)                                                                         \
    do {                                                                                           \
        store16(o, (uint16_t)e_value(e)); /* one or two; the second byte is scratch for LIT */ \
        o += 1 + e_kind(e);                                                                        \
    } while (0)

    while (o < oend) {
        if (FAST) {
            if (in_end - ip < 16) break;
            SK_REFILL();
        } else {
            while (bc <= 56 && ip < in_end) {
                bb |= (uint64_t)*ip++ << bc;
                bc += 8;
            }
        }
        uint32_t e = lt[bb & lmask];
        if (FAST && __builtin_expect(e_kind(e) <= LIT2, 1)) {
            // up to four primary entries (<= 11 bits each) from one refill of >= 56 bits
            SK_DROP(e);
            SK_PUT_LITERALS(e);
            e = lt[bb & lmask];
            if (__builtin_expect(e_kind(e) <= LIT2, 1)) {
                SK_DROP(e);
                SK_PUT_LITERALS(e);
                e = lt[bb & lmask];
                if (__builtin_expect(e_kind(e) <= LIT2, 1)) {
                    SK_DROP(e);
                    SK_PUT_LITERALS(e);
                    e = lt[bb & lmask];
                    if (__builtin_expect(e_kind(e) <= LIT2, 1)) {
                        SK_DROP(e);
                        SK_PUT_LITERALS(e);
                        continue;
                    }
                }
            }
            SK_REFILL(); // e is something else and still in the buffer: room for a whole match
        }
        if (e_kind(e) == SUB) {
            bb >>= kLitBits;
            bc -= kLitBits;
            e = lt[e_value(e) + (bb & ((1u << e_extra(e)) - 1))];
        }
        SK_DROP(e);
        if (e_kind(e) <= LIT2) {
            SK_PUT_LITERALS(e);
            if (__builtin_expect(bc < 0, 0)) {
                problem = "unexpected end of file";
                break;
            }
            continue;
        }
        if (e_kind(e) == EOB) {
            if (bc < 0) problem = "unexpected end of file";
            block_done = true;
            break;
        }
        if (__builtin_expect(e_kind(e) != BASE, 0)) {
            problem = "invalid literal/length code";
            break;
        }
        const uint32_t len = e_value(e) + (uint32_t)(bb & ((1u << e_extra(e)) - 1));
        bb >>= e_extra(e);
        bc -= (int)e_extra(e);
        uint32_t d = dt[bb & dmask];
        if (__builtin_expect(e_kind(d) == SUB, 0)) {
            bb >>= kDistBits;
            bc -= kDistBits;
            d = dt[e_value(d) + (bb & ((1u << e_extra(d)) - 1))];
        }
        SK_DROP(d);
        if (__builtin_expect(e_kind(d) != BASE, 0)) {
            problem = "invalid distance code";
            break;
        }
        const uint32_t dist = e_value(d) + (uint32_t)(bb & ((1u << e_extra(d)) - 1));
        bb >>= e_extra(d);
        bc -= (int)e_extra(d);
        if (__builtin_expect(bc < 0, 0)) {
            problem = "unexpected end of file";
            break;
        }
        if (__builtin_expect((size_t)(o - lowest) < dist, 0)) {
            problem = "invalid distance too far back";
            break;
        }
        const unsigned char *src = o - dist;
        unsigned char *const stop = o + len;
        if (dist >= 8) {
            do {
                store64(o, load64(src));
                o += 8;
                src += 8;
            } while (o < stop);
        } else if (dist == 1) {
            memset(o, *src, len);
        } else {
            do *o++ = *src++;
            while (o < stop);
        }
        o = stop;
    }
#undef SK_REFILL
#undef SK_DROP
#undef SK_PUT_LITERALS
    in = ip;
    bitbuf = bb;
    bitcnt = bc < 0 ? 0 : bc;
    opos = (size_t)(o - base);
    if (problem) return fail(problem);
    return true;
}

bool GzInflater::run_huffman(size_t olimit)
{
    bool block_done = false;
    if (!huffman_loop<true>(olimit, block_done)) return false;
    if (!block_done && opos < olimit && !huffman_loop<false>(olimit, block_done)) return false; // the input's last bytes
    if (block_done) state = last_block ? MEMBER_TRAILER : BLOCK_HEADER;
    return true;
}

// Checks the members that ended inside [abs_handed - produced, abs_handed) and carries the CRC of
// the unfinished one forward.
bool GzInflater::verify(const char *dst, size_t produced)
{
    const uint64_t call_start = abs_handed - produced;
    const unsigned char *p = (const unsigned char *)dst;
    size_t seg = 0;
    size_t done = 0;
    bool ok = true;
    for (; done < ends.size() && ends[done].abs_off <= abs_handed; ++done) {
        const MemberEnd &m = ends[done];
        const size_t end = (size_t)(m.abs_off - call_start);
        const size_t n = end - seg;
        const uint32_t c = parallel_crc(p + seg, n);
        const uint32_t whole = (uint32_t)crc32_combine(crc_running, c, (z_off_t)n);
        if (ok && (whole != m.crc || (uint32_t)(len_running + n) != m.isize)) {
            ok = false;
            fail(whole != m.crc ? "incorrect data check" : "incorrect length check");
        }
        crc_running = 0;
        len_running = 0;
        seg = end;
    }
    ends.erase(ends.begin(), ends.begin() + (long)done);
    if (seg < produced) {
        const size_t n = produced - seg;
        crc_running = (uint32_t)crc32_combine(crc_running, parallel_crc(p + seg, n), (z_off_t)n);
        len_running += n;
    }
    return ok;
}

size_t GzInflater::read(char *dst, size_t want)
{
    size_t produced = 0;
    while (produced < want) {
        if (rpos < opos) {
            const size_t n = std::min(opos - rpos, want - produced);
            memcpy(dst + produced, &win[rpos], n);
            rpos += n;
            produced += n;
            abs_handed += n;
            continue;
        }
        if (state == DONE || state == FAILED) break;
        if (opos > kWindow) { // keep the last 32 KiB as history at the front
            const size_t shift = opos - kWindow;
            memmove(&win[0], &win[shift], kWindow);
            opos = rpos = kWindow;
            mstart = mstart > shift ? mstart - shift : 0;
        }
        const size_t olimit = opos + std::min(want - produced, kChunk);
        while (opos < olimit && state != DONE && state != FAILED) {
            bool ok = true;
            switch (state) {
            case MEMBER_HEADER: ok = member_header(); break;
            case BLOCK_HEADER: ok = block_header(); break;
            case STORED: ok = run_stored(olimit); break;
            case HUFFMAN: ok = run_huffman(olimit); break;
            case MEMBER_TRAILER: ok = member_trailer(); break;
            default: break;
            }
            if (!ok) break;
        }
        if (opos == rpos && (state == DONE || state == FAILED)) break;
    }
    verify(dst, produced);
    return produced;
}
