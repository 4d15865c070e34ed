#include "GzInflater.h"

#include <zlib.h> // crc32, crc32_combine only

#include <algorithm>
#include <cstring>

#include "WorkerPool.h"

using namespace deflate_detail;


GzInflater::GzInflater(const unsigned char *data, size_t size) : DeflateStream(data, size), win(kWindow + kChunk + kSlack) {}

bool GzInflater::member_header()
{
    bool none = false;
    if (!gzip_header(first_member, &none)) return false;
    if (none) {
        state = DONE;
        return true;
    }
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

bool GzInflater::block_header()
{
    const int kind = next_block();
    if (kind < 0) return false;
    state = kind == 0 ? STORED : HUFFMAN;
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
        const uint32_t c = deflate_parallel_crc32(p + seg, n);
        const uint32_t whole = (uint32_t)crc32_combine(crc_running, c, (z_off_t)n);
        if (ok && (whole != m.crc || (uint32_t)(len_running + n) != m.isize)) {
            ok = false;
            fail(whole != m.crc ? "incorrect data check" : "incorrect length check");
            state = FAILED;
        }
        crc_running = 0;
        len_running = 0;
        seg = end;
    }
    ends.erase(ends.begin(), ends.begin() + (long)done);
    if (seg < produced) {
        const size_t n = produced - seg;
        crc_running = (uint32_t)crc32_combine(crc_running, deflate_parallel_crc32(p + seg, n), (z_off_t)n);
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
            if (!ok) {
                state = FAILED;
                break;
            }
        }
        if (opos == rpos && (state == DONE || state == FAILED)) break;
    }
    verify(dst, produced);
    return produced;
}
