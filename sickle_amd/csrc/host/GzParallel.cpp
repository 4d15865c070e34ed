#include "GzParallel.h"

#include <zlib.h> // crc32_combine only

#include <algorithm>
#include <cstring>

#include "Deflate.h"
#include "WorkerPool.h"

using namespace deflate_detail;

namespace {
constexpr size_t kWindow = 32768;
constexpr uint16_t kUnknown = 256; // symbol 256 + j = byte j of the window before the stretch
} // namespace

// One stretch of the stream: from a block header (known, or guessed by find_start) to the block
// boundary it stopped at, decoded into 16-bit symbols.
class GzParallel::Stretch : public DeflateStream {
public:
    enum End { NOTHING, AT_STOP, AT_LIMIT, STREAM_END, FAILED, DROPPED };
    struct Mark { // a gzip member ended after n_sym symbols of this stretch; its trailer said:
        size_t n_sym;
        uint32_t crc, isize;
    };
    std::vector<Mark> marks;
    Stretch(const unsigned char *d, size_t n) : DeflateStream(d, n), base(d), total(n) {}

    std::vector<uint16_t> sym; // [window: known bytes or placeholders | decoded symbols]
    size_t n_sym = 0;
    uint64_t begin_bit = 0, end_bit = 0;
    End end = NOTHING;
    size_t cap_symbols = SIZE_MAX;  // guessed starts: beyond this the stretch is taken to decode nonsense
    size_t soft_cap = SIZE_MAX;     // any stretch: stop at the next block boundary beyond this

    uint64_t bitpos() const { return (uint64_t)(in - base) * 8 - (uint64_t)bitcnt; }
    const char *problem() const { return err; }

    void start_known(uint64_t bit, const unsigned char *window)
    {
        prepare();
        for (size_t j = 0; j < kWindow; ++j) sym[j] = window[j];
        seek_bits(bit);
        begin_bit = bit;
        cap_symbols = SIZE_MAX;
    }

    // Guessed start: the first bit position in [from, to) where a non-final dynamic block header
    // with complete codes parses, its block decodes, and another block header follows.
    bool find_start(uint64_t from, uint64_t to)
    {
        prepare();
        for (size_t j = 0; j < kWindow; ++j) sym[j] = (uint16_t)(kUnknown + j);
        const uint64_t last = total * 8 > 128 ? total * 8 - 128 : 0;
        if (to > last) to = last;
        for (uint64_t p = from; p < to; ++p) {
            if (!plausible(p)) continue;
            seek_bits(p);
            n_sym = 0;
            if (next_block() == 1 && lit_complete && (dist_complete || dist_codes <= 1) && huffman_block()) {
                // what follows must parse too: another block header, or (after a final block) the
                // member's trailer and the next member's gzip header
                if (last_block) {
                    if (hop_member() == 1) {
                        begin_bit = p;
                        return true;
                    }
                } else {
                    boundary_bit = bitpos();
                    header_kind = next_block();
                    if (header_kind >= 0) {
                        header_ready = true;
                        begin_bit = p;
                        return true;
                    }
                }
            }
            marks.clear();
            err = nullptr;
            end = NOTHING;
        }
        return false;
    }

    // Decodes block after block until a boundary that is one of `stops`, or at/after `limit`, or
    // the end of the member.
    void run(const std::vector<uint64_t> &stops, uint64_t limit)
    {
        for (;;) {
            const uint64_t p = header_ready ? boundary_bit : bitpos();
            if (p >= limit) return stop(AT_LIMIT, p);
            if (std::binary_search(stops.begin(), stops.end(), p)) return stop(AT_STOP, p);
            // very compressible data (runs): end the stretch at this boundary rather than let its
            // symbols grow without bound; the next round starts here
            if (n_sym >= soft_cap) return stop(AT_LIMIT, p);
            int kind;
            if (header_ready) {
                kind = header_kind;
                header_ready = false;
            } else {
                kind = next_block();
            }
            if (kind < 0) return stop(FAILED, p);
            if (!(kind == 0 ? stored_block() : huffman_block())) return stop(end == DROPPED ? DROPPED : FAILED, p);
            if (last_block) {
                const uint64_t e = bitpos();
                const int hop = hop_member();
                if (hop < 0) return stop(FAILED, e);
                if (hop == 0) return stop(STREAM_END, e);
            }
        }
    }

    // gzip member header at byte `at`: where the first block starts, or *none
    bool member_header(size_t at, bool first, bool *none, size_t *after)
    {
        in = base + at;
        bitbuf = 0;
        bitcnt = 0;
        err = nullptr;
        if (!gzip_header(first, none)) return false;
        *after = (size_t)(in - base);
        return true;
    }

private:
    void stop(End e, uint64_t at)
    {
        end = e;
        end_bit = at;
    }

    // after a final block: notes the member's trailer and steps over the next member's header.
    // 1 = at the first block of the next member, 0 = no further member, -1 = damaged
    int hop_member()
    {
        const size_t at = (size_t)((bitpos() + 7) / 8);
        if (total - at < 8) return fail("unexpected end of file"), -1;
        const unsigned char *t = base + at;
        Mark m;
        m.n_sym = n_sym;
        m.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        m.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        marks.push_back(m);
        in = base + at + 8;
        bitbuf = 0;
        bitcnt = 0;
        last_block = false;
        bool none = false;
        if (!gzip_header(false, &none)) return -1;
        return none ? 0 : 1;
    }

    void prepare()
    {
        if (sym.size() < kWindow + (1u << 20)) sym.resize(kWindow + (1u << 20));
        n_sym = 0;
        end = NOTHING;
        header_ready = false;
        err = nullptr;
        marks.clear();
    }

    void seek_bits(uint64_t p)
    {
        in = base + (p >> 3);
        bitbuf = 0;
        bitcnt = 0;
        err = nullptr;
        last_block = false;
        const int skip = (int)(p & 7);
        if (skip && need_bits(8)) take_bits(skip);
    }

    // cheap test of the fixed part of a dynamic block header and of its code length code
    bool plausible(uint64_t p) const
    {
        const unsigned char *q = base + (p >> 3);
        const int bit = (int)(p & 7);
        const uint64_t v = load64(q) >> bit;
        if ((v & 6) != 4) return false; // BTYPE 2 (BFINAL either way: small members are one final block)
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) return false;
        const int hclen = (int)((v >> 13) & 15) + 4;
        const uint64_t w = load64(q + 4) >> bit; // bits 32.. of the header
        unsigned kraft = 0, codes = 0;
        for (int i = 0; i < hclen; ++i) {
            const int at = 17 + 3 * i;
            const unsigned len = (unsigned)((at + 3 <= 56 ? v >> at : w >> (at - 32)) & 7);
            if (len) {
                kraft += 128u >> len;
                ++codes;
            }
        }
        return kraft == 128 || (codes == 1 && kraft == 64);
    }

    bool room()
    {
        if (n_sym > cap_symbols) { // a guessed start that turned out to decode nonsense at length
            end = DROPPED;
            return false;
        }
        if (sym.size() < kWindow + n_sym + 70000) sym.resize(sym.size() + sym.size() / 2 + 70000);
        return true;
    }

    bool stored_block()
    {
        while (stored_left) {
            if (!room()) return false;
            const size_t n = std::min({stored_left, (size_t)65536, (size_t)(in_end - in)});
            if (n == 0) return fail("unexpected end of file");
            uint16_t *o = sym.data() + kWindow + n_sym;
            for (size_t i = 0; i < n; ++i) o[i] = in[i];
            in += n;
            n_sym += n;
            stored_left -= n;
        }
        return true;
    }

    // one Huffman-coded block into symbols; the same loop as GzInflater::huffman_loop on 16-bit units
    bool huffman_block()
    {
        const uint32_t *const lt = lit_table, *const dt = dist_table;
        constexpr uint64_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;
        for (;;) {
            if (!room()) return false;
            uint16_t *const first = sym.data() + kWindow;
            uint16_t *o = first + n_sym;
            uint16_t *const oend = sym.data() + sym.size() - 600;
            const unsigned char *ip = in;
            uint64_t bb = bitbuf;
            int bc = bitcnt;
            const char *problem = nullptr;
            bool block_done = false;

#define SK_REFILL()              \
    do {                         \
        bb |= load64(ip) << bc;  \
        ip += (63 - bc) >> 3;    \
        bc |= 56;                \
    } while (0)
#define SK_DROP(e)             \
    do {                       \
        bb >>= e_nbits(e);     \
        bc -= (int)e_nbits(e); \
    } while (0)
#define SK_PUT_LITERALS(e)                        \
    do {                                          \
        o[0] = (uint16_t)(e_value(e) & 0xff);     \
        o[1] = (uint16_t)(e_value(e) >> 8);       \
        o += 1 + e_kind(e);                       \
    } while (0)

            while (o < oend) {
                const bool fast = in_end - ip >= 16;
                if (fast) {
                    SK_REFILL();
                } else {
                    while (bc <= 56 && ip < in_end) {
                        bb |= (uint64_t)*ip++ << bc;
                        bc += 8;
                    }
                }
                uint32_t e = lt[bb & lmask];
                if (fast && __builtin_expect(e_kind(e) <= LIT2, 1)) {
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
                    SK_REFILL();
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
                // a distance never reaches before the window part of `sym` (it is at most 32768)
                const uint16_t *src = o - dist;
                uint16_t *const stop_at = o + len;
                if (dist >= 4) {
                    do {
                        memcpy(o, src, 8);
                        o += 4;
                        src += 4;
                    } while (o < stop_at);
                } else {
                    do *o++ = *src++;
                    while (o < stop_at);
                }
                o = stop_at;
            }
#undef SK_REFILL
#undef SK_DROP
#undef SK_PUT_LITERALS
            in = ip;
            bitbuf = bb;
            bitcnt = bc < 0 ? 0 : bc;
            n_sym = (size_t)(o - first);
            if (problem) return fail(problem);
            if (block_done) return true;
        }
    }

    const unsigned char *base;
    size_t total;
    bool header_ready = false;
    int header_kind = 0;
    uint64_t boundary_bit = 0;
};

GzParallel::GzParallel(const unsigned char *d, size_t n, size_t chunk_bytes, int max_stretches)
    : data(d), size(n), chunk(std::max<size_t>(chunk_bytes, 1024)), window(kWindow, 0)
{
    width = max_stretches > 0 ? max_stretches : std::max(1, WorkerPool::instance().size());
    for (int i = 0; i < width; ++i) stretches.emplace_back(new Stretch(data, size));
}

GzParallel::~GzParallel() {}

bool GzParallel::begin_member()
{
    bool none = false;
    size_t after = 0;
    Stretch &s = *stretches[0];
    if (!s.member_header((size_t)(start_bit / 8), first_member, &none, &after)) {
        err = s.problem();
        done = true;
        return false;
    }
    if (none) {
        done = true;
        return false;
    }
    first_member = false;
    at_member_start = false;
    start_bit = (uint64_t)after * 8;
    std::fill(window.begin(), window.end(), 0);
    crc_running = 0;
    len_running = 0;
    return true;
}

bool GzParallel::decode_round()
{
    round_out.clear();
    avail_at = 0;
    if (done) return false;
    if (at_member_start && !begin_member()) return false;
    ++rounds;
    const size_t byte0 = (size_t)(start_bit / 8);
    const size_t round_end = std::min(size, byte0 + (size_t)width * chunk);
    const uint64_t limit = round_end >= size ? UINT64_MAX : (uint64_t)round_end * 8;
    const int n = (int)std::min<size_t>((size_t)width, (round_end - byte0 + chunk - 1) / chunk);
    WorkerPool &pool = WorkerPool::instance();

    // A: every stretch but the first looks for a block header in its chunk
    std::vector<char> valid((size_t)n, 0);
    valid[0] = 1;
    stretches[0]->start_known(start_bit, window.data());
    for (int k = 0; k < n; ++k) stretches[(size_t)k]->soft_cap = std::max<size_t>(chunk * 24, 16u << 20);
    if (n > 1)
        pool.parallel_for((size_t)n - 1, (size_t)n - 1, [&](size_t lo, size_t hi, size_t) {
            for (size_t k = lo + 1; k < hi + 1; ++k) {
                Stretch &s = *stretches[k];
                const uint64_t from = (uint64_t)(byte0 + k * chunk) * 8;
                const uint64_t to = (uint64_t)std::min(size, byte0 + (k + 1) * chunk) * 8;
                s.cap_symbols = chunk * 40 + (1u << 20); // beyond ~40:1 a guess is taken to be decoding nonsense
                valid[k] = s.find_start(from, to) ? 1 : 0;
            }
        });
    // B: all decode until they arrive on a later stretch's start (or the end of the round)
    pool.parallel_for((size_t)n, (size_t)n, [&](size_t lo, size_t hi, size_t) {
        for (size_t k = lo; k < hi; ++k) {
            if (!valid[k]) continue;
            std::vector<uint64_t> stops;
            for (size_t j = k + 1; j < (size_t)n; ++j)
                if (valid[j]) stops.push_back(stretches[j]->begin_bit);
            std::sort(stops.begin(), stops.end());
            stretches[k]->run(stops, limit);
        }
    });
    // the chain of stretches that start where the one before ended, from the known one
    std::vector<int> chain{0};
    for (;;) {
        const Stretch &c = *stretches[(size_t)chain.back()];
        if (c.end != Stretch::AT_STOP) break;
        int next = -1;
        for (int j = chain.back() + 1; j < n; ++j)
            if (valid[(size_t)j] && stretches[(size_t)j]->begin_bit == c.end_bit) {
                next = j;
                break;
            }
        if (next < 0) break; // cannot happen: a stop is some later stretch's start
        chain.push_back(next);
    }
    stretches_used += chain.size();
    for (int k = 0; k < n; ++k)
        if (valid[(size_t)k] && std::find(chain.begin(), chain.end(), k) == chain.end()) ++stretches_dropped;

    // windows: the 32 KiB before each chain element, resolved one after the other (only the tails)
    std::vector<std::vector<unsigned char>> win(chain.size() + 1);
    win[0] = window;
    std::vector<size_t> offset(chain.size() + 1, 0);
    for (size_t i = 0; i < chain.size(); ++i) {
        const Stretch &c = *stretches[(size_t)chain[i]];
        offset[i + 1] = offset[i] + c.n_sym;
        const size_t t = std::min(c.n_sym, kWindow);
        std::vector<unsigned char> &w = win[i + 1];
        w.resize(kWindow);
        memcpy(w.data(), win[i].data() + t, kWindow - t);
        const uint16_t *s = c.sym.data() + kWindow + c.n_sym - t;
        for (size_t j = 0; j < t; ++j) w[kWindow - t + j] = s[j] < kUnknown ? (unsigned char)s[j] : win[i][s[j] - kUnknown];
    }
    const size_t total = offset[chain.size()];
    round_out.resize(total);
    // symbols -> bytes, all stretches at once in 1 Mi-symbol pieces
    struct Piece {
        size_t el, from, to;
    };
    std::vector<Piece> pieces;
    for (size_t i = 0; i < chain.size(); ++i)
        for (size_t at = 0; at < stretches[(size_t)chain[i]]->n_sym; at += 1u << 20)
            pieces.push_back({i, at, std::min(at + (1u << 20), stretches[(size_t)chain[i]]->n_sym)});
    if (!pieces.empty())
        pool.parallel_for(pieces.size(), pieces.size(), [&](size_t lo, size_t hi, size_t) {
            for (size_t q = lo; q < hi; ++q) {
                const Piece &pc = pieces[q];
                const uint16_t *s = stretches[(size_t)chain[pc.el]]->sym.data() + kWindow;
                const unsigned char *w = win[pc.el].data();
                unsigned char *o = (unsigned char *)round_out.data() + offset[pc.el];
                for (size_t j = pc.from; j < pc.to; ++j) o[j] = s[j] < kUnknown ? (unsigned char)s[j] : w[s[j] - kUnknown];
            }
        });
    window = win[chain.size()];

    // CRC-32 and length of every member that ended in this round; the open member's running values
    // go on to the next round.  Segments = the stretches of output between member ends.
    {
        struct Seg {
            size_t from, to;
            bool closes; // a member ends at `to`
            uint32_t crc, isize, got;
        };
        std::vector<Seg> segs;
        size_t from = 0;
        for (size_t i = 0; i < chain.size(); ++i)
            for (const Stretch::Mark &m : stretches[(size_t)chain[i]]->marks) {
                const size_t to = offset[i] + m.n_sym;
                segs.push_back({from, to, true, m.crc, m.isize, 0});
                from = to;
            }
        if (from < total) segs.push_back({from, total, false, 0, 0, 0});
        const unsigned char *bytes = (const unsigned char *)round_out.data();
        if (segs.size() <= 2) { // long members: each segment on all threads
            for (Seg &g : segs) g.got = deflate_parallel_crc32(bytes + g.from, g.to - g.from);
        } else { // many short members: the segments side by side
            pool.parallel_for(segs.size(), std::min(segs.size(), (size_t)pool.size() * 4), [&](size_t lo, size_t hi, size_t) {
                for (size_t q = lo; q < hi; ++q) segs[q].got = deflate_crc32(bytes + segs[q].from, segs[q].to - segs[q].from);
            });
        }
        for (const Seg &g : segs) {
            const size_t n = g.to - g.from;
            crc_running = (uint32_t)crc32_combine(crc_running, g.got, (z_off_t)n);
            len_running += n;
            if (g.closes) {
                if (!err && crc_running != g.crc) err = "incorrect data check";
                else if (!err && (uint32_t)len_running != g.isize) err = "incorrect length check";
                crc_running = 0;
                len_running = 0;
            }
        }
        if (err) done = true;
    }
    const Stretch &tail = *stretches[(size_t)chain.back()];
    switch (tail.end) {
    case Stretch::AT_LIMIT:
        start_bit = tail.end_bit;
        break;
    case Stretch::STREAM_END: // the last member is complete and nothing that is a gzip member follows
        done = true;
        break;
    default: // FAILED on the chain: the data is damaged there (what was decoded before it is kept)
        if (!err) err = tail.problem() ? tail.problem() : "invalid deflate data";
        done = true;
        break;
    }
    return total > 0;
}

size_t GzParallel::read(char *dst, size_t want)
{
    size_t produced = 0;
    while (produced < want) {
        if (avail_at < round_out.size()) {
            const size_t n = std::min(round_out.size() - avail_at, want - produced);
            if (n >= (8u << 20)) { // one core copies ~10 GB/s: a round is tens of MiB
                const char *from = round_out.data() + avail_at;
                char *to = dst + produced;
                WorkerPool &pool = WorkerPool::instance();
                pool.parallel_for(n, std::min<size_t>((size_t)pool.size(), n >> 20), [&](size_t lo, size_t hi, size_t) { memcpy(to + lo, from + lo, hi - lo); });
            } else {
                memcpy(dst + produced, round_out.data() + avail_at, n);
            }
            avail_at += n;
            produced += n;
            continue;
        }
        if (done) break;
        decode_round();
    }
    return produced;
}
