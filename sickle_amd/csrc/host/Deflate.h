// Deflate.h -- what the gzip decoders of the ingest share: the bit reader, the Huffman decode
// tables of a deflate block (RFC 1951) and the block header parser.  GzInflater decodes one stream
// serially into bytes; GzParallel decodes several stretches of one stream at once.
#ifndef SICKLE_DEFLATE_H
#define SICKLE_DEFLATE_H

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace deflate_detail {
// table entry: bits 0-5 code bits to drop (the shift count as the CPU takes it), 8-12 extra bits
// (subtable index bits for SUB, length of the first code for LIT2), 13-15 kind, 16-31 literal /
// two literals / base value / subtable offset
enum Kind : uint32_t { LIT = 0, LIT2 = 1, BASE = 2, EOB = 3, SUB = 4, INVALID = 5 };
inline uint32_t entry(uint32_t nbits, Kind k, uint32_t extra, uint32_t value) { return nbits | (extra << 8) | (k << 13) | (value << 16); }
inline uint32_t e_nbits(uint32_t e) { return e & 63; }
inline uint32_t e_kind(uint32_t e) { return (e >> 13) & 7; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint64_t load64(const unsigned char *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v; // little-endian host (x86-64)
}
inline void store64(unsigned char *p, uint64_t v) { memcpy(p, &v, 8); }
inline void store16(unsigned char *p, uint16_t v) { memcpy(p, &v, 2); }

} // namespace deflate_detail

uint32_t deflate_crc32(const unsigned char *p, size_t n);          // on the calling thread
uint32_t deflate_parallel_crc32(const unsigned char *p, size_t n); // large buffers: on all host threads

// what GZReader pulls decoded bytes from
class GzSource {
public:
    virtual ~GzSource() {}
    // up to `want` decoded bytes into dst; fewer only at the end of the input or on an error
    virtual size_t read(char *dst, size_t want) = 0;
    virtual bool finished() const = 0;
    virtual const char *error() const = 0;
};

class DeflateStream {
public:
    static constexpr int kLitBits = 11, kDistBits = 8;
    const char *error() const { return err; }

protected:
    DeflateStream(const unsigned char *data, size_t size) : in(data), in_end(data + size) {}
    bool fail(const char *what)
    {
        if (!err) err = what;
        return false;
    }
    bool need_bits(int n);
    uint32_t take_bits(int n);
    void align_to_byte();
    // skips a gzip member header at `in`; *none = no (further) member there
    bool gzip_header(bool first_member, bool *none);
    // reads the next block header: 0 = stored (stored_left set), 1 = Huffman (tables built), -1 = error
    int next_block();
    bool build(const uint8_t *lens, int n, uint32_t *table, int primary_bits, bool dist);
    void fixed_tables();
    bool dynamic_tables();

    const unsigned char *in, *in_end;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    const char *err = nullptr;
    bool last_block = false;
    size_t stored_left = 0;
    uint32_t lit_table[(1 << kLitBits) + 288 * 16];
    uint32_t dist_table[(1 << kDistBits) + 32 * 128];
    bool tables_are_fixed = false;
    // of the tables built last: does every bit string decode (Kraft sum exactly 1)?  A compressor
    // writes complete codes (or a single distance code); random bits almost never form them.
    bool lit_complete = false, dist_complete = false;
    int dist_codes = 0;
};

#endif
