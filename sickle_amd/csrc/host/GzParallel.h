// GzParallel.h -- one gzip stream decoded on all host threads.
//
// A deflate stream is serial: a block can only be found by decoding the one before it, and a
// match may copy from the 32 KiB before its block.  The reference (gzgets, src/GZReader.cpp:59-92)
// and any zlib reader therefore run at one core's decode rate.  This decoder cuts the compressed
// file into chunks and gives each to a thread, which
//   1. searches its chunk bit by bit for something that parses as the header of a dynamic block
//      with complete Huffman codes (real headers do, random bits practically never), and
//   2. decodes from there into 16-bit symbols: a literal byte, or "byte j of the 32 KiB before my
//      start" for what a match copied from the window it does not have,
// until it arrives exactly on the block start another thread began at (or the round's end).  The
// first stretch of a round starts at a known block with its window known.  Correctness does not
// rest on the guess: stretches are joined only where one ENDS exactly on the bit another STARTED
// at, beginning with the known one, so every stretch used began on a true block boundary; a wrong
// guess is never arrived at and its work is dropped (the neighbour simply decodes on through
// it).  The placeholders are then filled in from the resolved tail of the preceding stretch, and
// CRC-32 and length are checked per member as in GzInflater.  (The approach of Kerbiriou & Chikhi,
// "Parallel decompression of gzip-compressed files and random access to DNA sequences", 2019.)
#ifndef SICKLE_GZPARALLEL_H
#define SICKLE_GZPARALLEL_H

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "Deflate.h"

class GzParallel : public GzSource {
public:
    // chunk_bytes: compressed bytes per stretch (tests shrink it to force many rounds)
    GzParallel(const unsigned char *data, size_t size, size_t chunk_bytes = 2u << 20, int max_stretches = 0);
    ~GzParallel();
    size_t read(char *dst, size_t want) override;
    bool finished() const override { return done && avail_at == round_out.size(); }
    const char *error() const override { return err; }
    // diagnostics
    uint64_t stretches_used = 0, stretches_dropped = 0, rounds = 0;

private:
    class Stretch;
    bool decode_round();
    bool begin_member();
    void check_member_end(size_t out_end);

    const unsigned char *data;
    size_t size;
    size_t chunk;
    int width;
    std::vector<std::unique_ptr<Stretch>> stretches;
    uint64_t start_bit = 0;       // a block header known to start here
    bool at_member_start = true;  // the next thing to read is a gzip member header at byte start_bit / 8
    bool first_member = true;
    bool done = false;
    const char *err = nullptr;
    std::vector<unsigned char> window; // the 32 KiB before start_bit, resolved
    std::vector<char> round_out;
    size_t avail_at = 0;
    // CRC-32 / length of the current member up to the end of the previous round
    uint32_t crc_running = 0;
    uint64_t len_running = 0;
    size_t member_from = 0; // offset in round_out where the current member's bytes of this round begin
};

#endif
