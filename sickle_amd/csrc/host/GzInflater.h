// GzInflater.h -- gzip (RFC 1952 / RFC 1951) decoder for the ingest of gzipped FASTQ.
//
// The reference reads gzip input line by line through zlib's gzgets (src/GZReader.cpp:59-92); a
// plain gzip file is one serial deflate stream, so its decode rate bounds the whole run.  This
// decoder works on the compressed file mapped into memory: 64-bit bit buffer refilled without a
// branch, one table lookup per literal/length and per distance symbol (11- and 8-bit primary
// tables with subtables for longer codes), word-wise match copies.  The CRC-32 of each member is
// checked after the fact on all host threads (WorkerPool), not in the decode loop.  Concatenated
// members are decoded one after the other and bytes after the last member are ignored, as zlib's
// gz reader does.  Damaged data ends the input at that point with error() set.
#ifndef SICKLE_GZINFLATER_H
#define SICKLE_GZINFLATER_H

#include <cstddef>
#include <cstdint>
#include <vector>

#include "Deflate.h"

class GzInflater : public GzSource, public DeflateStream {
public:
    GzInflater(const unsigned char *data, size_t size);
    // up to `want` decoded bytes into dst; fewer only at the end of the input or on an error
    size_t read(char *dst, size_t want) override;
    bool finished() const override { return state == DONE || state == FAILED; }
    const char *error() const override { return DeflateStream::error(); }

private:
    enum State { MEMBER_HEADER, BLOCK_HEADER, STORED, HUFFMAN, MEMBER_TRAILER, DONE, FAILED };
    struct MemberEnd {
        uint64_t abs_off; // decoded bytes of the whole input up to the end of this member
        uint32_t crc, isize;
    };
    static constexpr size_t kWindow = 32768, kChunk = 256 * 1024, kSlack = 512;

    bool member_header();
    bool block_header();
    bool run_huffman(size_t olimit);
    template <bool FAST> bool huffman_loop(size_t olimit, bool &block_done);
    bool run_stored(size_t olimit);
    bool member_trailer();
    bool verify(const char *dst, size_t produced);

    State state = MEMBER_HEADER;
    bool first_member = true;

    std::vector<unsigned char> win; // [history | bytes being produced | slack]
    size_t opos = 0;                // decoded up to here
    size_t rpos = 0;                // handed to the caller up to here
    size_t mstart = 0;              // where the current member began (matches may not reach before it)
    uint64_t abs_handed = 0;        // decoded bytes handed to the caller so far

    // CRC and length of the part of the current member handed out in EARLIER read() calls
    uint32_t crc_running = 0;
    uint64_t len_running = 0;
    std::vector<MemberEnd> ends; // members whose trailer has been read and whose bytes are not all checked yet

};

#endif
