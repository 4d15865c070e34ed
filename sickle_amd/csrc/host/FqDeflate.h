// FqDeflate.h -- deflate encoder for the -g output, shaped for FASTQ text, and the BGZF block
// writer around it.
//
// The fast setting of the -g writer (SICKLE_GZ_LEVEL=fast; the default is zlib at its default level,
// like the reference's gzopen(path, "w")).  It looks only for what is cheap to find in FASTQ text --
// a copy from the same column of the line four lines up (the previous record's header, its '+'
// line; one or two columns to the side once a numeric field changed width) and runs of one byte --
// codes everything else as literals, and builds the block's two Huffman codes from the resulting
// counts (dynamic block, RFC 1951 3.2.7).  About 10x the speed of zlib level 6 for files about a
// sixth larger on real Illumina reads.  Output is plain deflate: any inflater
// reads it.  One call = one block of at most 65 280 input bytes.
#ifndef SICKLE_FQDEFLATE_H
#define SICKLE_FQDEFLATE_H

#include <cstddef>
#include <string>

// deflate stream (one final block, or stored blocks if the text does not compress) for text[0, n);
// returns its size, 0 if cap is too small (cap >= n + 16 is always enough)
size_t fq_deflate(const char *text, size_t n, unsigned char *out, size_t cap);

constexpr size_t kBgzfInput = 0xff00; // payload bytes per BGZF block, as bgzip
extern const unsigned char kBgzfEofBlock[28];
// appends the BGZF block (a gzip member with the "BC" size field, SAM spec 4.1) holding [p, p+n),
// n <= kBgzfInput.  level < 0: fq_deflate; 0..9: zlib at that level.
void bgzf_append_block(const char *p, size_t n, int level, std::string &out);

#endif
