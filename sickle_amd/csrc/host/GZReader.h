// GZReader.h -- FASTQ ingest: (optionally gzipped) file -> batches of lines.
//
// Role of reference src/GZReader.{h,cpp} and src/Batch.{h,cpp}; written fresh.  The reference
// reads line by line with gzgets into one heap string per line; this reader pulls large blocks
// (plain regular files are mapped and indexed in place, pipes read(2); gzip decoded from the mapped file on all host threads (GzParallel), or
// through zlib's gzread when the input is not a regular file; and for BGZF -- blocked gzip as written by bgzip and by this
// program's own -g -- all blocks of a chunk inflated at once on the host threads)
// straight into the batch's own text buffer, finds
// the newlines of a block with all host threads at once, and indexes the lines in place.  What
// it keeps is the reference's BATCH CUT RULE, because that rule is observable:
//   - a batch ends after the line that drives the byte budget to <= 0 (src/GZReader.cpp:61-92),
//     the budget being batch_len minus the carried lines;
//   - the trailing (lines mod 4) lines -- mod 8 for interleaved input -- are carried into the
//     next batch (src/GZReader.cpp:104-129);
//   - a batch left with no lines ends the whole run, and lines still carried at end of file
//     are dropped (src/GZReader.cpp:29-41);
//   - every stored line is its gzgets piece minus its LAST character (src/GZReader.cpp:81-88):
//     the newline normally, a real character for a final line without one or for a piece cut
//     at batch_len-1 characters.
// The order records are written in (-a T deals them into T queues per batch), the two-file
// "different lengths" abort and the paired summary's "Total input" line all depend on it.
#ifndef SICKLE_GZREADER_H
#define SICKLE_GZREADER_H

#include <sys/mman.h>
#include <zlib.h>

#include "Deflate.h"

#include <cstdint>
#include <cstdlib>
#include <string_view>
#include <vector>

// growable byte buffer without value-initialisation (a std::vector<char> would zero every block
// before read() overwrites it); or a borrowed view of bytes someone else owns (a mapped file)
class RawBuf {
public:
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf()
    {
        if (!borrowed) free(p);
        else drop_view_pages();
    }
    // A consumed view of the mapped input hands its page-table entries back now (the pages stay in the page
    // cache; a neighbouring view that shares a boundary page simply faults it in again).  Otherwise the whole
    // multi-gigabyte mapping is torn down by the kernel at process exit, serially, while the caller waits:
    // measured on one box, interleaved, 20 M reads: 1.36-1.44 s without, 1.13-1.16 s with (tools/probes/extern_time.sh).
    void drop_view_pages()
    {
        if (n < (1u << 20)) return;
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + 4095) & ~(uintptr_t)4095;
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(p) + n) & ~(uintptr_t)4095;
        if (hi > lo) madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_DONTNEED);
    }
    char *data() { return p; }
    const char *data() const { return p; }
    size_t size() const { return n; }
    bool is_borrowed() const { return borrowed; }
    void borrow(const char *from, size_t bytes) // a view; the owner outlives it.  Never written through.
    {
        if (!borrowed) free(p);
        p = const_cast<char *>(from);
        n = bytes;
        cap = (size_t)-1;
        borrowed = true;
    }
    void reserve(size_t c)
    {
        if (c <= cap) return;
        size_t nc = cap ? cap : (1u << 20);
        while (nc < c) nc += nc >> 1;
        p = (char *)realloc(p, nc);
        if (!p) abort();
        cap = nc;
    }
    void set_size(size_t s) { n = s; }
    void swap(RawBuf &o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
        std::swap(borrowed, o.borrowed);
    }

private:
    char *p = nullptr;
    size_t n = 0, cap = 0;
    bool borrowed = false;
};

// One batch: the text of its lines and where each line sits.  Role of reference src/Batch.h.
class Batch {
public:
    bool has_lines() const { return cursor < line_len.size(); }
    std::string_view next_line()
    {
        std::string_view v(text.data() + line_off[cursor], line_len[cursor]);
        ++cursor;
        return v;
    }
    std::string_view line(size_t i) const { return std::string_view(text.data() + line_off[i], line_len[i]); }
    int n_lines() const { return (int)line_len.size(); }
    size_t cursor_pos() const { return cursor; }
    void skip_lines(size_t k) { cursor += k; }
    long sequences_len = 0; // sum of the line lengths, as reference src/Batch.cpp:15
    void free_this() {}     // storage is owned by the object; kept for source compatibility

private:
    friend class GZReader;
    RawBuf text;
    std::vector<uint64_t> line_off;
    std::vector<uint32_t> line_len;
    size_t cursor = 0;
};

class GZReader {
public:
    GZReader(const char *path, int batch_len, bool interleaved = false);
    ~GZReader();
    bool is_open() const { return file != nullptr || fd >= 0; }
    // the next batch, or NULL when the run is over (see the cut rule above); caller deletes
    Batch *get_batch_buffering_lines();
    bool reached_end() const { return eof; }
    const char *path;

private:
    bool fill();        // read another block behind `pending`
    size_t fill_bgzf(size_t old); // BGZF input: inflate the next run of blocks behind pending[old)
    void stream_from(uint64_t offset); // the rest of a gzip file as one serial stream
    void index_more();  // find the newlines of the bytes not yet indexed

    gzFile file = nullptr; // gzip input read through zlib (pipes, SICKLE_ZLIB_INFLATE=1)
    GzSource *fast = nullptr;   // gzip input decoded from the mapped file (GzParallel, or GzInflater for small files)
    const unsigned char *map = nullptr;
    size_t map_len = 0;
    int fd = -1;           // plain input: read(2) / parallel pread(2), no zlib copy
    bool regular = false;  // a regular file of known size: blocks are pread in parallel slices
    bool mapped = false;   // ... or, by default, the file is mapped and batches are views of the mapping
    bool bgzf = false;     // gzip input whose members carry their size (BGZF): inflated in parallel
    RawBuf cbuf;           // compressed bytes of the current BGZF chunk
    uint64_t file_size = 0, file_pos = 0;
    bool eof = false;      // gzgets would have returned NULL: no further batch
    bool in_eof = false;   // the underlying stream is exhausted
    int batch_len;
    int min_lines_in_batch;
    // Bytes read from the file and not yet handed out in a batch.  [0, indexed) is cut into
    // lines (idx_off / idx_bytes, bytes INCLUDING the terminating character that is not stored);
    // the first n_carry of them were carried over from the previous batch.
    RawBuf pending;
    size_t indexed = 0;
    std::vector<uint64_t> idx_off;
    std::vector<uint32_t> idx_bytes;
    size_t n_carry = 0;
};

#endif
