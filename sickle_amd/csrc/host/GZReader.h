// GZReader.h -- FASTQ ingest: (optionally gzipped) file -> batches of lines.
//
// Role of reference src/GZReader.{h,cpp} and src/Batch.{h,cpp}; written fresh.  The reference
// reads line by line with gzgets into one heap string per line; this reader pulls megabyte
// blocks with gzread straight into the batch's own text buffer and indexes the lines in
// place.  What it keeps is the reference's BATCH CUT RULE, because that rule is observable:
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

#include <zlib.h>

#include <cstdint>
#include <string_view>
#include <vector>

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
    long sequences_len = 0; // sum of the line lengths, as reference src/Batch.cpp:15
    void free_this() {}     // storage is owned by the object; kept for source compatibility

private:
    friend class GZReader;
    std::vector<char> text;
    std::vector<uint64_t> line_off;
    std::vector<uint32_t> line_len;
    size_t cursor = 0;
};

class GZReader {
public:
    GZReader(const char *path, int batch_len, bool interleaved = false);
    ~GZReader();
    bool is_open() const { return file != nullptr; }
    // the next batch, or NULL when the run is over (see the cut rule above); caller deletes
    Batch *get_batch_buffering_lines();
    bool reached_end() const { return eof; }
    const char *path;

private:
    bool fill();                            // gzread another block behind `pending`
    bool next_piece(size_t *start, size_t *len); // one gzgets-equivalent piece inside `pending`

    gzFile file = nullptr;
    bool eof = false;    // gzgets would have returned NULL: no further batch
    bool in_eof = false; // the underlying stream is exhausted
    int batch_len;
    int min_lines_in_batch;
    // bytes read from the file and not yet handed out in a batch: first the carried lines
    // (already indexed in carry_off/len, relative to pending), then unparsed bytes from `scan`
    std::vector<char> pending;
    size_t scan = 0;
    std::vector<uint64_t> carry_off;
    std::vector<uint32_t> carry_len;
};

#endif
