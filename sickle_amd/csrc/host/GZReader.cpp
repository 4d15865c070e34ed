#include "GZReader.h"

#include <cstdio>
#include <cstring>
#include <iostream>

#include "sickle.h"

namespace {
constexpr size_t kBlock = 8u << 20; // bytes per gzread
}

GZReader::GZReader(const char *path_, int batch_len_, bool interleaved) : path(path_), batch_len(batch_len_)
{
    min_lines_in_batch = interleaved ? 8 : 4; // reference src/GZReader.cpp:7-11
    file = gzopen(path, "r");
    if (!file) {
        fprintf(stderr, "****Error: Could not open input file '%s'.\n\n", path); // src/GZReader.cpp:15
        eof = true;
        return;
    }
    gzbuffer(file, 1u << 20);
}

GZReader::~GZReader()
{
    if (file) gzclose(file);
}

bool GZReader::fill()
{
    if (in_eof || !file) return false;
    const size_t old = pending.size();
    pending.resize(old + kBlock);
    int got = gzread(file, pending.data() + old, (unsigned)kBlock);
    if (got < 0) got = 0;
    pending.resize(old + (size_t)got);
    if ((size_t)got < kBlock) in_eof = true;
    return got > 0;
}

// What one gzgets(file, buf, batch_len) call would return: up to batch_len-1 characters, ending
// after the first newline.  False at end of input.
bool GZReader::next_piece(size_t *start, size_t *len)
{
    const size_t limit = (size_t)(batch_len > 1 ? batch_len - 1 : 1);
    for (;;) {
        const size_t avail = pending.size() - scan;
        const size_t look = avail < limit ? avail : limit;
        const char *base = pending.data() + scan;
        const char *nl = look ? (const char *)memchr(base, '\n', look) : nullptr;
        if (nl) {
            *start = scan;
            *len = (size_t)(nl - base) + 1;
            scan += *len;
            return true;
        }
        if (look == limit) { // a piece cut by the buffer size, no newline in it
            *start = scan;
            *len = limit;
            scan += limit;
            return true;
        }
        if (!fill()) {
            if (avail == 0) return false;
            *start = scan; // the last line of a file that does not end in a newline
            *len = avail;
            scan += avail;
            return true;
        }
    }
}

Batch *GZReader::get_batch_buffering_lines()
{
    if (eof) return nullptr; // src/GZReader.cpp:31
    Batch *batch = new Batch();
    std::vector<uint64_t> &off = batch->line_off;
    std::vector<uint32_t> &len = batch->line_len;
    long remaining = batch_len; // src/GZReader.cpp:61
    off = carry_off;
    len = carry_len;
    for (uint32_t l : len) remaining -= l; // src/GZReader.cpp:68-75
    carry_off.clear();
    carry_len.clear();
    do {
        size_t start, plen;
        if (!next_piece(&start, &plen)) { // src/GZReader.cpp:77-80
            eof = true;
            break;
        }
        const size_t stored = plen - 1; // the piece minus its last character, src/GZReader.cpp:81-88
        remaining -= (long)stored;
        off.push_back(start);
        len.push_back((uint32_t)stored);
    } while (remaining > 0);

    // src/GZReader.cpp:104-129: the trailing lines that do not complete a record are carried
    const size_t extra = len.size() % (size_t)min_lines_in_batch;
    const size_t keep = len.size() - extra;
    // everything from the first carried line (or from `scan`) on stays in `pending`
    const size_t cut = extra ? (size_t)off[keep] : scan;
    std::vector<char> rest(pending.begin() + (long)cut, pending.end());
    for (size_t i = keep; i < len.size(); ++i) {
        carry_off.push_back(off[i] - cut);
        carry_len.push_back(len[i]);
    }
    off.resize(keep);
    len.resize(keep);
    pending.resize(cut);
    batch->text.swap(pending);
    pending.swap(rest);
    scan -= cut;
    for (uint32_t l : len) batch->sequences_len += l;
    if (keep == 0) { // src/GZReader.cpp:33-40
        delete batch;
        return nullptr;
    }
    return batch;
}
