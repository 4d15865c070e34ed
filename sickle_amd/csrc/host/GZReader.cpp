#include "GZReader.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "WorkerPool.h"
#include "sickle.h"

namespace {
constexpr size_t kBlock = 32u << 20; // bytes per read
}

GZReader::GZReader(const char *path_, int batch_len_, bool interleaved) : path(path_), batch_len(batch_len_)
{
    min_lines_in_batch = interleaved ? 8 : 4; // reference src/GZReader.cpp:7-11
    fd = open(path, O_RDONLY);
    if (fd < 0) {
        fprintf(stderr, "****Error: Could not open input file '%s'.\n\n", path); // src/GZReader.cpp:15
        eof = true;
        return;
    }
    unsigned char magic[2] = {0, 0};
    const ssize_t got = pread(fd, magic, 2, 0);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) { // gzip: hand the descriptor to zlib
        file = gzdopen(fd, "r");
        fd = -1;
        if (!file) {
            fprintf(stderr, "****Error: Could not open input file '%s'.\n\n", path);
            eof = true;
            return;
        }
        gzbuffer(file, 4u << 20);
    } else {
        posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            regular = true;
            file_size = (uint64_t)st.st_size;
        }
    }
}

GZReader::~GZReader()
{
    if (file) gzclose(file);
    if (fd >= 0) close(fd);
}

bool GZReader::fill()
{
    if (in_eof) return false;
    const size_t old = pending.size();
    pending.reserve(old + kBlock);
    size_t got = 0;
    if (file) {
        const int r = gzread(file, pending.data() + old, (unsigned)kBlock);
        got = r > 0 ? (size_t)r : 0;
        if (got < kBlock) in_eof = true;
    } else if (regular) {
        // one block = several slices pread concurrently on the host pool (a single read(2) stream
        // is a page-cache memcpy on one core, ~3 GB/s; slices scale with the cores)
        const uint64_t want = std::min<uint64_t>(kBlock, file_size - file_pos);
        if (want == 0) {
            in_eof = true;
        } else {
            char *dst = pending.data() + old;
            const uint64_t base = file_pos;
            const size_t slices = (size_t)std::min<uint64_t>(8, (want + (4u << 20) - 1) / (4u << 20));
            std::atomic<bool> failed{false};
            WorkerPool::instance().parallel_for((size_t)want, slices, [&](size_t b, size_t e, size_t) {
                size_t done = b;
                while (done < e) {
                    const ssize_t r = pread(fd, dst + done, e - done, (off_t)(base + done));
                    if (r <= 0) { // the file shrank under us, or an I/O error
                        failed = true;
                        return;
                    }
                    done += (size_t)r;
                }
            });
            if (failed) {
                fprintf(stderr, "****Error: could not read input file '%s'.\n\n", path);
                exit(EXIT_FAILURE);
            }
            got = (size_t)want;
            file_pos += want;
            if (file_pos >= file_size) in_eof = true;
        }
    } else {
        while (got < kBlock) { // a pipe or device: read(2) may return short counts
            const ssize_t r = read(fd, pending.data() + old + got, kBlock - got);
            if (r <= 0) {
                in_eof = true;
                break;
            }
            got += (size_t)r;
        }
    }
    pending.set_size(old + got);
    return got > 0;
}

// Cuts [indexed, pending.size()) into lines: every newline ends one; at end of input the
// unterminated tail is a line too.  The newline search runs on all host threads.
void GZReader::index_more()
{
    const char *base = pending.data();
    const size_t begin = indexed, end = pending.size();
    if (begin >= end) return;
    WorkerPool &pool = WorkerPool::instance();
    const size_t span = end - begin;
    size_t parts = span / (2u << 20);
    if (parts < 1) parts = 1;
    if (parts > (size_t)pool.size() * 2) parts = (size_t)pool.size() * 2;
    std::vector<std::vector<uint64_t>> found(parts);
    pool.parallel_for(span, parts, [&](size_t b, size_t e, size_t part) {
        std::vector<uint64_t> &out = found[part];
        out.reserve((e - b) / 64 + 16);
        const char *p = base + begin + b, *stop = base + begin + e;
        while (p < stop) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(stop - p));
            if (!nl) break;
            out.push_back((uint64_t)(nl - base));
            p = nl + 1;
        }
    });
    // line table: line j of part p starts after the previous newline (the last newline of the
    // nearest earlier non-empty part, or `begin`); the parts fill their slices concurrently
    std::vector<size_t> first(parts + 1, 0);
    for (size_t p = 0; p < parts; ++p) first[p + 1] = first[p] + found[p].size();
    const size_t added = first[parts];
    const size_t old_lines = idx_off.size();
    idx_off.resize(old_lines + added);
    idx_bytes.resize(old_lines + added);
    std::vector<uint64_t> part_start(parts, begin);
    {
        uint64_t prev = begin;
        for (size_t p = 0; p < parts; ++p) {
            part_start[p] = prev;
            if (!found[p].empty()) prev = found[p].back() + 1;
        }
    }
    pool.parallel_for(parts, parts, [&](size_t lo, size_t hi, size_t) {
        for (size_t p = lo; p < hi; ++p) {
            uint64_t st = part_start[p];
            uint64_t *off = idx_off.data() + old_lines + first[p];
            uint32_t *len = idx_bytes.data() + old_lines + first[p];
            const std::vector<uint64_t> &v = found[p];
            for (size_t j = 0; j < v.size(); ++j) {
                off[j] = st;
                len[j] = (uint32_t)(v[j] + 1 - st);
                st = v[j] + 1;
            }
        }
    });
    size_t start = begin;
    for (size_t p = parts; p-- > 0;)
        if (!found[p].empty()) {
            start = (size_t)found[p].back() + 1;
            break;
        }
    if (in_eof && start < end) { // the last line of a file that does not end in a newline
        idx_off.push_back(start);
        idx_bytes.push_back((uint32_t)(end - start));
        start = end;
    }
    indexed = start;
}

Batch *GZReader::get_batch_buffering_lines()
{
    if (eof) return nullptr; // src/GZReader.cpp:31
    // gzgets(file, buf, batch_len) hands out at most batch_len-1 characters per call: a longer
    // line arrives in pieces, each stored minus its last character
    const uint32_t limit = (uint32_t)(batch_len > 1 ? batch_len - 1 : 1);
    long remaining = batch_len; // src/GZReader.cpp:61
    size_t k = 0;               // lines of idx_* taken into this batch
    for (; k < n_carry; ++k) remaining -= (long)(idx_bytes[k] - 1); // src/GZReader.cpp:68-75
    do {
        if (k == idx_off.size()) { // out of indexed lines: index what is buffered, else read on
            index_more();
            if (k == idx_off.size()) {
                const bool got = fill();
                index_more(); // at end of input this also takes an unterminated last line
                if (k == idx_off.size()) {
                    if (!got) { // src/GZReader.cpp:77-80: gzgets returned NULL
                        eof = true;
                        break;
                    }
                    continue; // a whole block without a newline: keep reading
                }
            }
        }
        if (idx_bytes[k] > limit) { // split off the first piece of an over-long line
            idx_off.insert(idx_off.begin() + (long)k + 1, idx_off[k] + limit);
            idx_bytes.insert(idx_bytes.begin() + (long)k + 1, idx_bytes[k] - limit);
            idx_bytes[k] = limit;
        }
        remaining -= (long)(idx_bytes[k] - 1); // the piece minus its last character, src/GZReader.cpp:81-88
        ++k;
    } while (remaining > 0);

    // src/GZReader.cpp:104-129: the trailing lines that do not complete a record are carried
    const size_t extra = k % (size_t)min_lines_in_batch;
    const size_t keep = k - extra;
    Batch *batch = new Batch();
    batch->line_off.assign(idx_off.begin(), idx_off.begin() + (long)keep);
    batch->line_len.resize(keep);
    {
        WorkerPool &pool = WorkerPool::instance();
        const size_t parts = (size_t)pool.size();
        std::vector<long> sums(parts, 0);
        pool.parallel_for(keep, parts, [&](size_t lo, size_t hi, size_t part) {
            long t = 0;
            for (size_t i = lo; i < hi; ++i) {
                batch->line_len[i] = idx_bytes[i] - 1;
                t += idx_bytes[i] - 1;
            }
            sums[part] = t;
        });
        long total = 0;
        for (long t : sums) total += t;
        batch->sequences_len = total;
    }

    // everything from the first carried line on stays in `pending` for the next batch
    const size_t cut = keep < idx_off.size() ? (size_t)idx_off[keep] : indexed;
    RawBuf rest;
    const size_t rest_bytes = pending.size() - cut;
    rest.reserve(rest_bytes + 1);
    memcpy(rest.data(), pending.data() + cut, rest_bytes);
    rest.set_size(rest_bytes);
    pending.set_size(cut);
    batch->text.swap(pending);
    pending.swap(rest);
    const size_t left = idx_off.size() - keep;
    for (size_t i = 0; i < left; ++i) {
        idx_off[i] = idx_off[keep + i] - cut;
        idx_bytes[i] = idx_bytes[keep + i];
    }
    idx_off.resize(left);
    idx_bytes.resize(left);
    indexed -= cut;
    n_carry = extra;
    if (keep == 0) { // src/GZReader.cpp:33-40
        delete batch;
        return nullptr;
    }
    return batch;
}
