#include "GZReader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "GzInflater.h"
#include "GzParallel.h"
#include "WorkerPool.h"
#include "sickle.h"

namespace {
constexpr size_t kBlock = 32u << 20; // bytes per read
constexpr size_t kBgzfLargest = 0x10000; // no BGZF block is longer
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
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        // gzip.  A BGZF file (first member carries a "BC" size field) is inflated block-parallel;
        // anything else -- and whatever follows the first block that is not BGZF or does not
        // check out -- goes through zlib's own gz reader, like the reference's gzgets.
        struct stat st;
        unsigned char head[18];
        const char *off = getenv("SICKLE_NO_BGZF"); // diagnostics: force the streaming reader
        if (!(off && *off && *off != '0') && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && pread(fd, head, 18, 0) == 18 && head[2] == 8 &&
            head[3] == 4 && head[10] == 6 && head[11] == 0 && head[12] == 'B' && head[13] == 'C') {
            bgzf = true;
            file_size = (uint64_t)st.st_size;
        } else {
            stream_from(0);
        }
    } else {
        posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            regular = true;
            file_size = (uint64_t)st.st_size;
            // map it: batches become views of the page cache instead of copies of it (the copy is
            // most of the system time of a run).  SICKLE_NO_MMAP=1 keeps to pread.
            const char *no = getenv("SICKLE_NO_MMAP");
            if (file_size > 0 && !(no && *no && *no != '0')) {
                void *m = mmap(nullptr, (size_t)file_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    madvise(m, (size_t)file_size, MADV_SEQUENTIAL);
                    map = (const unsigned char *)m;
                    map_len = (size_t)file_size;
                    mapped = true;
                    pending.borrow((const char *)map, 0);
                }
            }
        }
    }
}

GZReader::~GZReader()
{
    delete fast;
    if (map) munmap((void *)map, map_len);
    if (file) gzclose(file);
    if (fd >= 0) close(fd);
}

void GZReader::stream_from(uint64_t offset)
{
    bgzf = false;
    struct stat st;
    const char *z = getenv("SICKLE_ZLIB_INFLATE"); // diagnostics: decode with zlib instead
    if (!(z && *z && *z != '0') && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (uint64_t)st.st_size > offset) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            map = (const unsigned char *)m;
            map_len = (size_t)st.st_size;
            // small inputs are not worth cutting up; SICKLE_GZ_SERIAL=1 keeps to the serial decoder
            const char *ser = getenv("SICKLE_GZ_SERIAL");
            const size_t left = map_len - (size_t)offset;
            const char *ch = getenv("SICKLE_GZ_CHUNK"); // diagnostics/tests: compressed bytes per stretch
            if (ch && atol(ch) > 0) fast = new GzParallel(map + offset, left, (size_t)atol(ch));
            else if (left < (8u << 20) || (ser && *ser && *ser != '0')) fast = new GzInflater(map + offset, left);
            else fast = new GzParallel(map + offset, left);
            return;
        }
    }
    lseek(fd, (off_t)offset, SEEK_SET);
    file = gzdopen(fd, "r");
    fd = -1;
    if (!file) {
        fprintf(stderr, "****Error: Could not open input file '%s'.\n\n", path);
        eof = in_eof = true;
        return;
    }
    gzbuffer(file, 4u << 20);
}

namespace {
struct Inflater { // one raw-inflate state per worker thread, reset per block
    z_stream zs;
    bool live = false;
    ~Inflater()
    {
        if (live) inflateEnd(&zs);
    }
    bool prepare()
    {
        if (live) return inflateReset(&zs) == Z_OK;
        memset(&zs, 0, sizeof zs);
        live = inflateInit2(&zs, -15) == Z_OK;
        return live;
    }
};
} // namespace

// Loads the next chunk of compressed bytes, walks the block headers (each says how long its block
// is: SAM spec 4.1), and inflates the blocks side by side into pending[old...).  Returns the bytes
// produced.  A header that is not BGZF, a block cut off by the end of the file, or a block whose
// data, length or CRC does not check out ends the parallel part: everything before it is kept and
// the rest of the file is read by zlib's gzread, which then behaves (and fails) like the
// reference's reader on the same bytes.
size_t GZReader::fill_bgzf(size_t old)
{
    constexpr uint64_t kChunk = 24u << 20;
    const uint64_t want = std::min<uint64_t>(kChunk, file_size - file_pos);
    if (want == 0) {
        in_eof = true;
        return 0;
    }
    cbuf.reserve((size_t)want);
    unsigned char *c = (unsigned char *)cbuf.data();
    {
        const uint64_t base = file_pos;
        std::atomic<bool> failed{false};
        const size_t slices = (size_t)std::min<uint64_t>(8, (want + (4u << 20) - 1) / (4u << 20));
        WorkerPool::instance().parallel_for((size_t)want, slices, [&](size_t b, size_t e, size_t) {
            size_t done = b;
            while (done < e) {
                const ssize_t r = pread(fd, c + done, e - done, (off_t)(base + done));
                if (r <= 0) {
                    failed = true;
                    return;
                }
                done += (size_t)r;
            }
        });
        if (failed) {
            fprintf(stderr, "****Error: could not read input file '%s'.\n\n", path);
            fatal_exit(EXIT_FAILURE);
        }
    }
    struct Blk {
        size_t at, data, clen, out;
        uint32_t isize, crc;
    };
    std::vector<Blk> blks;
    size_t at = 0, out = 0;
    bool foreign = false; // stopped at something that is not a whole BGZF block
    while (at < want && out < kBlock) {
        const unsigned char *h = c + at;
        if (at + 18 > want) {
            foreign = true;
            break;
        }
        const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4 || at + 12 + xlen > want) {
            foreign = true;
            break;
        }
        size_t total = 0;
        for (size_t x = 0; x + 4 <= xlen;) {
            const unsigned char *f = h + 12 + x;
            const size_t flen = (size_t)f[2] | ((size_t)f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && flen == 2 && x + 6 <= xlen) total = ((size_t)f[4] | ((size_t)f[5] << 8)) + 1;
            x += 4 + flen;
        }
        if (total < 12 + xlen + 8 || at + total > want) {
            foreign = true;
            break;
        }
        const unsigned char *t = h + total - 8;
        const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (isize > 0x10000) {
            foreign = true;
            break;
        }
        blks.push_back({at, at + 12 + xlen, total - 12 - xlen - 8, out, isize, crc});
        out += isize;
        at += total;
    }
    // a chunk boundary in the middle of a block is not foreign: the next call starts at that block
    if (foreign && blks.empty() == false && at + kBgzfLargest > want && file_pos + want < file_size) foreign = false;
    pending.reserve(old + out);
    char *dst = pending.data() + old;
    std::atomic<size_t> bad{blks.size()};
    WorkerPool &pool = WorkerPool::instance();
    pool.parallel_for(blks.size(), std::min(blks.size(), (size_t)pool.size() * 4), [&](size_t lo, size_t hi, size_t) {
        static thread_local Inflater inf;
        for (size_t i = lo; i < hi; ++i) {
            const Blk &b = blks[i];
            bool ok = b.isize == 0 ? b.crc == 0 : inf.prepare(); // the empty end-of-file block
            if (ok && b.isize) {
                inf.zs.next_in = c + b.data;
                inf.zs.avail_in = (uInt)b.clen;
                inf.zs.next_out = (Bytef *)dst + b.out;
                inf.zs.avail_out = b.isize;
                ok = inflate(&inf.zs, Z_FINISH) == Z_STREAM_END && inf.zs.total_out == b.isize && inf.zs.avail_in == 0 &&
                     (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)dst + b.out, b.isize) == b.crc;
            }
            if (!ok) {
                size_t cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {
                }
                return;
            }
        }
    });
    const size_t nbad = bad.load();
    if (nbad < blks.size()) {
        stream_from(file_pos + blks[nbad].at);
        return blks[nbad].out;
    }
    if (foreign) {
        stream_from(file_pos + at);
        return out;
    }
    file_pos += at;
    if (file_pos >= file_size) in_eof = true;
    return out;
}

bool GZReader::fill()
{
    if (in_eof) return false;
    const size_t old = pending.size();
    size_t got = 0;
    while (bgzf && got == 0 && !in_eof) got = fill_bgzf(old);
    if (got > 0 || in_eof) {
        pending.set_size(old + got);
        return got > 0;
    }
    pending.reserve(old + kBlock);
    if (fast) {
        got = fast->read(pending.data() + old, kBlock);
        if (fast->finished()) in_eof = true;
        if (fast->error()) fprintf(stderr, "****Warning: '%s': %s; the input ends there.\n", path, fast->error());
    } else if (file) {
        // in pieces: a gzread call that runs into damaged data returns nothing at all, and what
        // the earlier pieces delivered is kept (the reference's gzgets keeps the lines before it)
        constexpr unsigned kPiece = 1u << 20;
        while (got < kBlock) {
            const int r = gzread(file, pending.data() + old + got, kPiece);
            if (r > 0) got += (size_t)r;
            if (r < (int)kPiece) {
                in_eof = true;
                int zerr = Z_OK;
                const char *what = gzerror(file, &zerr);
                if (r < 0 || (zerr != Z_OK && zerr != Z_STREAM_END))
                    fprintf(stderr, "****Warning: '%s': %s; the input ends there.\n", path, what);
                break;
            }
        }
    } else if (mapped) {
        const uint64_t want = std::min<uint64_t>(kBlock, file_size - file_pos);
        if (want == 0) in_eof = true;
        got = (size_t)want; // pending (a view that ends at file_pos) simply grows
        file_pos += want;
        if (file_pos >= file_size) in_eof = true;
    } else if (regular) {
        // one block = several slices pread concurrently on the host pool (a single read(2) stream
        // is a page-cache memcpy on one core, ~3 GB/s; slices scale with the cores)
        const uint64_t want = std::min<uint64_t>(kBlock, file_size - file_pos);
        if (want == 0) {
            in_eof = true;
        } else {
            char *dst = pending.data() + old;
            const uint64_t base = file_pos;
            static const uint64_t max_slices = [] {
                const char *e = getenv("SICKLE_READ_SLICES");
                return e && atoi(e) > 0 ? (uint64_t)atoi(e) : (uint64_t)8;
            }();
            const size_t slices = (size_t)std::min<uint64_t>(max_slices, (want + (4u << 20) - 1) / (4u << 20));
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
                fatal_exit(EXIT_FAILURE);
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
    if (pending.is_borrowed()) {
        rest.borrow(pending.data() + cut, rest_bytes);
    } else {
        rest.reserve(rest_bytes + 1);
        memcpy(rest.data(), pending.data() + cut, rest_bytes);
        rest.set_size(rest_bytes);
    }
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
