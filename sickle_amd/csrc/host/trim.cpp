#include "trim.h"

#include "Deflate.h"
#include "FqDeflate.h"
#include "WorkerPool.h"

#include <errno.h>
#include <time.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <iostream>

void error(const char *content)
{
    std::cerr << "[ERROR] " << content << std::endl;
    std::flush(std::cerr);
}
void error(const std::string &content) { error(content.c_str()); }

void msg(const char *content)
{
    static const bool on = [] {
        const char *e = getenv("SICKLE_DEBUG_CHATTER");
        return e && *e && *e != '0';
    }();
    if (on) std::cout << "[DEBUGGING] " << content << std::endl;
}
void msg(const std::string &content) { msg(content.c_str()); }

bool sickle_leave_fast = false;
int sickle_done_fd = -1;

// SICKLE_STAGE_TIMES=1: the wall clock (CLOCK_REALTIME, what `date +%s.%N` shows) at main() and at the exit, so that a
// script can tell start-up and teardown from the run
void sickle_wallclock_mark(const char *what)
{
    const char *e = getenv("SICKLE_STAGE_TIMES");
    if (!e || !*e || *e == '0') return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    fprintf(stderr, "[wall] %-8s %ld.%09ld\n", what, (long)ts.tv_sec, ts.tv_nsec);
}

void fatal_exit(int status)
{
    sickle_wallclock_mark("exit");
    std::cout.flush();
    std::cerr.flush();
    fflush(stdout);
    fflush(stderr);
    if (sickle_done_fd >= 0) { // sickle.h: the front process leaves with this status now
        const unsigned char code = (unsigned char)status;
        ::close(1); // whoever reads our stdout / stderr through a pipe sees their end now, not after the teardown
        ::close(2);
        (void)!::write(sickle_done_fd, &code, 1);
        ::close(sickle_done_fd);
    }
    _exit(status);
}

void sickle_front_process()
{
    const char *off = getenv("SICKLE_NO_FRONT");
    if (off && *off && *off != '0') return;
    // a profiler's preloaded tool library brings the GPU runtime up before main(): a fork would hand the child a
    // runtime it cannot use
    const char *pl = getenv("LD_PRELOAD");
    if ((pl && (strstr(pl, "rocprof") || strstr(pl, "roctracer") || strstr(pl, "rocprofiler"))) || getenv("HSA_TOOLS_LIB") ||
        getenv("ROCP_TOOL_LIBRARIES"))
        return;
    int fds[2];
    if (pipe(fds) != 0) return;
    // A caller may have started us with some of stdin / stdout / stderr closed: pipe() then hands out 0, 1 or 2, and
    // the worker's summary or error text would go into the status pipe (the front process would leave with its
    // first byte as the exit status).  Both ends go above 2.
    for (int &fd : fds) {
        if (fd > 2) continue;
        const int up = fcntl(fd, F_DUPFD_CLOEXEC, 3);
        if (up < 0) { // no descriptor to be had: do the work in this process
            ::close(fds[0]);
            ::close(fds[1]);
            return;
        }
        ::close(fd);
        fd = up;
    }
    fflush(stdout);
    fflush(stderr);
    const pid_t parent = getpid();
    const pid_t pid = fork();
    if (pid < 0) {
        ::close(fds[0]);
        ::close(fds[1]);
        return; // no second process: do the work here
    }
    if (pid == 0) {
        ::close(fds[0]);
        sickle_done_fd = fds[1];
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        if (getppid() != parent) _exit(1); // the front process is gone already
        return;
    }
    ::close(fds[1]);
    unsigned char code = 0;
    ssize_t n;
    do n = ::read(fds[0], &code, 1); while (n < 0 && errno == EINTR);
    if (n == 1) _exit(code);
    int st = 0;
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
    _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st));
}

// ---------------------------------------------------------------- OutFile
bool OutFile::open(const char *path, bool gz)
{
    fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    pos = 0;
    gzip = gz;
    gz_level = 6; // zlib's default, what the reference's gzopen(path, "w") uses
    if (const char *e = getenv("SICKLE_GZ_LEVEL"))
        gz_level = strcmp(e, "gpu") == 0 ? -2 : strcmp(e, "fast") == 0 ? -1 : std::max(1, std::min(9, atoi(e)));
    struct stat st;
    seekable = fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    return fd >= 0;
}

static void write_all(int fd, const char *p, size_t n)
{
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w <= 0) return;
        p += w;
        n -= (size_t)w;
    }
}

static void pwrite_all(int fd, const char *p, size_t n, uint64_t at)
{
    while (n) {
        const ssize_t w = pwrite(fd, p, n, (off_t)at);
        if (w <= 0) return; // like the reference's ofstream: write errors are not reported
        p += w;
        n -= (size_t)w;
        at += (uint64_t)w;
    }
}

void OutFile::put(const char *p, size_t n)
{
    if (seekable) pwrite_all(fd, p, n, pos);
    else write_all(fd, p, n);
    pos += n;
}

// gzip output is written as BGZF (the blocked gzip of htslib's bgzip: SAM/BAM spec 4.1): every block
// is a complete gzip member of at most 64 KiB whose header carries its own compressed size, so any
// gzip reader inflates the file as one stream (RFC 1952 2.2) and a BGZF-aware one -- GZReader
// here, bgzip -@, samtools -- finds the block boundaries without inflating and works on the
// blocks in parallel.  The blocks are made by FqDeflate.cpp on the worker pool.

int OutFile::gpu_device = 0;

// the parts of a batch -> BGZF members, the deflate streams made by sk_bgzf_deflate.  The text goes
// through a pinned staging buffer in runs of up to kGpuRun blocks (pageable copies run at a fifth of
// the PCIe rate), copied in and framed out on the worker pool.
void OutFile::gpu_bgzf(const std::vector<std::string> &parts)
{
    constexpr size_t kGpuRun = 1024; // blocks per device call: 64 MiB of text
    struct Piece {
        const char *p;
        uint32_t n;
    };
    std::vector<Piece> pieces;
    for (const std::string &s : parts)
        for (size_t at = 0; at < s.size(); at += kBgzfInput) pieces.push_back({s.data() + at, (uint32_t)std::min(kBgzfInput, s.size() - at)});
    if (pieces.empty()) return;
    if (!pin_text) {
        pin_text = (unsigned char *)sk_bgzf_host_alloc(kGpuRun * kBgzfInput);
        pin_out = (unsigned char *)sk_bgzf_host_alloc(kGpuRun * 65536);
        if (!pin_text || !pin_out) {
            error("could not allocate pinned staging for the GPU deflate");
            fatal_exit(EXIT_FAILURE);
        }
    }
    WorkerPool &pool = WorkerPool::instance();
    std::vector<uint32_t> sizes(kGpuRun), csize(kGpuRun);
    std::vector<size_t> at(kGpuRun + 1);
    std::string framed;
    for (size_t first = 0; first < pieces.size(); first += kGpuRun) {
        const size_t nb = std::min(kGpuRun, pieces.size() - first);
        pool.parallel_for(nb, std::min<size_t>(nb, (size_t)pool.size()), [&](size_t lo, size_t hi, size_t) {
            for (size_t b = lo; b < hi; ++b) {
                memcpy(pin_text + b * kBgzfInput, pieces[first + b].p, pieces[first + b].n);
                sizes[b] = pieces[first + b].n;
            }
        });
        const int rc = sk_bgzf_deflate(gpu_device, pin_text, sizes.data(), (uint32_t)nb, pin_out, csize.data());
        if (rc != SK_OK) { // no CPU fallback behind the GPU setting
            error(std::string("sk_bgzf_deflate failed: ") + sk_bgzf_last_error());
            fatal_exit(EXIT_FAILURE);
        }
        at[0] = 0;
        for (size_t b = 0; b < nb; ++b) {
            if (!(csize[b] && csize[b] < sizes[b] + 5)) csize[b] = 0; // did not compress: a stored block
            at[b + 1] = at[b] + 26 + (csize[b] ? csize[b] : sizes[b] + 5);
        }
        framed.resize(at[nb]);
        pool.parallel_for(nb, std::min<size_t>(nb, (size_t)pool.size()), [&](size_t lo, size_t hi, size_t) {
            for (size_t b = lo; b < hi; ++b) {
                const unsigned char *p = (const unsigned char *)pieces[first + b].p;
                const uint32_t n = sizes[b];
                unsigned char *m = (unsigned char *)framed.data() + at[b];
                memcpy(m, kBgzfEofBlock, 16);
                size_t clen;
                if (csize[b]) {
                    clen = csize[b];
                    memcpy(m + 18, pin_out + b * 65536, clen);
                } else {
                    clen = n + 5;
                    m[18] = 1;
                    m[19] = (unsigned char)(n & 0xff);
                    m[20] = (unsigned char)(n >> 8);
                    m[21] = (unsigned char)(~n & 0xff);
                    m[22] = (unsigned char)((~n >> 8) & 0xff);
                    memcpy(m + 23, p, n);
                }
                const uint32_t whole = (uint32_t)(18 + clen + 8), crc = deflate_crc32(p, n);
                m[16] = (unsigned char)((whole - 1) & 0xff);
                m[17] = (unsigned char)((whole - 1) >> 8);
                unsigned char *tail = m + 18 + clen;
                for (int i = 0; i < 4; ++i) tail[i] = (unsigned char)(crc >> (8 * i));
                for (int i = 0; i < 4; ++i) tail[4 + i] = (unsigned char)(n >> (8 * i));
            }
        });
        put(framed.data(), framed.size());
    }
}

void OutFile::write(const std::string &data)
{
    if (data.empty() || fd < 0) return;
    if (!gzip) {
        put(data.data(), data.size());
        return;
    }
    // The reference hands the text to gzprintf as the FORMAT string (src/trim_single.cpp:418),
    // which mangles any '%' (= Sanger Q4).  This writes the bytes themselves.
    std::vector<std::string> one;
    one.push_back(data);
    write_parts(one);
}

void OutFile::write_parts(const std::vector<std::string> &parts)
{
    if (fd < 0) return;
    if (gzip) {
        struct Piece {
            const char *p;
            size_t n;
        };
        std::vector<Piece> pieces;
        for (const std::string &s : parts)
            for (size_t at = 0; at < s.size(); at += kBgzfInput)
                pieces.push_back({s.data() + at, std::min(kBgzfInput, s.size() - at)});
        if (pieces.empty()) return;
        if (gz_level == -2) { // the blocks are deflated on the GPU and framed here
            gpu_bgzf(parts);
            return;
        }
        const size_t groups = std::min(pieces.size(), (size_t)WorkerPool::instance().size() * 4);
        std::vector<std::string> packed(groups);
        WorkerPool::instance().parallel_for(pieces.size(), groups, [&](size_t lo, size_t hi, size_t g) {
            packed[g].reserve((hi - lo) * (kBgzfInput / 3));
            for (size_t i = lo; i < hi; ++i) bgzf_append_block(pieces[i].p, pieces[i].n, gz_level, packed[g]);
        });
        for (const std::string &m : packed) put(m.data(), m.size());
        return;
    }
    if (!seekable) {
        for (const std::string &p : parts) write(p);
        return;
    }
    std::vector<uint64_t> at(parts.size() + 1, pos);
    for (size_t i = 0; i < parts.size(); ++i) at[i + 1] = at[i] + parts[i].size();
    // one writer per file by default (the files of a batch are still written side by side):
    // concurrent writers to ONE file queue up on its page-cache locks.  Measured on tmpfs, 10 M
    // pairs, whole run (tools/probes/ab_env.py): 1 writer 1.77 s, 2: 2.12 s, 4: 2.08 s, 8: 1.98 s,
    // and a quarter of the system time.
    static const size_t writers = [] {
        const char *e = getenv("SICKLE_WRITE_SLICES");
        return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)1;
    }();
    WorkerPool::instance().parallel_for(parts.size(), std::min(parts.size(), writers), [&](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; ++i) pwrite_all(fd, parts[i].data(), parts[i].size(), at[i]);
    });
    pos = at[parts.size()];
}

void OutFile::close()
{
    if (fd >= 0) {
        if (gzip) put((const char *)kBgzfEofBlock, sizeof kBgzfEofBlock); // the empty block that ends a BGZF file
        sk_bgzf_host_free(pin_text);
        sk_bgzf_host_free(pin_out);
        pin_text = pin_out = nullptr;
        ::close(fd);
    }
    fd = -1;
}

// ---------------------------------------------------------------- Abstract_Trimmer
Abstract_Trimmer::Abstract_Trimmer()
    : qualtype(-1), length_threshold(20), qual_threshold(20), no_fiveprime(0), trunc_n(0), debug(0), threads(1),
      batch_len(1024 * 1024 * DEFAULT_BATCH_LEN), input(nullptr), outfn(nullptr), infn(nullptr), quiet(0),
      gzip_output(0), kept(0), discard(0), total(0)
{
}

Abstract_Trimmer::~Abstract_Trimmer() { close_device(); }

int Abstract_Trimmer::open_device()
{
    if (!device_ids.empty()) return 0;
    // which GPUs: SICKLE_DEVICES=0,1,2 (several) or SICKLE_DEVICE=n (one), default device 0
    if (const char *list = getenv("SICKLE_DEVICES")) {
        for (const char *p = list; *p;) {
            char *end;
            const long d = strtol(p, &end, 10);
            if (end == p) break;
            device_ids.push_back((int)d);
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (device_ids.empty()) {
        const char *e = getenv("SICKLE_DEVICE");
        device_ids.push_back(e ? atoi(e) : 0);
    }
    OutFile::gpu_device = device_ids[0];
    ctxs.assign(device_ids.size(), nullptr);
    slots.assign((size_t)n_slots(), Slot());
    device_opener = std::thread([this] {
        // Pinned staging sized for a whole ingest batch, allocated during start-up so that hipHostMalloc
        // (0.08 s per 0.4 GB) also hides behind the first reads.  A batch holds at most ~batch_len
        // bytes of text per input file; quality is under half of it.  Too small only means a
        // later grow().  With one device the pinning runs on a thread of its own, beside the creation of
        // the context (the first hipStreamCreate takes 0.16 s): tools/probes/hip_init_parallel.py -- both are
        // done after 0.27 s instead of 0.32 s.
        const size_t text = (size_t)batch_len * (size_t)staging_files;
        auto pin_all = [this, text](bool with_ctx) {
            for (size_t i = 0; i < slots.size(); ++i) {
                {   // a published slot belongs to the main thread from then on
                    std::lock_guard<std::mutex> lk(open_lock);
                    if (slots_ready > (int)i) continue;
                }
                // without a context a failure is not fatal here: the pass with the context repeats the slot and
                // reports (no device at all must end in sk_create's message, not in this one)
                if (!grow(with_ctx ? ctx_of((int)i) : nullptr, slots[i], text / 2 + (text >> 4), text / 96 + 1024, trunc_n != 0)) return;
                std::lock_guard<std::mutex> lk(open_lock);
                if (slots_ready < (int)i + 1) slots_ready = (int)i + 1;
                open_cv.notify_all();
            }
        };
        std::thread pinner;
        // (the context-free allocator pins through the current device of its thread, i.e. device 0: only there
        // does the early pinning belong to the context the scans will run in)
        if (device_ids.size() == 1 && device_ids[0] == 0) pinner = std::thread(pin_all, false);
        for (size_t g = 0; g < device_ids.size(); ++g) {
            const int rc = sk_create(device_ids[g], kSlots, &ctxs[g]);
            if (rc != SK_OK) {
                // no CPU fallback: the scan exists only as HIP kernels for gfx950
                fprintf(stderr, "****Error: no usable MI355X (gfx950) device %d for the quality scan (sk_create: %d).\n\n",
                        device_ids[g], rc);
                ctxs[g] = nullptr;
                if (pinner.joinable()) pinner.join();
                std::lock_guard<std::mutex> lk(open_lock);
                open_failed = open_done = true;
                open_cv.notify_all();
                return;
            }
        }
        {
            std::lock_guard<std::mutex> lk(open_lock);
            ctx_ready = true;
            open_cv.notify_all();
        }
        if (pinner.joinable()) pinner.join();
        pin_all(true); // nothing to do for the slots the pinner finished
        std::lock_guard<std::mutex> lk(open_lock);
        devices_ok = open_done = true;
        open_cv.notify_all();
    });
    return 0;
}

void Abstract_Trimmer::ensure_device()
{
    if (device_opener.joinable()) device_opener.join();
    if (!devices_ok) fatal_exit(EXIT_FAILURE); // the opener has printed why
}

void Abstract_Trimmer::ensure_slot(int slot)
{
    std::unique_lock<std::mutex> lk(open_lock);
    open_cv.wait(lk, [&] { return open_done || (ctx_ready && slots_ready > slot); });
    if (open_failed) fatal_exit(EXIT_FAILURE); // the opener has printed why
}

void Abstract_Trimmer::close_device()
{
    if (device_opener.joinable()) device_opener.join();
    for (size_t i = 0; i < slots.size(); ++i) {
        sk_ctx *c = ctxs[i % ctxs.size()];
        if (!c) continue;
        Slot &s = slots[i];
        sk_host_free(c, s.qual);
        sk_host_free(c, s.seq);
        sk_host_free(c, s.offsets);
        sk_host_free(c, s.cuts);
        s = Slot();
    }
    for (sk_ctx *&c : ctxs) {
        if (c) sk_destroy(c);
        c = nullptr;
    }
    slots.clear();
    ctxs.clear();
    device_ids.clear();
    devices_ok = false;
}

bool Abstract_Trimmer::grow(sk_ctx *ctx, Slot &s, size_t bytes, size_t reads, bool need_seq)
{
    // ctx == nullptr: start-up, the context is still being created on another thread -- the context-free
    // pinned allocator of the library (the same hipHostMalloc / hipHostFree underneath)
    auto fail = [&](const char *what) {
        if (!ctx) return false;
        fprintf(stderr, "****Error: could not allocate pinned %s buffer: %s\n\n", what, sk_last_error(ctx));
        fatal_exit(EXIT_FAILURE);
        return false;
    };
    auto sk_host_alloc = [](sk_ctx *c, size_t bytes) -> void * { return c ? ::sk_host_alloc(c, bytes) : sk_bgzf_host_alloc(bytes); };
    auto sk_host_free = [](sk_ctx *c, void *p) { if (c) ::sk_host_free(c, p); else sk_bgzf_host_free(p); };
    if (bytes + 64 > s.cap_bytes || (need_seq && !s.seq)) {
        const size_t cap = std::max(s.cap_bytes, bytes + 64 + (bytes >> 3));
        sk_host_free(ctx, s.qual);
        sk_host_free(ctx, s.seq);
        s.seq = nullptr;
        s.qual = (uint8_t *)sk_host_alloc(ctx, cap);
        if (!s.qual) return fail("quality");
        if (need_seq) {
            s.seq = (uint8_t *)sk_host_alloc(ctx, cap);
            if (!s.seq) return fail("sequence");
        }
        s.cap_bytes = cap;
    }
    if (reads + 1 > s.cap_reads) {
        const size_t cap = reads + 1 + (reads >> 3);
        sk_host_free(ctx, s.offsets);
        sk_host_free(ctx, s.cuts);
        s.offsets = (uint64_t *)sk_host_alloc(ctx, cap * sizeof(uint64_t));
        s.cuts = (sk_cut *)sk_host_alloc(ctx, cap * sizeof(sk_cut));
        if (!s.offsets || !s.cuts) return fail("index");
        s.cap_reads = cap;
    }
    return true;
}

size_t Abstract_Trimmer::piece_reads() const
{
    static const size_t forced = [] { const char *e = getenv("SICKLE_SUBBATCH_READS"); return e ? (size_t)atoll(e) : (size_t)0; }();
    if (threads != 1) return (size_t)-1;
    return forced ? forced : (size_t)600000;
}

void Abstract_Trimmer::submit_scan(int slot, Span<FQEntry> reads)
{
    ensure_slot(slot);
    Slot &s = slots[(size_t)slot];
    sk_ctx *ctx = ctx_of(slot);
    const int dev_slot = slot / (int)ctxs.size();
    const size_t n = reads.size();
    // Layout.  Equal-length batches (the usual case): fixed stride, uniform length -> the tiled
    // kernel with the matrix-pipe window sums.  Mixed lengths up to the tiled kernel's limit: the
    // reads are grouped by length (counting sort) into tiles of <= 64 equal-length rows, each group
    // at its own stride, described to the device by one sk_tile per tile; every tile is uniform
    // inside, so the batch stays on the same fast path with no padding to the longest read, and
    // the device scatters the cuts back to input order (out_index).  Longer reads of ONE length (up
    // to 4 096 bases): back to back at a fixed stride = their length -- the library's medium-read tiles
    // take those up to ~2 200 bases on the matrix path, its general kernels the rest.  Anything else:
    // packed back to back with an offsets array -> the general kernels.
    size_t total_len = 0, max_len = 0;
    bool uniform = true;
    const size_t len0 = n ? reads[0].qual.length() : 0;
    WorkerPool &pool = WorkerPool::instance();
    const size_t parts = (size_t)pool.size() * 4;
    {
        struct Acc {
            size_t total = 0, longest = 0;
            bool same = true;
        };
        std::vector<Acc> acc(parts);
        pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t part) {
            Acc a;
            for (size_t i = lo; i < hi; ++i) {
                const size_t l = reads[i].qual.length();
                a.total += l;
                if (l > a.longest) a.longest = l;
                a.same = a.same && l == len0;
            }
            acc[part] = a;
        });
        for (const Acc &a : acc) {
            total_len += a.total;
            if (a.longest > max_len) max_len = a.longest;
            uniform = uniform && a.same;
        }
    }
    // stride: a multiple of 8 with an ODD number of 8-byte units, so that the per-lane row walks of
    // the tiled kernel (ds_read_b64 at lane*stride) spread over all LDS banks; an even count
    // (e.g. 250 -> 256) puts every lane on the same banks (measured: 2.6x slower)
    auto stride_for = [](size_t len) {
        size_t s8 = (len + 7) / 8;
        if (s8 % 2 == 0) ++s8;
        return s8 * 8;
    };
    const bool tiled = n > 0 && max_len > 0 && stride_for(max_len) <= SK_TILE_MAX_STRIDE;
    const bool segmented = tiled && !uniform;
    const bool medium = !tiled && uniform && n > 0 && len0 > 0 && len0 <= 4096;
    const bool need_seq = trunc_n != 0;

    sk_batch b;
    memset(&b, 0, sizeof b);
    if (segmented) {
        // group g = reads of length g: first slot, byte offset of its first tile, stride
        std::vector<uint32_t> first_slot(max_len + 2, 0);
        for (const FQEntry &r : reads) first_slot[r.qual.length() + 1]++;
        for (size_t l = 0; l <= max_len; ++l) first_slot[l + 1] += first_slot[l];
        std::vector<uint64_t> group_off(max_len + 1, 0);
        s.tiles.clear();
        uint64_t at = 0;
        for (size_t l = 1; l <= max_len; ++l) {
            const uint32_t cnt = first_slot[l + 1] - first_slot[l];
            if (!cnt) continue;
            const uint32_t st = (uint32_t)stride_for(l);
            at = (at + 15) & ~(uint64_t)15;
            group_off[l] = at;
            for (uint32_t a0 = 0; a0 < cnt; a0 += 64) {
                sk_tile t;
                t.byte_off = at + (uint64_t)a0 * st;
                t.slot0 = first_slot[l] + a0;
                t.stride = st;
                t.rows = (uint16_t)std::min<uint32_t>(64, cnt - a0);
                t.read_len = (uint16_t)l;
                t.reserved = 0;
                s.tiles.push_back(t);
            }
            at += (uint64_t)cnt * st;
        }
        grow(ctx, s, (size_t)at, n, need_seq);
        // slot of every read (stable within a length), then the copies, on the pool
        uint32_t *out_index = reinterpret_cast<uint32_t *>(s.offsets); // the index buffer holds either
        std::vector<uint32_t> next = first_slot;
        std::vector<uint32_t> slot_of(n);
        for (size_t i = 0; i < n; ++i) slot_of[i] = next[reads[i].qual.length()]++;
        pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; ++i) {
                const size_t l = reads[i].qual.length();
                const uint32_t slot = slot_of[i];
                const uint64_t dst = group_off[l] + (uint64_t)(slot - first_slot[l]) * stride_for(l);
                memcpy(s.qual + dst, reads[i].qual.data(), l);
                if (need_seq) memcpy(s.seq + dst, reads[i].seq.data(), l);
                out_index[slot] = (uint32_t)i;
            }
        });
        b.stride = (uint32_t)stride_for(max_len);
        b.tiles = s.tiles.data();
        b.n_tiles = (uint32_t)s.tiles.size();
        b.out_index = out_index;
        b.cuts_in_slot_order = 1;
        s.slot_order = true;
    } else if (tiled) {
        const size_t stride = stride_for(len0);
        grow(ctx, s, n * stride, n, need_seq);
        pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; ++i) {
                memcpy(s.qual + i * stride, reads[i].qual.data(), len0);
                if (need_seq) memcpy(s.seq + i * stride, reads[i].seq.data(), len0);
            }
        });
        b.stride = (uint32_t)stride;
        b.read_len = (uint32_t)len0;
    } else if (medium) {
        grow(ctx, s, total_len, n, need_seq);
        pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; ++i) {
                memcpy(s.qual + i * len0, reads[i].qual.data(), len0);
                if (need_seq) memcpy(s.seq + i * len0, reads[i].seq.data(), len0);
            }
        });
        b.stride = (uint32_t)len0;
        b.read_len = (uint32_t)len0;
    } else {
        grow(ctx, s, total_len, n, need_seq);
        size_t at = 0;
        for (size_t i = 0; i < n; ++i) {
            s.offsets[i] = at;
            at += reads[i].qual.length();
        }
        s.offsets[n] = at;
        pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; ++i) {
                const size_t l = reads[i].qual.length();
                memcpy(s.qual + s.offsets[i], reads[i].qual.data(), l);
                if (need_seq) memcpy(s.seq + s.offsets[i], reads[i].seq.data(), l);
            }
        });
        b.offsets = s.offsets;
    }
    if (!segmented) s.slot_order = false;
    b.qual = s.qual;
    b.seq = need_seq ? s.seq : nullptr;
    b.n_reads = n;
    sk_params p = {qualtype, qual_threshold, length_threshold, no_fiveprime, trunc_n};
    const int rc = sk_submit(ctx, dev_slot, &p, &b, s.cuts);
    if (rc != SK_OK) {
        // the one limit the reference does not have: a read beyond SK_MAX_READ_LEN (16 Mi bases; the device's error
        // word keeps 24 bits of position) -- say so, with the record, instead of the general message
        for (size_t i = 0; i < n; ++i)
            if (reads[i].qual.length() > SK_MAX_READ_LEN) {
                fprintf(stderr, "****Error: record '%.*s' has %zu bases; this build scans reads of up to %u bases.\n\n",
                        (int)std::min<size_t>(reads[i].name.length(), 200), reads[i].name.data(), reads[i].qual.length(), SK_MAX_READ_LEN);
                fatal_exit(EXIT_FAILURE);
            }
        fprintf(stderr, "****Error: device scan could not be started (%d): %s\n\n", rc, sk_last_error(ctx));
        fatal_exit(EXIT_FAILURE);
    }
}

const cutsites *Abstract_Trimmer::wait_scan(int slot, Span<FQEntry> reads)
{
    static_assert(sizeof(cutsites) == sizeof(sk_cut), "cutsites must match the C ABI's sk_cut");
    sk_err e;
    sk_ctx *ctx = ctx_of(slot);
    const int rc = sk_wait(ctx, slot / (int)ctxs.size(), &e);
    if (rc == SK_ERANGE) {
        // reference src/trim.cpp:130-136
        const FQEntry &r = reads[e.read];
        const int32_t *k = sk_quality_constants(qualtype);
        const char *tn = sk_typename(qualtype);
        fprintf(stderr, "ERROR: Quality value (%d) does not fall within correct range for %s encoding.\n", e.ch, tn);
        fprintf(stderr, "Range for %s encoding: %d-%d\n", tn, k[1], k[2]);
        fprintf(stderr, "FastQ record: %s\n", std::string(r.name).c_str());
        fprintf(stderr, "Quality string: %s\n", std::string(r.qual).c_str());
        fprintf(stderr, "Quality char: '%c'\n", (char)e.ch);
        fprintf(stderr, "Quality position: %d\n", (int)e.pos + 1);
        fatal_exit(1);
    }
    if (rc != SK_OK) {
        fprintf(stderr, "****Error: device scan failed (%d): %s\n\n", rc, sk_last_error(ctx));
        fatal_exit(EXIT_FAILURE);
    }
    Slot &s = slots[(size_t)slot];
    const cutsites *cuts = reinterpret_cast<const cutsites *>(s.cuts);
    if (!s.slot_order) return cuts;
    // slot order -> read order (the index buffer of a segmented batch lives in s.offsets)
    const size_t n = reads.size();
    const uint32_t *out_index = reinterpret_cast<const uint32_t *>(s.offsets);
    s.ordered.resize(n);
    cutsites *ordered = s.ordered.data();
    WorkerPool &pool = WorkerPool::instance();
    pool.parallel_for(n, (size_t)pool.size(), [&](size_t lo, size_t hi, size_t) {
        for (size_t k = lo; k < hi; ++k) ordered[out_index[k]] = cuts[k];
    });
    return ordered;
}

std::thread Abstract_Trimmer::prefetch_batches(GZReader *reader, Channel<Batch *> &out)
{
    return std::thread([reader, &out] {
        for (;;) {
            Batch *b = reader->get_batch_buffering_lines();
            out.push(b);
            if (!b) break;
        }
        out.close();
    });
}

void Abstract_Trimmer::append_record(std::string &out, const FQEntry &read, const cutsites &cs)
{
    const size_t five = (size_t)cs.five_prime_cut;
    const size_t n = (size_t)(cs.three_prime_cut - cs.five_prime_cut);
    out.append(read.name.data(), read.name.size());
    out.push_back('\n');
    out.append(read.seq.data() + five, n);
    out.push_back('\n');
    out.append(read.comment.data(), read.comment.size());
    out.push_back('\n');
    out.append(read.qual.data() + five, n);
    out.push_back('\n');
}

// reference src/trim_single.cpp:194-211 / src/trim_paired.cpp:246-263 (the caller halves max for PE)
int Abstract_Trimmer::recommended_batch_len_for(const char *path, unsigned long long max_len) const
{
    struct stat st;
    if (stat(path, &st) != 0) {
        // the reference dies here with an uncaught std::filesystem error (abort)
        fprintf(stderr, "****Error: Could not open input file '%s'.\n\n", path);
        fatal_exit(EXIT_FAILURE);
    }
    const unsigned long long min_len = 20;
    const unsigned long long recommended = (unsigned long long)st.st_size / 8;
    unsigned long long chosen = recommended;
    if (recommended < min_len) chosen = min_len;
    else if (recommended > max_len) chosen = max_len;
    msg(std::string("Batch size is ") + std::to_string(chosen / (1024 * 1024)) + std::string("MB"));
    return (int)chosen;
}
