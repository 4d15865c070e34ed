// trim.h -- Abstract_Trimmer: the base of the two mode drivers.
//
// Role of reference src/trim.{h,cpp}.  In the reference this class IS the hot path
// (sliding_window / get_quality_num run on CPU threads).  Here the scan runs on the GPU behind
// the C ABI of include/sickle_amd.h, and this class holds what both drivers share around it:
// the configuration ints (same names as reference src/trim.h:16-30), the device session with
// its pinned staging slots, batch packing, record emission and the range-error exit.
#ifndef SICKLE_TRIM_H
#define SICKLE_TRIM_H

#include <sys/resource.h>
#include <zlib.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <initializer_list>
#include <utility>
#include <cstdio>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "FQEntry.h"
#include "GZReader.h"
#include "sickle.h"
#include "sickle_amd.h"

// plain or gzip output file (reference: std::ofstream / gzFile pairs, src/trim.h:24-25)
class OutFile {
public:
    bool open(const char *path, bool gzip);
    void write(const std::string &data);
    // the pieces of one batch, in order: plain files take them as concurrent pwrite(2) slices
    // (a single write stream is a page-cache memcpy on one core); gzip files are deflated on the
    // worker pool as BGZF blocks (gzip members of <= 64 KiB) and the blocks written in order
    void write_parts(const std::vector<std::string> &parts);
    void close();
    bool is_open() const { return fd >= 0; }
    static int gpu_device; // SICKLE_GZ_LEVEL=gpu: the device that deflates the blocks

private:
    void put(const char *p, size_t n); // at the current position
    void gpu_bgzf(const std::vector<std::string> &parts);
    unsigned char *pin_text = nullptr, *pin_out = nullptr; // pinned staging of the GPU deflate
    int fd = -1;
    bool seekable = false; // a regular file: positional writes; pipes and devices get plain write(2)
    bool gzip = false;
    int gz_level = 6;  // SICKLE_GZ_LEVEL=1..9: zlib at that level; =fast (-1): FqDeflate on the host; =gpu (-2): on the GPU
    uint64_t pos = 0;
};

// Wall-clock accumulators per pipeline stage, printed to stderr when SICKLE_STAGE_TIMES=1.
class StageClock {
public:
    class Scope {
    public:
        explicit Scope(StageClock &c) : clk(&c), t0(std::chrono::steady_clock::now()) {}
        void stop()
        {
            if (!clk) return;
            clk->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            clk = nullptr;
        }
        ~Scope() { stop(); }

    private:
        StageClock *clk;
        std::chrono::steady_clock::time_point t0;
    };
    static void report(std::initializer_list<std::pair<const char *, StageClock *>> stages)
    {
        const char *e = getenv("SICKLE_STAGE_TIMES");
        if (!e || !*e || *e == '0') return;
        for (const auto &s : stages) fprintf(stderr, "[stage] %-16s %8.3f s\n", s.first, s.second->seconds);
    }
    double seconds = 0;
    // timeline marks (seconds since the first mark), also under SICKLE_STAGE_TIMES=1
    static void mark(const char *what)
    {
        static const bool on = [] {
            const char *e = getenv("SICKLE_STAGE_TIMES");
            return e && *e && *e != '0';
        }();
        if (!on) return;
        static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[mark] %7.3f s  %s  (cpu so far: user %.2f s, system %.2f s; peak RSS %.2f GB)\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what,
                ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6,
                ru.ru_maxrss / 1048576.0);
    }
};

// bounded hand-off between pipeline stages
template <typename T> class Channel {
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void push(T v)
    {
        std::unique_lock<std::mutex> lk(m_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    bool pop(T &out) // false once closed and drained
    {
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        not_empty_.notify_all();
    }

private:
    std::mutex m_;
    std::condition_variable not_empty_, not_full_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

class Abstract_Trimmer {
public:
    virtual int parse_args(int argc, char *argv[]) = 0;
    virtual int trim_main() = 0;
    virtual void usage(int status, char const *msg) = 0;
    virtual ~Abstract_Trimmer();

protected:
    Abstract_Trimmer();

    // ---- the scan of one batch of reads on the device (what processing_thread did on CPU threads)
    // Two staging slots per GPU.  SICKLE_DEVICES=0,1,... spreads the batches over several GPUs
    // (batch i goes to slot i mod n_slots(), i.e. to GPU i mod G): reads are independent, nothing
    // is exchanged between devices and the writer keeps batch order, so the output does not change.
    static const int kSlots = 2;
    int n_slots() const { return kSlots * (int)device_ids.size(); }
    // Creates the device session on a helper thread (HIP start-up takes ~0.2 s, which the first
    // batch's file reads hide); the first submit_scan waits for it and exits with the message
    // below if there is no usable gfx950 device.  Always returns 0.
    int open_device();
    void require_device() { ensure_device(); }
    void close_device();
    // packs the quality (and, with -n, sequence) bytes of `reads` into the slot's pinned buffers
    // and enqueues H2D + scan + D2H; returns at once
    void submit_scan(int slot, Span<FQEntry> reads);
    // blocks until the slot is done; on an out-of-range quality prints the reference's message
    // (src/trim.cpp:130-135) and exits 1.  The cut array stays valid until the slot is reused.
    const cutsites *wait_scan(int slot, Span<FQEntry> reads);

    // At -a 1 the file order is the input order whatever the batches are, so an ingest batch (up to 256 MiB of
    // text per file) may go through the device and the output stages in several pieces: the first piece is
    // written while the rest is still being packed, and the last one leaves a short tail.  With -a T > 1 the
    // queue-major order is per ingest batch (src/trim_paired.cpp:388-403) and a batch stays whole.
    // -> number of records per piece (SICKLE_SUBBATCH_READS overrides: tests force tiny pieces)
    size_t piece_reads() const;

    // Frames records [0, n) of a batch on all host threads: record i is made from the four lines
    // starting at line first_line(i) with position position(i).  If any record is malformed the
    // FIRST one in order is re-validated on the calling thread, which prints the reference's
    // messages and exits (src/FQEntry.cpp:53-97).
    template <typename LineOf, typename PosOf>
    static void frame_records(RawVec<FQEntry> &reads, const Batch &batch, size_t n, LineOf first_line,
                              PosOf position, size_t dst0 = 0, size_t dst_step = 1);

    // Runs reader->get_batch_buffering_lines() ahead of the consumer on its own thread: batches
    // arrive in order through `out`, a NULL marks the end (the reference's NULL return).
    static std::thread prefetch_batches(GZReader *reader, Channel<Batch *> &out);

    // name\n seq[five,three)\n comment\n qual[five,three)\n  (src/trim_single.cpp:393-396)
    static void append_record(std::string &out, const FQEntry &read, const cutsites &cs);

    int recommended_batch_len_for(const char *path, unsigned long long max_len) const;

    // reference src/trim.h:16-30
    int qualtype;
    int length_threshold;
    int qual_threshold;
    int no_fiveprime;
    int trunc_n;
    int debug;
    int threads, batch_len;
    GZReader *input;
    char *outfn;
    char *infn;
    int quiet;
    int gzip_output;
    int kept;
    int discard;
    int total;
    int staging_files = 1; // input files feeding one device batch (2 for two-file PE)

private:
    struct Slot {
        uint8_t *qual = nullptr, *seq = nullptr;
        size_t cap_bytes = 0;
        uint64_t *offsets = nullptr;
        sk_cut *cuts = nullptr;
        size_t cap_reads = 0;
        std::vector<sk_tile> tiles; // segmented batches: one descriptor per tile
        // segmented batches come back in slot order (one coalesced stream of cuts on the device);
        // wait_scan puts them into read order here, through the out_index it packed
        bool slot_order = false;
        std::vector<cutsites> ordered;
    };
    std::vector<int> device_ids; // set by open_device() before it returns
    std::vector<sk_ctx *> ctxs;  // one per entry of device_ids, created by the opener thread
    bool devices_ok = false;
    std::thread device_opener;
    // the opener publishes its progress: contexts created, then slot after slot pinned -- a batch only waits
    // for ITS slot (pinning 0.4 GB takes 0.1 s; round 1 made the first batch wait for every slot)
    std::mutex open_lock;
    std::condition_variable open_cv;
    int slots_ready = 0;      // slots [0, slots_ready) have their pinned staging
    bool ctx_ready = false;   // every sk_ctx exists
    bool open_failed = false, open_done = false;
    void ensure_device();     // everything the opener does
    void ensure_slot(int slot);
    std::vector<Slot> slots;     // n_slots(): slot s belongs to ctxs[s % G]
    sk_ctx *ctx_of(int slot) { return ctxs[(size_t)slot % ctxs.size()]; }
    bool grow(sk_ctx *ctx, Slot &s, size_t bytes, size_t reads, bool need_seq); // false: only without a context
};

#include "WorkerPool.h"

template <typename LineOf, typename PosOf>
void Abstract_Trimmer::frame_records(RawVec<FQEntry> &reads, const Batch &batch, size_t n, LineOf first_line,
                                     PosOf position, size_t dst0, size_t dst_step)
{
    WorkerPool &pool = WorkerPool::instance();
    const size_t parts = (size_t)pool.size() * 4;
    std::vector<size_t> first_bad(parts, (size_t)-1);
    pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t part) {
        for (size_t i = lo; i < hi; ++i) {
            FQEntry &e = reads[dst0 + i * dst_step];
            e = FQEntry(batch, first_line(i), position(i));
            if (!e.well_formed() && first_bad[part] == (size_t)-1) first_bad[part] = i;
        }
    });
    for (size_t part = 0; part < parts; ++part)
        if (first_bad[part] != (size_t)-1) {
            reads[dst0 + first_bad[part] * dst_step].validate(); // prints and exits
            break;
        }
}

#endif
