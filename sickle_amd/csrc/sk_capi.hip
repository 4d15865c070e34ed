// sk_capi.hip -- the C ABI of include/sickle_amd.h over the kernels of sk_kernels.hip.
// HIP only: there is no CPU path behind these entry points.

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "sickle_amd.h"
#include "sk_device.h"

static_assert(sizeof(sk_tile) == sizeof(sk_tile_dev) && sizeof(sk_tile) == 24, "sk_tile layout");

namespace {

// reference src/sickle.h:85-91
const int32_t kQualityConstants[4][3] = {
    {0, 4, 60},    // PHRED
    {33, 33, 126}, // SANGER
    {64, 58, 112}, // SOLEXA
    {64, 64, 110}, // ILLUMINA
};
// reference src/sickle.h:68-73
const char *const kTypeNames[4] = {"Phred", "Sanger", "Solexa", "Illumina"};

constexpr unsigned long long kNoError = ~0ull;
constexpr uint32_t SK_SEG_MAX_CLASSES = 16;

// device scratch of the regrouping of mixed-length ragged batches (sk_sort.hip): one per stream of scans
struct SortScratch {
    uint64_t *perm = nullptr;            // 8 lists x cap_list tiles x 64 entries
    unsigned long long *lists = nullptr; // 8 lists x cap_list descriptors of 32 bytes
    uint32_t *counts = nullptr;          // 2 x 16 words: tiles per list [8], flags [2] -- one set per scan, in turns: a scan's
                                         // first kernel clears the set of the scan after it (no memset on the launch path)
    uint32_t turn = 0;
    size_t cap_list = 0;
    void release_lists()
    {
        if (perm) (void)hipFree(perm);
        if (lists) (void)hipFree(lists);
        perm = nullptr, lists = nullptr, cap_list = 0;
    }
    void release()
    {
        release_lists();
        if (counts) (void)hipFree(counts);
        counts = nullptr;
    }
};

struct Slot {
    bool busy = false;
    SortScratch sort;
    uint8_t *d_qual = nullptr, *d_seq = nullptr;
    size_t cap_bytes = 0;
    uint64_t *d_offsets = nullptr;
    uint32_t *d_lengths = nullptr;
    size_t cap_reads = 0;
    sk_tile_dev *d_tiles = nullptr;
    uint32_t *d_out_index = nullptr;
    size_t cap_tiles = 0, cap_index = 0;
    sk_cut_dev *d_out = nullptr;
    std::vector<sk_seg_class> classes;   // segmented batches without a class table: cut here
    unsigned long long *d_err = nullptr; // device error word of this slot
    unsigned long long *h_err = nullptr; // pinned copy
    hipEvent_t copied = nullptr;         // H2D done (copy stream)
    hipEvent_t finished = nullptr;       // D2H of cuts + error word done (compute stream)
};

} // namespace

struct sk_ctx {
    int device = 0;
    int cu_count = 256;
    hipStream_t compute = nullptr, copy = nullptr;
    uint32_t *d_band = nullptr;          // the band matrices of every window width the tile kernels take (sk_device.h)
    unsigned long long *d_err = nullptr; // error word of the device-resident path on the NULL stream
    unsigned long long *h_err = nullptr;
    // ... and one per other stream the caller scans on: errors of scans enqueued on different streams do
    // not meet in one word (each sk_scan_device_finish reports what ITS stream's scans found)
    struct ErrWord {
        unsigned long long *d = nullptr, *h = nullptr; // d: error word, hand-over word, four pair counters (6 x 8 bytes)
        SortScratch sort;
    };
    SortScratch sort0; // ... of the NULL stream
    std::map<hipStream_t, ErrWord> stream_err;
    std::mutex stream_err_lock;
    std::vector<Slot> slots;
    unsigned long long scan_counter = 0;
    char last_error[512] = {0};
};

namespace {

// the error word (device, pinned host copy) of the device-resident scans on `stream`
int err_word_of(sk_ctx *ctx, hipStream_t stream, unsigned long long **d, unsigned long long **h, SortScratch **sort = nullptr);

void set_error(sk_ctx *ctx, const char *fmt, ...)
{
    if (!ctx) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->last_error, sizeof ctx->last_error, fmt, ap);
    va_end(ap);
}

#define SK_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error((ctx), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SK_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

bool tile_eligible(const sk_batch *b);

int make_args(sk_ctx *ctx, const sk_params *p, const sk_batch *b, sk_scan_args *a)
{
    if (!p || !b) return SK_EINVAL;
    if (p->qualtype < 0 || p->qualtype > 3) {
        set_error(ctx, "invalid qualtype %d", p->qualtype);
        return SK_EINVAL;
    }
    if (p->qual_threshold < 0 || p->length_threshold < 0) {
        set_error(ctx, "thresholds must be >= 0"); // reference src/trim_single.cpp:124,132
        return SK_EINVAL;
    }
    if (b->n_reads > 0 && !b->qual) {
        set_error(ctx, "qual is NULL");
        return SK_EINVAL;
    }
    if (p->trunc_n && b->n_reads > 0 && !b->seq) {
        set_error(ctx, "trunc_n needs seq");
        return SK_EINVAL;
    }
    if (b->tiles) {
        if (b->offsets || b->lengths || !b->out_index || b->n_tiles == 0 || b->stride % 8 != 0 ||
            b->stride > SK_TILE_MAX_STRIDE || b->stride == 0) {
            set_error(ctx, "segmented batch: need tiles, out_index, n_tiles and the largest stride (multiple of 8, <= %u)",
                      SK_TILE_MAX_STRIDE);
            return SK_EINVAL;
        }
    } else if (!b->offsets) {
        if (b->stride == 0 || (!b->lengths && b->read_len > b->stride)) {
            set_error(ctx, "fixed-stride batch: need stride >= read_len > 0");
            return SK_EINVAL;
        }
        if (b->read_len > SK_MAX_READ_LEN) return SK_EINVAL;
    }
    if (b->n_reads >= (1ull << 32)) {
        set_error(ctx, "more than 2^32-1 reads in one batch");
        return SK_EINVAL;
    }
    const int32_t *k = kQualityConstants[p->qualtype];
    // Every threshold above the largest representable quality (max - offset <= 93) behaves the
    // same: no window and no char can reach it.  Clamping keeps the device integers small.
    const int32_t qthr = p->qual_threshold > 127 ? 127 : p->qual_threshold;
    a->n_reads = b->n_reads;
    a->stride = b->stride;
    a->read_len = b->read_len;
    a->qmin = k[1];
    a->qmax = k[2];
    a->craw = qthr + k[0];
    a->cthr = a->craw > 128 ? 128 : a->craw;
    a->cthr_raw = a->craw;
    a->lthr = p->length_threshold;
    a->no5 = p->no_fiveprime ? 1 : 0;
    a->truncn = p->trunc_n ? 1 : 0;
    a->n_tiles = b->tiles ? b->n_tiles : 0;
    static const int order = [] { const char *e = getenv("SK_TILE_ORDER"); return e ? atoi(e) : 0; }();
    a->tile_order = order;
    a->buf_bytes = 0;
    a->slot_order = (b->tiles && b->cuts_in_slot_order) ? 1 : 0;
    a->scan_id = 0;
    a->team_rbuf = 0;
    a->team_maxlen = 0;
    a->stream_nb = 0;
    a->stream_read_cost = 0;
    a->stream_tbl = 0;
    static const uint32_t seg_shift = [] { const char *e = getenv("SK_SEG_CHUNK_SHIFT"); return e ? (uint32_t)atoi(e) : 0u; }();
    a->seg_chunk_shift = seg_shift > 6u ? 6u : seg_shift;
    a->sort_flags = nullptr;
    a->band_table = ctx ? ctx->d_band : nullptr;
    return SK_OK;
}

// LDS bytes per wave for the tiles of a ragged batch whose reads are at most `max_len` bytes
// (0 = unknown): 64 reads + what a lane may read past its row, so that uniform 150 bp gets 10 KiB
// (16 waves per CU); tiles that do not fit go to the general kernel.
uint32_t rag_buf_bytes(uint64_t max_len, bool uniform)
{
    if (max_len == 0) return SK_RAG_BUF_DEFAULT;
    // == rag_pitch<UNIFORM>() of the kernels
    (void)uniform;
    const uint64_t pitch = 16 * (((max_len + 15) >> 4) | 1);
    uint64_t need = (64 * pitch + SK_TILE_SLACK + 15) & ~(uint64_t)15;
    if (need < 4096) need = 4096;
    if (need > SK_RAG_BUF_MAX) need = SK_RAG_BUF_MAX;
    return (uint32_t)need;
}

// which path a batch takes: 3 segmented, 1 tiled (aligned rows), 5 tiled (rows at any address),
// 2 general kernel only
// The general kernels, by the caller's longest-read hint: teams of 16 lanes per read (sk_team.hip) up to SK_STREAM_MIN
// bytes, the streaming wave-per-read kernel (sk_stream.hip) beyond and when there is no hint.  SK_GENERAL=band|team|stream
// forces one (diagnostics, tests; band = round 3's wave-per-read kernel with matrix-pipe window sums, sk_band.hip:
// parity-green, not faster than the teams, selected by nothing).  Uniform medium reads do not come here: wide_takes().
constexpr uint64_t SK_STREAM_MIN_DEFAULT = 4096;
hipError_t launch_general(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets, const uint32_t *lengths, sk_cut_dev *out,
                          unsigned long long *errword, const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream)
{
    const char *force = getenv("SK_GENERAL");
    static const uint64_t stream_min = [] { const char *e = getenv("SK_STREAM_MIN"); return e ? (uint64_t)atoll(e) : SK_STREAM_MIN_DEFAULT; }();
    const bool use_stream = force ? force[0] == 's' : (max_len == 0 || max_len > stream_min);
    if (!use_stream && !(force && force[0] == 'b')) return sk_launch_team(qual, seq, offsets, lengths, out, errword, a, max_len, cu_count, stream);
    return use_stream ? sk_launch_stream(qual, seq, offsets, lengths, out, errword, a, max_len, cu_count, stream)
                      : sk_launch_band(qual, seq, offsets, lengths, out, errword, a, max_len, cu_count, stream);
}

// uniform medium reads (rows beyond the 64-read tiles, windows of 32 and more, two waves' images to a CU): the tile kernel
// with 32-read tiles (sk_kernels.hip, WIDE).  SK_GENERAL=team|band|stream sends them to a general kernel instead (A/B runs,
// tests), SK_WIDE_MAX bounds the lengths it takes.
bool wide_takes(const sk_batch *b)
{
    if (b->offsets || b->lengths || b->tiles || getenv("SK_GENERAL")) return false;
    static const uint32_t wide_max = [] { const char *e = getenv("SK_WIDE_MAX"); return e ? (uint32_t)atoi(e) : 2200u; }();
    return b->read_len <= wide_max && b->stride < (1u << 24) && sk_wide_lds_bytes(b->read_len) != 0; // (the loader's 24-bit row offsets)
}

int path_of(const sk_batch *b)
{
    if (b->tiles) return 3;
    if (tile_eligible(b)) return 1;
    if (!b->offsets && !b->lengths) return (b->stride <= SK_TILE_MAX_STRIDE) ? 5 : 2;
    return 5; // ragged: tiles that fit a wave's LDS buffer by the tile kernel, the others by the general one
}

bool tile_eligible(const sk_batch *b)
{
    return !b->offsets && !b->tiles && b->stride % 8 == 0 && b->stride >= 8 && b->stride <= SK_TILE_MAX_STRIDE &&
           (reinterpret_cast<uintptr_t>(b->qual) & 15) == 0 && (!b->seq || (reinterpret_cast<uintptr_t>(b->seq) & 15) == 0);
}

// enqueue the kernel.  All pointers are device pointers.  The error word is "no error" on
// entry: it is reset when the context is created and again by whoever reads it
// (reset_error_word), so nothing but the kernel sits on the launch path.
// rag_fit (batches with `offsets` or `lengths`): does any 64-read tile fit the tile kernel's LDS buffer?  1 yes, 0 no
// (the general kernel takes the whole batch in spans of equal cost), -1 not known: sk_submit counts on the host,
// a device-resident batch is taken for a long-read batch when the caller's longest-read hint is beyond
// SK_LONG_BATCH_HINT -- 64 CONSECUTIVE reads of at most 2 040 bases do not occur in one.
#define SK_LONG_BATCH_HINT 4096u
// the longest read whose 64-row image fits a wave buffer of `buf` bytes (== rag_tile_fits of the kernels)
uint32_t rag_fit_len(uint32_t buf)
{
    uint32_t best = 0;
    for (uint32_t L = 16; L <= SK_RAG_MAX_LEN; L += 16)
        if (64u * 16u * ((L >> 4) | 1u) + SK_TILE_SLACK <= buf) best = L;
    return best;
}

// scratch for a regrouping of n reads; false (and no sorting) if it cannot be had
bool ensure_sort(sk_ctx *ctx, SortScratch &s, uint64_t n)
{
    // per list: the tiles of every eighth window, at most SK_SORT_WINDOW / 64 full ones + one partial one per class each;
    // 32 bytes of descriptor and 64 x 8 bytes of entries per tile (~9 bytes per read as the lists fill, 13 at worst)
    const size_t windows = (size_t)((n + SK_SORT_WINDOW - 1) / SK_SORT_WINDOW);
    const size_t per_list = ((windows + 7) / 8) * (SK_SORT_WINDOW / 64 + 64) + 8;
    if (hipSetDevice(ctx->device) != hipSuccess) return false;
    if (!s.counts) {
        if (hipMalloc(&s.counts, 32 * sizeof(uint32_t)) != hipSuccess) return false;
        if (hipMemset(s.counts, 0, 32 * sizeof(uint32_t)) != hipSuccess) return false; // synchronous: done before any scan is enqueued
        s.turn = 0;
    }
    if (per_list > s.cap_list) {
        s.release_lists();
        const size_t cap = per_list + (per_list >> 3);
        if (hipMalloc(&s.lists, 8 * cap * 4 * sizeof(unsigned long long)) != hipSuccess) return false;
        if (hipMalloc(&s.perm, 8 * cap * 64 * sizeof(uint64_t)) != hipSuccess) {
            s.release_lists();
            return false;
        }
        s.cap_list = cap;
    }
    return true;
}

int enqueue_scan(sk_ctx *ctx, const sk_scan_args *a, const sk_batch *b, sk_cut_dev *out, unsigned long long *d_err,
                 hipStream_t stream, int rag_fit = -1, SortScratch *sort = nullptr)
{
    if (a->n_reads == 0) return SK_OK;
    const uint8_t *seq = a->truncn ? b->seq : nullptr;
    const int path = path_of(b);
    if (path == 3) {
        SK_HIP(ctx, sk_launch_seg(b->qual, seq, reinterpret_cast<const sk_tile_dev *>(b->tiles), b->out_index, out, d_err, a,
                                  b->classes, b->n_classes, ctx->cu_count, stream));
    } else if (path == 1) {
        SK_HIP(ctx, sk_launch_tile(b->qual, seq, b->lengths, out, d_err, a, ctx->cu_count, stream));
    } else if (path == 5) {
        sk_scan_args ar = *a;
        const bool ragged = b->offsets || b->lengths;
        if (ragged && (rag_fit == 0 || (rag_fit < 0 && b->offsets && b->stride > SK_LONG_BATCH_HINT))) {
            SK_HIP(ctx, launch_general(b->qual, seq, b->offsets, b->lengths, out, d_err, a, b->stride, ctx->cu_count, stream));
            return SK_OK;
        }
        // the scans of a context are numbered upwards: the tile kernel leaves the number in the word after
        // the error word when it skips a tile, the general kernel returns at once if it does not find it
        ar.scan_id = ++ctx->scan_counter;
        // b->stride of an `offsets` batch is the caller's hint of the longest read (0 = unknown); with
        // stride + lengths it bounds the reads; packed uniform batches: the read length is known
        ar.buf_bytes = rag_buf_bytes(ragged ? b->stride : b->read_len, !ragged);
        // A big `offsets` batch is regrouped first (sk_sort.hip: windows of 8192 reads counting-sorted by window width,
        // 8 bytes of scratch per read).  If that finds reads of different lengths -- and none too long for the tiles --
        // the sorted scan below takes the batch on the matrix path and the two kernels after it return at once; a
        // batch of one length (or with long reads) is theirs as before and the sorted scan returns.  SK_SORT=0: never.
        static const bool sort_on = [] { const char *e = getenv("SK_SORT"); return !(e && *e == '0'); }();
        static const uint64_t sort_min = [] { const char *e = getenv("SK_SORT_MIN"); return e ? (uint64_t)atoll(e) : (uint64_t)SK_SORT_MIN_READS; }(); // (tests lower it)
        uint32_t *sort_counts = nullptr;
        const bool sorted = sort_on && sort && b->offsets && a->n_reads >= sort_min && ensure_sort(ctx, *sort, a->n_reads);
        if (sorted) {
            uint32_t fit = rag_fit_len(ar.buf_bytes);
            if (b->stride && b->stride < fit) fit = b->stride;
            sort_counts = sort->counts + 16u * (sort->turn & 1u);
            SK_HIP(ctx, sk_launch_sort(b->offsets, a->n_reads, fit, sort->perm, sort->lists, (uint32_t)sort->cap_list, sort_counts,
                                       sort->counts + 16u * (~sort->turn & 1u), stream));
            ++sort->turn;
            ar.sort_flags = sort_counts + 8;
        }
        SK_HIP(ctx, sk_launch_any(b->qual, seq, b->offsets, b->lengths, out, d_err, &ar, ctx->cu_count, stream));
        if (sorted) {
            sk_scan_args as = ar;
            as.n_tiles = (uint32_t)sort->cap_list;
            SK_HIP(ctx, sk_launch_sorted(b->qual, seq, b->offsets, sort->perm, sort->lists, sort_counts, out, d_err, &as, ctx->cu_count, stream));
        }
        // the tiles that kernel leaves: those whose reads are too long for a wave's buffer (none in a
        // packed uniform batch)
        if (ragged) SK_HIP(ctx, launch_general(b->qual, seq, b->offsets, b->lengths, out, d_err, &ar, b->stride, ctx->cu_count, stream));
    } else if (wide_takes(b)) {
        SK_HIP(ctx, sk_launch_wide(b->qual, seq, out, d_err, a, ctx->cu_count, stream));
    } else {
        SK_HIP(ctx, launch_general(b->qual, seq, b->offsets, b->lengths, out, d_err, a, b->read_len, ctx->cu_count, stream));
    }
    return SK_OK;
}

int err_word_of(sk_ctx *ctx, hipStream_t stream, unsigned long long **d, unsigned long long **h, SortScratch **sort)
{
    if (stream == nullptr) {
        *d = ctx->d_err;
        *h = ctx->h_err;
        if (sort) *sort = &ctx->sort0;
        return SK_OK;
    }
    std::lock_guard<std::mutex> lock(ctx->stream_err_lock);
    sk_ctx::ErrWord &w = ctx->stream_err[stream];
    if (!w.d) {
        SK_HIP(ctx, hipSetDevice(ctx->device));
        SK_HIP(ctx, hipMalloc(&w.d, 8 * sizeof(unsigned long long)));
        SK_HIP(ctx, hipMemset(w.d, 0xff, sizeof(unsigned long long))); // synchronous: done before any scan is enqueued
        SK_HIP(ctx, hipMemset(w.d + 1, 0, 7 * sizeof(unsigned long long)));
        SK_HIP(ctx, hipHostMalloc(&w.h, 6 * sizeof(unsigned long long), hipHostMallocDefault));
        *w.h = kNoError;
    }
    *d = w.d;
    *h = w.h;
    if (sort) *sort = &w.sort;
    return SK_OK;
}

int reset_error_word(sk_ctx *ctx, unsigned long long *d_err, hipStream_t stream)
{
    SK_HIP(ctx, hipMemsetAsync(d_err, 0xff, sizeof(unsigned long long), stream));
    return SK_OK;
}

int decode_error(unsigned long long word, sk_err *err)
{
    if (word == kNoError) return SK_OK;
    if (err) {
        err->read = (uint32_t)(word >> 32);
        err->pos = (uint32_t)((word >> 8) & 0xffffffu);
        err->ch = (int32_t)(int8_t)(word & 0xff);
    }
    return SK_ERANGE;
}

size_t batch_bytes(const sk_batch *b)
{
    // bytes of qual (and seq) the batch spans on the host
    if (b->n_reads == 0) return 0;
    if (b->tiles) {
        const sk_tile &last = b->tiles[b->n_tiles - 1];
        return (size_t)last.byte_off + (size_t)last.rows * last.stride;
    }
    if (b->offsets) return (size_t)b->offsets[b->n_reads];
    return (size_t)b->n_reads * b->stride;
}

int grow_slot(sk_ctx *ctx, Slot &s, size_t bytes, size_t reads, bool need_seq, bool need_off, bool need_len)
{
    const size_t pad = 256; // the tiled kernel's last 16-byte chunk never crosses this
    if (bytes + pad > s.cap_bytes || (need_seq && !s.d_seq)) {
        size_t cap = bytes + pad > s.cap_bytes ? (bytes + pad) + (bytes >> 3) : s.cap_bytes;
        if (s.d_qual) (void)hipFree(s.d_qual);
        if (s.d_seq) (void)hipFree(s.d_seq);
        s.d_qual = s.d_seq = nullptr;
        s.cap_bytes = 0;
        SK_HIP(ctx, hipMalloc(&s.d_qual, cap));
        if (need_seq) SK_HIP(ctx, hipMalloc(&s.d_seq, cap));
        s.cap_bytes = cap;
    }
    if (reads + 1 > s.cap_reads || (need_off && !s.d_offsets) || (need_len && !s.d_lengths)) {
        size_t cap = reads + 1 > s.cap_reads ? (reads + 1) + (reads >> 3) : s.cap_reads;
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_offsets) (void)hipFree(s.d_offsets);
        if (s.d_lengths) (void)hipFree(s.d_lengths);
        s.d_out = nullptr;
        s.d_offsets = nullptr;
        s.d_lengths = nullptr;
        s.cap_reads = 0;
        SK_HIP(ctx, hipMalloc(&s.d_out, cap * sizeof(sk_cut_dev)));
        if (need_off) SK_HIP(ctx, hipMalloc(&s.d_offsets, cap * sizeof(uint64_t)));
        if (need_len) SK_HIP(ctx, hipMalloc(&s.d_lengths, cap * sizeof(uint32_t)));
        s.cap_reads = cap;
    }
    return SK_OK;
}

} // namespace

extern "C" {

const int32_t *sk_quality_constants(int32_t qualtype)
{
    return (qualtype < 0 || qualtype > 3) ? nullptr : kQualityConstants[qualtype];
}

const char *sk_typename(int32_t qualtype) { return (qualtype < 0 || qualtype > 3) ? nullptr : kTypeNames[qualtype]; }

int sk_abi_version(void) { return SK_ABI_VERSION; }

int sk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sk_create(int device, int slots, sk_ctx **out)
{
    if (!out) return SK_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SK_ENODEV;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return SK_ENODEV;
    }
    if (device >= n) return SK_ENODEV;
    if (slots < 1) slots = 1;
    if (slots > 16) slots = 16;
    sk_ctx *ctx = new (std::nothrow) sk_ctx();
    if (!ctx) return SK_EINVAL;
    ctx->device = device;
    auto fail = [&](int rc) {
        fprintf(stderr, "sickle_amd: sk_create failed: %s\n", ctx->last_error);
        sk_destroy(ctx);
        return rc;
    };
#define SK_TRY(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_error(ctx, "%s failed: %s", #call, hipGetErrorString(e_));              \
            return fail(SK_EHIP);                                                       \
        }                                                                               \
    } while (0)
    SK_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    SK_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(ctx, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        return fail(SK_ENODEV);
    }
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    SK_TRY(hipStreamCreateWithFlags(&ctx->compute, hipStreamNonBlocking));
    SK_TRY(hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    {
        std::vector<uint32_t> band(SK_BAND_TABLE_DWORDS);
        for (uint32_t w = 0; w < SK_BAND_WIDTHS; ++w)
            for (int b = 0; b < 3; ++b)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j)
                        band[((w * 3u + (uint32_t)b) * 64u + (uint32_t)lane) * 4u + (uint32_t)j] = sk_band_dword(lane, (int)w, 16 * (lane >> 5) + 4 * j + 32 * b);
        SK_TRY(hipMalloc(&ctx->d_band, band.size() * sizeof(uint32_t)));
        SK_TRY(hipMemcpy(ctx->d_band, band.data(), band.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    SK_TRY(hipMalloc(&ctx->d_err, 8 * sizeof(unsigned long long)));
    SK_TRY(hipMemset(ctx->d_err, 0xff, sizeof(unsigned long long)));
    SK_TRY(hipMemset(ctx->d_err + 1, 0, 7 * sizeof(unsigned long long))); // the hand-over word (see enqueue_scan), four pair counters, tiles left by the tile kernel
    SK_TRY(hipHostMalloc(&ctx->h_err, 6 * sizeof(unsigned long long), hipHostMallocDefault));
    *ctx->h_err = kNoError;
    ctx->slots.resize((size_t)slots);
    for (Slot &s : ctx->slots) {
        SK_TRY(hipMalloc(&s.d_err, 8 * sizeof(unsigned long long))); // the same block as the context's (words 2..5 unused)
        SK_TRY(hipMemset(s.d_err, 0xff, sizeof(unsigned long long)));
        SK_TRY(hipMemset(s.d_err + 1, 0, 7 * sizeof(unsigned long long)));
        SK_TRY(hipHostMalloc(&s.h_err, sizeof(unsigned long long), hipHostMallocDefault));
        SK_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        SK_TRY(hipEventCreateWithFlags(&s.finished, hipEventDisableTiming));
    }
#undef SK_TRY
    *out = ctx;
    return SK_OK;
}

void sk_destroy(sk_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->compute) (void)hipStreamSynchronize(ctx->compute);
    if (ctx->copy) (void)hipStreamSynchronize(ctx->copy);
    for (Slot &s : ctx->slots) {
        if (s.d_qual) (void)hipFree(s.d_qual);
        if (s.d_seq) (void)hipFree(s.d_seq);
        if (s.d_offsets) (void)hipFree(s.d_offsets);
        if (s.d_lengths) (void)hipFree(s.d_lengths);
        if (s.d_tiles) (void)hipFree(s.d_tiles);
        if (s.d_out_index) (void)hipFree(s.d_out_index);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_err) (void)hipFree(s.d_err);
        if (s.h_err) (void)hipHostFree(s.h_err);
        s.sort.release();
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.finished) (void)hipEventDestroy(s.finished);
    }
    if (ctx->d_band) (void)hipFree(ctx->d_band);
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    if (ctx->h_err) (void)hipHostFree(ctx->h_err);
    for (auto &kv : ctx->stream_err) {
        if (kv.second.d) (void)hipFree(kv.second.d);
        if (kv.second.h) (void)hipHostFree(kv.second.h);
        kv.second.sort.release();
    }
    ctx->sort0.release();
    if (ctx->compute) (void)hipStreamDestroy(ctx->compute);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    delete ctx;
}

const char *sk_last_error(const sk_ctx *ctx) { return ctx ? ctx->last_error : "no context"; }
int sk_device(const sk_ctx *ctx) { return ctx ? ctx->device : -1; }

void *sk_host_alloc(sk_ctx *ctx, size_t bytes)
{
    if (!ctx) return nullptr;
    void *p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        set_error(ctx, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void sk_host_free(sk_ctx *ctx, void *p)
{
    (void)ctx;
    if (p) (void)hipHostFree(p);
}

int sk_kernel_for(const sk_batch *batch)
{
    if (!batch) return 0;
    if (batch->tiles) return 3;
    if (!tile_eligible(batch)) {
        const int path = path_of(batch);
        const bool ragged = batch->offsets || batch->lengths;
        const uint64_t longest = ragged ? batch->stride : batch->read_len;
        const bool general_only = path == 2 || (path == 5 && batch->offsets && batch->stride > SK_LONG_BATCH_HINT);
        if (path == 2 && wide_takes(batch)) return 8;
        return general_only ? ((longest == 0 || longest > SK_STREAM_MIN_DEFAULT) ? 6 : 2) : path;
    }
    // uniform batches without a sequence buffer (no -n) and rows of 72..160 bytes: the tile comes in
    // through the wave's registers instead of LDS-DMA
    if (!batch->lengths && !batch->seq && sk_tile_is_staged(batch->stride, batch->read_len, 0)) return 4;
    return 1;
}

uint32_t sk_seg_classes(const sk_tile *tiles, uint32_t n_tiles, sk_seg_class *out, uint32_t max_classes)
{
    if (!tiles || !out || n_tiles == 0 || max_classes == 0) return 0;
    // occupancy class of a tile = single-wave workgroups per CU its LDS buffer allows (16 down to 4)
    auto per_cu = [](uint32_t stride) {
        const uint32_t lds = 64u * stride + SK_TILE_SLACK;
        const uint32_t n = lds ? SK_LDS_PER_CU / lds : 16u;
        return n > 16u ? 16u : n;
    };
    auto wide = [](const sk_tile &t) -> uint32_t { return (uint32_t)(t.read_len / 10) > 33u; };
    // pass 1: maximal runs of equal (occupancy, wide)
    std::vector<sk_seg_class> runs;
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t pc = per_cu(tiles[t].stride), w = wide(tiles[t]);
        if (!runs.empty() && per_cu(runs.back().max_stride) == pc && runs.back().wide == w) {
            runs.back().n_tiles++;
            if (tiles[t].stride > runs.back().max_stride) runs.back().max_stride = tiles[t].stride;
        } else {
            if (runs.size() >= 4096) return 0; // not a sorted batch
            runs.push_back(sk_seg_class{t, 1u, tiles[t].stride, w});
        }
    }
    // pass 2: runs are gathered into a class until it can fill the device a few times over; then the
    // next run of a different kind opens a new class (a class takes the widest stride and the wide
    // loop of its runs); a short last class joins the one before it
    static const uint32_t min_tiles = [] { const char *e = getenv("SK_SEG_MIN_TILES"); return e ? (uint32_t)atoi(e) : 65536u; }();
    std::vector<sk_seg_class> merged;
    auto join = [](sk_seg_class &m, const sk_seg_class &r) {
        m.n_tiles += r.n_tiles;
        if (r.max_stride > m.max_stride) m.max_stride = r.max_stride;
        m.wide |= r.wide;
    };
    for (const sk_seg_class &r : runs) {
        if (!merged.empty() && merged.back().n_tiles < min_tiles) join(merged.back(), r);
        else merged.push_back(r);
    }
    if (merged.size() > 1 && merged.back().n_tiles < min_tiles) {
        const sk_seg_class last = merged.back();
        merged.pop_back();
        join(merged.back(), last);
    }
    if (merged.size() > max_classes) return 0;
    for (size_t i = 0; i < merged.size(); ++i) out[i] = merged[i];
    return (uint32_t)merged.size();
}

const char *sk_kernel_name(int which)
{
    switch (which) {
    case 1: return "sk_scan_tile_kernel";
    case 2: return "sk_scan_team_kernel";
    case 3: return "sk_scan_tile_kernel";
    case 4: return "sk_scan_tile_staged_kernel";
    case 5: return "sk_scan_tile_any_kernel";
    case 6: return "sk_scan_stream_kernel";
    case 7: return "sk_scan_band_kernel";
    case 8: return "sk_scan_tile_wide_kernel";
    default: return "";
    }
}

int sk_count_pairs_device_async(sk_ctx *ctx, const sk_cut *cuts, uint64_t n_pairs, uint8_t *classes, void *hip_stream)
{
    if (!ctx || (!cuts && n_pairs) || (reinterpret_cast<uintptr_t>(cuts) & 15)) return SK_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    unsigned long long *d, *h;
    int rc = err_word_of(ctx, stream, &d, &h);
    if (rc != SK_OK) return rc;
    SK_HIP(ctx, sk_launch_pair_count(reinterpret_cast<const sk_cut_dev *>(cuts), n_pairs, classes, d + 2, ctx->cu_count, stream));
    return SK_OK;
}

int sk_count_pairs_device_finish(sk_ctx *ctx, void *hip_stream, sk_pair_counts *counts)
{
    if (!ctx || !counts) return SK_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    unsigned long long *d, *h;
    int rc = err_word_of(ctx, stream, &d, &h);
    if (rc != SK_OK) return rc;
    SK_HIP(ctx, hipMemcpyAsync(h + 2, d + 2, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    SK_HIP(ctx, hipMemsetAsync(d + 2, 0, 4 * sizeof(unsigned long long), stream));
    SK_HIP(ctx, hipStreamSynchronize(stream));
    counts->both = h[2];
    counts->only_first = h[3];
    counts->only_second = h[4];
    counts->none = h[5];
    return SK_OK;
}

int sk_probe_read_bandwidth(sk_ctx *ctx, const void *dev_buf, size_t bytes, int launches, void *hip_stream, double *gb_per_s)
{
    if (!ctx || !dev_buf || !gb_per_s || launches < 1 || bytes < (1u << 20) || (reinterpret_cast<uintptr_t>(dev_buf) & 15)) return SK_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    SK_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0, e1;
    SK_HIP(ctx, hipEventCreate(&e0));
    SK_HIP(ctx, hipEventCreate(&e1));
    uint32_t *sink = reinterpret_cast<uint32_t *>(ctx->d_err); // never written: the kernel's store is unreachable for real data
    for (int i = 0; i < 3; ++i) SK_HIP(ctx, sk_launch_read_probe(dev_buf, bytes, sink, ctx->cu_count, stream));
    SK_HIP(ctx, hipEventRecord(e0, stream));
    for (int i = 0; i < launches; ++i) SK_HIP(ctx, sk_launch_read_probe(dev_buf, bytes, sink, ctx->cu_count, stream));
    SK_HIP(ctx, hipEventRecord(e1, stream));
    SK_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0;
    SK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gb_per_s = ms > 0 ? (double)bytes * launches / (ms * 1e-3) / 1e9 : 0.0;
    return SK_OK;
}

int sk_scan_device_async(sk_ctx *ctx, const sk_params *params, const sk_batch *batch, sk_cut *out, void *hip_stream)
{
    if (!ctx || !out) return SK_EINVAL;
    sk_scan_args a;
    int rc = make_args(ctx, params, batch, &a);
    if (rc != SK_OK) return rc;
    // NULL is HIP's default (null) stream, like everywhere else in HIP: ordered after whatever the
    // caller queued there (e.g. the kernels that produced the batch)
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    unsigned long long *d_err, *h_err;
    SortScratch *sort = nullptr;
    rc = err_word_of(ctx, stream, &d_err, &h_err, &sort);
    if (rc != SK_OK) return rc;
    return enqueue_scan(ctx, &a, batch, reinterpret_cast<sk_cut_dev *>(out), d_err, stream, -1, sort);
}

int sk_scan_device_finish(sk_ctx *ctx, void *hip_stream, sk_err *err)
{
    if (!ctx) return SK_EINVAL;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    unsigned long long *d_err, *h_err;
    int rc = err_word_of(ctx, stream, &d_err, &h_err);
    if (rc != SK_OK) return rc;
    SK_HIP(ctx, hipMemcpyAsync(h_err, d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    rc = reset_error_word(ctx, d_err, stream);
    if (rc != SK_OK) return rc;
    SK_HIP(ctx, hipStreamSynchronize(stream));
    return decode_error(*h_err, err);
}

int sk_submit(sk_ctx *ctx, int slot, const sk_params *params, const sk_batch *batch, sk_cut *out)
{
    if (!ctx || slot < 0 || (size_t)slot >= ctx->slots.size() || !batch || (!out && batch->n_reads)) return SK_EINVAL;
    Slot &s = ctx->slots[(size_t)slot];
    if (s.busy) return SK_EBUSY;
    sk_scan_args a;
    int rc = make_args(ctx, params, batch, &a);
    if (rc != SK_OK) return rc;
    SK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = batch_bytes(batch);
    const size_t n = (size_t)batch->n_reads;
    if (!batch->offsets && batch->lengths) {
        for (size_t r = 0; r < n; ++r)
            if (batch->lengths[r] > batch->stride) {
                set_error(ctx, "lengths[%zu] = %u exceeds stride %u", r, batch->lengths[r], batch->stride);
                return SK_EINVAL;
            }
    }
    uint64_t rag_max_len = 0;
    int rag_fit = -1;
    if (batch->offsets) {
        // host offsets are checked here (ascending, reads within SK_MAX_READ_LEN: the error word keeps
        // 24 bits of position) and give the exact longest read, which sizes the tile kernel's buffers
        for (size_t r = 0; r < n; ++r) {
            const uint64_t o = batch->offsets[r], e = batch->offsets[r + 1];
            if (e < o || e - o > SK_MAX_READ_LEN) {
                set_error(ctx, "offsets[%zu..%zu] = %llu, %llu: not ascending, or a read longer than %u", r, r + 1,
                          (unsigned long long)o, (unsigned long long)e, SK_MAX_READ_LEN);
                return SK_EINVAL;
            }
            if (e - o > rag_max_len) rag_max_len = e - o;
        }
        // does any tile of 64 consecutive reads fit the tile kernel's buffer (rag_tile_fits of the kernels)?
        const uint32_t buf = rag_buf_bytes(rag_max_len, false);
        rag_fit = 0;
        for (size_t t = 0; t < n && !rag_fit; t += 64) {
            uint64_t lmax = 0;
            for (size_t r = t; r < n && r < t + 64; ++r) lmax = std::max<uint64_t>(lmax, batch->offsets[r + 1] - batch->offsets[r]);
            if (lmax <= SK_RAG_MAX_LEN && 64 * 16 * (((lmax + 15) >> 4) | 1) + SK_TILE_SLACK <= buf) rag_fit = 1;
        }
    }
    rc = grow_slot(ctx, s, bytes, n, a.truncn != 0, batch->offsets != nullptr, batch->lengths != nullptr);
    if (rc != SK_OK) return rc;
    if (batch->tiles) {
        // descriptors are checked here, on the host, before any of them reaches a kernel
        uint64_t reads = 0;
        for (uint32_t t = 0; t < batch->n_tiles; ++t) {
            const sk_tile &d = batch->tiles[t];
            if (d.rows == 0 || d.rows > 64 || d.stride % 8 != 0 || d.stride > batch->stride || d.read_len > d.stride ||
                d.byte_off % 16 != 0 || (size_t)d.byte_off + (size_t)d.rows * d.stride > bytes ||
                (uint64_t)d.slot0 + d.rows > batch->n_reads) {
                set_error(ctx, "segmented batch: tile %u is malformed", t);
                return SK_EINVAL;
            }
            reads += d.rows;
        }
        if (reads != batch->n_reads) {
            set_error(ctx, "segmented batch: tiles hold %llu reads, n_reads is %llu", (unsigned long long)reads,
                      (unsigned long long)batch->n_reads);
            return SK_EINVAL;
        }
        for (size_t i = 0; i < n; ++i)
            if (batch->out_index[i] >= batch->n_reads) {
                set_error(ctx, "segmented batch: out_index[%zu] out of range", i);
                return SK_EINVAL;
            }
        if (batch->n_tiles > s.cap_tiles) {
            if (s.d_tiles) (void)hipFree(s.d_tiles);
            s.d_tiles = nullptr;
            s.cap_tiles = 0;
            const size_t cap = batch->n_tiles + (batch->n_tiles >> 3) + 16;
            SK_HIP(ctx, hipMalloc(&s.d_tiles, cap * sizeof(sk_tile_dev)));
            s.cap_tiles = cap;
        }
        if (n > s.cap_index) {
            if (s.d_out_index) (void)hipFree(s.d_out_index);
            s.d_out_index = nullptr;
            s.cap_index = 0;
            const size_t cap = n + (n >> 3) + 16;
            SK_HIP(ctx, hipMalloc(&s.d_out_index, cap * sizeof(uint32_t)));
            s.cap_index = cap;
        }
        if (!batch->classes) {
            s.classes.resize(SK_SEG_MAX_CLASSES);
            s.classes.resize(sk_seg_classes(batch->tiles, batch->n_tiles, s.classes.data(), SK_SEG_MAX_CLASSES));
        }
        SK_HIP(ctx, hipMemcpyAsync(s.d_tiles, batch->tiles, batch->n_tiles * sizeof(sk_tile), hipMemcpyHostToDevice, ctx->copy));
        SK_HIP(ctx, hipMemcpyAsync(s.d_out_index, batch->out_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->copy));
    }

    // H2D on the copy stream (overlaps the previous slot's kernel on the compute stream)
    if (bytes) {
        SK_HIP(ctx, hipMemcpyAsync(s.d_qual, batch->qual, bytes, hipMemcpyHostToDevice, ctx->copy));
        if (a.truncn) SK_HIP(ctx, hipMemcpyAsync(s.d_seq, batch->seq, bytes, hipMemcpyHostToDevice, ctx->copy));
    }
    if (batch->offsets)
        SK_HIP(ctx, hipMemcpyAsync(s.d_offsets, batch->offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->copy));
    if (!batch->offsets && batch->lengths && n)
        SK_HIP(ctx, hipMemcpyAsync(s.d_lengths, batch->lengths, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->copy));
    SK_HIP(ctx, hipEventRecord(s.copied, ctx->copy));
    SK_HIP(ctx, hipStreamWaitEvent(ctx->compute, s.copied, 0));

    sk_batch dev = *batch;
    dev.qual = s.d_qual;
    dev.seq = a.truncn ? s.d_seq : nullptr;
    dev.offsets = batch->offsets ? s.d_offsets : nullptr;
    if (batch->offsets) dev.stride = (uint32_t)rag_max_len;
    dev.lengths = (!batch->offsets && batch->lengths) ? s.d_lengths : nullptr;
    dev.tiles = batch->tiles ? reinterpret_cast<const sk_tile *>(s.d_tiles) : nullptr;
    dev.out_index = batch->tiles ? s.d_out_index : nullptr;
    if (batch->tiles && !batch->classes && !s.classes.empty()) {
        dev.classes = s.classes.data();
        dev.n_classes = (uint32_t)s.classes.size();
    }
    rc = enqueue_scan(ctx, &a, &dev, s.d_out, s.d_err, ctx->compute, rag_fit, &s.sort);
    if (rc != SK_OK) return rc;
    if (n) SK_HIP(ctx, hipMemcpyAsync(out, s.d_out, n * sizeof(sk_cut_dev), hipMemcpyDeviceToHost, ctx->compute));
    SK_HIP(ctx, hipMemcpyAsync(s.h_err, s.d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->compute));
    rc = reset_error_word(ctx, s.d_err, ctx->compute);
    if (rc != SK_OK) return rc;
    SK_HIP(ctx, hipEventRecord(s.finished, ctx->compute));
    s.busy = true;
    return SK_OK;
}

int sk_wait(sk_ctx *ctx, int slot, sk_err *err)
{
    if (!ctx || slot < 0 || (size_t)slot >= ctx->slots.size()) return SK_EINVAL;
    Slot &s = ctx->slots[(size_t)slot];
    if (!s.busy) return SK_EINVAL;
    SK_HIP(ctx, hipEventSynchronize(s.finished));
    s.busy = false;
    return decode_error(*s.h_err, err);
}

int sk_trim_batch(sk_ctx *ctx, const sk_params *params, const sk_batch *batch, sk_cut *out, sk_err *err)
{
    int rc = sk_submit(ctx, 0, params, batch, out);
    if (rc != SK_OK) return rc;
    return sk_wait(ctx, 0, err);
}

} // extern "C"
