/*
 * sickle_amd.h -- C ABI of libsickle_amd.so: the MI355X (gfx950) replacement for the
 * per-batch quality scan of pentalpha/sickle.
 *
 * The reference has no FFI; the seam this ABI fills is what its worker threads do for
 * one batch:
 *   Trim_Single::processing_thread   reference src/trim_single.cpp:357-372
 *   Trim_Paired::processing_thread   reference src/trim_paired.cpp:483-504
 * i.e. `saved_cutsites[i] = sliding_window(*queue[i])` for every read of the batch, with
 *   Abstract_Trimmer::sliding_window  reference src/trim.cpp:3-116
 *   Abstract_Trimmer::get_quality_num reference src/trim.cpp:118-140
 * Conventions kept from the reference: the result per read is its `cutsites` pair
 * (src/sickle.h:93-96), a read is kept iff three >= 0 (src/trim_single.cpp:368), and a
 * quality character outside the encoding's range is fatal (src/trim.cpp:129-137).
 * Conventions changed because this is a library: the caller owns every buffer, nothing
 * is malloc'd per read, and the range error is RETURNED (sk_err) instead of exit(1) --
 * the host pipeline prints the reference's message and exits.
 *
 * There is no CPU fallback: every entry point that computes runs the HIP kernels and
 * fails (SK_ENODEV / SK_EHIP) when no gfx950 device is usable.
 *
 * Batch layout (struct-of-arrays, packed by the ingest side):
 *   ragged      : offsets != NULL; read r = bytes [offsets[r], offsets[r+1]) of qual (and seq),
 *                 offsets ascending.  batch->stride is then a HINT: the longest read of the batch
 *                 (0 = unknown), which sizes the kernel's LDS tiles (sk_submit / sk_trim_batch see
 *                 the offsets and work it out themselves).  (ABI 1 ignored `stride` on such batches: a
 *                 caller that leaves a stale value there gets the right cuts from a slower kernel --
 *                 above 4096 the batch goes to the long-read kernel whole.  Set it to 0 or to the truth.)
 *                 A batch of 65 536 reads or more whose reads differ in length is regrouped on the
 *                 device first (windows of 8192 reads counting-sorted by window width, ~9 bytes of
 *                 scratch per read owned by the context, per stream): its tiles then hold reads of one
 *                 window width and take the matrix path; batches of one length and batches with reads
 *                 too long for a tile are scanned as they lie.  Cuts and errors keep the caller's numbering.  Tiles of 64 consecutive reads are
 *                 re-strided on their way into LDS and scanned one lane per read; a tile whose
 *                 reads are too long for that goes to the general kernel (a wave per read, the read
 *                 streamed through LDS: any length).  A hint beyond 4096 declares a long-read batch:
 *                 the general kernel takes all of it, in spans of equal cost per wave.
 *   fixed stride: offsets == NULL; read r = bytes [r*stride, r*stride + len_r) with
 *                 len_r = lengths ? lengths[r] : read_len.  Fastest: stride % 8 == 0, stride <=
 *                 SK_TILE_MAX_STRIDE, base pointers 16-byte aligned, stride/8 ODD (152, 104, 264
 *                 ...: the rows then spread over all LDS banks; stride/8 even still works, slower).
 *                 Any other stride <= SK_TILE_MAX_STRIDE or alignment (e.g. reads packed back to
 *                 back, stride == read_len) is re-strided like a ragged batch; longer rows: one
 *                 length (lengths == NULL) of up to ~2200 bases keeps the tile kernel with 32 or 16
 *                 reads to a tile (round 3), anything else goes through the general kernels.
 *   segmented   : tiles != NULL (offsets and lengths NULL).  The caller has grouped the reads
 *                 by length: tile t holds `rows` (<= 64) reads of `read_len` bytes each at
 *                 qual[byte_off + i*stride] (stride % 8 == 0, byte_off % 16 == 0), and slot
 *                 slot0+i of out_index[] says where read i's cut goes: out[out_index[slot0+i]] (or
 *                 out[slot0+i] with cuts_in_slot_order).
 *                 Every tile is uniform inside, so mixed-length batches keep the fast tiled
 *                 kernel (matrix-pipe window sums) with no padding to the longest read.
 *                 batch->stride = the largest tile stride, n_reads = number of reads.
 * seq is only read when params->trunc_n != 0 (the N rule, src/trim.cpp:86-98) and may be
 * NULL otherwise.
 */
#ifndef SICKLE_AMD_H
#define SICKLE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_ABI_VERSION 2

/* quality_type, reference src/sickle.h:61-66 */
enum { SK_PHRED = 0, SK_SANGER = 1, SK_SOLEXA = 2, SK_ILLUMINA = 3 };

/* return codes */
enum {
    SK_OK = 0,
    SK_ERANGE = 1, /* a quality char outside the encoding's range was read: *err is filled */
    SK_EINVAL = -1,
    SK_ENODEV = -2, /* no usable gfx950 device */
    SK_EHIP = -3,   /* a HIP runtime call failed: see sk_last_error() */
    SK_EBUSY = -4   /* slot still in flight */
};

#define SK_TILE_MAX_STRIDE 512u /* two LDS buffers of 64 reads per wave must fit the 160 KiB of a CU */
/* The longest read a batch may hold: 16 Mi bases (sk_err and the device's error word keep 24 bits of position).  The
 * reference has no such limit (it has no limits at all: std::string).  sk_submit / sk_trim_batch return SK_EINVAL for
 * an `offsets` batch with a longer read, and so does a fixed-stride batch with read_len beyond it; the CLI names the
 * record and exits 1. */
#define SK_MAX_READ_LEN (1u << 24)

/* the config ints of Abstract_Trimmer, reference src/trim.h:16-20 */
typedef struct {
    int32_t qualtype;         /* SK_SANGER / SK_SOLEXA / SK_ILLUMINA (SK_PHRED accepted) */
    int32_t qual_threshold;   /* -q, >= 0 */
    int32_t length_threshold; /* -l, >= 0 */
    int32_t no_fiveprime;     /* -x */
    int32_t trunc_n;          /* -n */
} sk_params;

/* == reference `cutsites`, src/sickle.h:93-96; discarded reads are (-1,-1) */
typedef struct {
    int32_t five;
    int32_t three;
} sk_cut;

/* the first out-of-range quality char: what src/trim.cpp:130-135 prints */
typedef struct {
    uint32_t read; /* lowest erroring read index in the batch */
    uint32_t pos;  /* 0-based position in the read (the reference prints pos+1) */
    int32_t ch;    /* (int)(char) value, i.e. bytes >= 0x80 are negative */
} sk_err;

/* one tile of a segmented batch */
typedef struct {
    uint64_t byte_off; /* of the tile's first read in qual (and seq); multiple of 16 */
    uint32_t slot0;    /* index of the tile's first read in out_index[] */
    uint32_t stride;   /* bytes between the tile's reads; multiple of 8, <= SK_TILE_MAX_STRIDE */
    uint16_t rows;     /* reads in this tile, 1..64 */
    uint16_t read_len; /* their common length, <= stride */
    uint32_t reserved; /* 0 */
} sk_tile;

/* a run of tiles of a segmented batch that is launched together (optional, see sk_batch) */
typedef struct {
    uint32_t first_tile; /* index of the run's first tile */
    uint32_t n_tiles;
    uint32_t max_stride; /* the largest tile stride in the run: sizes the LDS buffer of its waves */
    uint32_t wide;       /* != 0 if any tile of the run has read_len / 10 > 33 */
} sk_seg_class;

typedef struct {
    const uint8_t *qual;
    const uint8_t *seq;      /* NULL unless trunc_n */
    const uint64_t *offsets; /* n_reads+1 entries, or NULL */
    uint32_t stride;         /* fixed-stride layout; segmented: the largest tile stride; ragged: longest read or 0 */
    uint32_t read_len;       /* fixed-stride layout with lengths == NULL */
    const uint32_t *lengths; /* fixed-stride layout, per-read lengths, or NULL */
    uint64_t n_reads;
    const sk_tile *tiles;      /* segmented layout: n_tiles descriptors, or NULL */
    uint32_t n_tiles;
    const uint32_t *out_index; /* segmented layout: n_reads entries */
    /* segmented layout, optional: the tile array cut into runs that are launched one after the other,
     * each with LDS sized for its own widest row (a batch sorted by length then gives its short reads
     * full occupancy).  HOST memory, also for device-resident batches; NULL = one run (sk_submit /
     * sk_trim_batch then cut the host tiles themselves).  sk_seg_classes() fills such a table. */
    const sk_seg_class *classes;
    uint32_t n_classes;
    /* segmented layout, optional: != 0 = out[slot] takes the cut of the read in slot `slot` (tile order)
     * instead of out[out_index[slot]].  The device then writes its cuts as one coalesced stream (the
     * scattered 8-byte stores cost 22 % extra HBM traffic on a 75-301 bp mix) and the caller, who has
     * out_index, puts them in order while it consumes them.  Range errors are still reported with the
     * caller's read number (out_index is only consulted for an erroring read). */
    uint32_t cuts_in_slot_order;
} sk_batch;

typedef struct sk_ctx sk_ctx;

/* {offset, min, max} of an encoding, reference src/sickle.h:85-91; NULL if qualtype invalid */
const int32_t *sk_quality_constants(int32_t qualtype);
/* "Phred" / "Sanger" / "Solexa" / "Illumina", reference src/sickle.h:68-73 */
const char *sk_typename(int32_t qualtype);
int sk_abi_version(void);
/* number of visible HIP devices (0 when none / no driver) */
int sk_device_count(void);

/* One context per device and host thread of use; owns two streams (compute, copy) and
 * `slots` staging slots for the asynchronous host path.  device < 0 -> current device. */
int sk_create(int device, int slots, sk_ctx **out);
void sk_destroy(sk_ctx *ctx);
const char *sk_last_error(const sk_ctx *ctx);
int sk_device(const sk_ctx *ctx);

/* pinned host memory for batches and cut arrays (plain malloc'd memory also works with
 * sk_submit, only slower) */
void *sk_host_alloc(sk_ctx *ctx, size_t bytes);
void sk_host_free(sk_ctx *ctx, void *p);

/*
 * Device-resident batch: every pointer in *batch and `out` is a DEVICE pointer.  Enqueues
 * the scan on `hip_stream` (a hipStream_t; NULL = HIP's default stream, so it is ordered after
 * the work the caller queued there) and
 * returns without waiting.  out[r] is written for every read.  Range errors of the scans enqueued
 * on one stream since that stream's last sk_scan_device_finish accumulate in one device word per
 * stream (lowest read index wins), so scans on different streams of one context do not steal each
 * other's errors; nothing but the kernel(s) is enqueued here.  One host thread per context at a time.
 */
int sk_scan_device_async(sk_ctx *ctx, const sk_params *params, const sk_batch *batch,
                         sk_cut *out, void *hip_stream);
/* Waits for the stream and reports (and clears) the range error of the scans enqueued on it
 * since the previous finish: SK_OK, or SK_ERANGE with *err filled. */
int sk_scan_device_finish(sk_ctx *ctx, void *hip_stream, sk_err *err);

/*
 * Host batch, synchronous: H2D, scan, D2H.  What processing_thread does for one batch.
 * Returns SK_OK, SK_ERANGE (*err filled; `out` then holds the cuts of the reads the device
 * still scanned, the caller is expected to abort like the reference does) or < 0.
 */
int sk_trim_batch(sk_ctx *ctx, const sk_params *params, const sk_batch *batch, sk_cut *out,
                  sk_err *err);

/*
 * Host batch, asynchronous, double-buffered: sk_submit copies the batch to the device on
 * the copy stream, scans it on the compute stream and copies the cuts back into `out`;
 * sk_wait blocks until that slot is done.  The caller keeps *batch's buffers and `out`
 * alive and untouched in between.  With slots >= 2 the H2D copy of one batch overlaps the
 * scan of the previous one.
 */
int sk_submit(sk_ctx *ctx, int slot, const sk_params *params, const sk_batch *batch, sk_cut *out);
int sk_wait(sk_ctx *ctx, int slot, sk_err *err);

/* Cuts the n_tiles HOST tile descriptors into at most max_classes runs of equal occupancy class (and
 * of equal need for the wide-window matrix loop), merging runs too short to fill the device.
 * Returns the number of runs written to out (>= 1 for n_tiles > 0), or 0 when the tiles change class
 * too often for max_classes runs (pass classes = NULL then). */
uint32_t sk_seg_classes(const sk_tile *tiles, uint32_t n_tiles, sk_seg_class *out, uint32_t max_classes);

/* Which kernel a batch of this shape would use: 1 = tiled (lane per read, LDS tile by LDS-DMA),
 * 8 = uniform medium reads (fixed stride, one length of ~505 .. 2200 bases: tiles of 32 or 16 reads, a pair / four lanes
 * per read, windows of any width on the matrix path), 2 = general, medium reads (teams of 16 lanes per read, up to a
 * longest read of 4096: ragged medium reads, uniform ones beyond 8's range), 6 = general, long reads (a
 * wave per read with the read streamed through LDS), 3 = tiled over a segmented batch, 4 = tiled with the tile
 * staged through registers (equal lengths, no sequence buffer, row stride 72..320), 5 = tiled with rows re-strided
 * on the way into LDS (packed / misaligned fixed stride, ragged; a ragged batch of mixed lengths is regrouped on
 * the device first -- the device decides, so the answer stays 5).  7 = round 3's matrix-pipe wave-per-read kernel
 * for medium reads (sk_band.hip): built and parity-tested, no shape selects it (SK_GENERAL=band forces it; it is
 * not faster than 2).  For tests and bench labels. */
int sk_kernel_for(const sk_batch *batch);

/* For bench.py's roofline: name of the dominant kernel as rocprofv3 reports it */
const char *sk_kernel_name(int which);

/*
 * Pair classification on the device: reference src/trim_paired.cpp:543-567.  `cuts` (DEVICE pointer, the
 * output of a scan in read order) holds the mates of pair k at 2k and 2k+1; a mate is kept iff its three >= 0
 * (src/trim_paired.cpp:500,502).  Counts the four classes into the context (per stream, like the error word)
 * and, if classes != NULL (device pointer, n_pairs bytes), writes each pair's class: SK_PAIR_BOTH (both kept:
 * two paired records), SK_PAIR_FIRST / SK_PAIR_SECOND (one single record), SK_PAIR_NONE.  The reference's
 * six counters follow: kept_p = 2*both, kept_s1 = discard_s2 = only_first, kept_s2 = discard_s1 =
 * only_second, discard_p = 2*none.  Enqueued on hip_stream behind the scan that produced `cuts`.
 */
enum { SK_PAIR_BOTH = 0, SK_PAIR_FIRST = 1, SK_PAIR_SECOND = 2, SK_PAIR_NONE = 3 };
typedef struct {
    uint64_t both, only_first, only_second, none; /* pairs */
} sk_pair_counts;
int sk_count_pairs_device_async(sk_ctx *ctx, const sk_cut *cuts, uint64_t n_pairs, uint8_t *classes, void *hip_stream);
/* Waits for the stream, returns and clears the counts accumulated on it since the last finish. */
int sk_count_pairs_device_finish(sk_ctx *ctx, void *hip_stream, sk_pair_counts *counts);

/* Measurement aid (bench.py's second roofline denominator): streams `bytes` of device memory at
 * dev_buf through a read-only kernel (16-byte nt loads, nothing written) `launches` times on
 * hip_stream and returns the average rate in GB/s, timed with HIP events on that stream. */
int sk_probe_read_bandwidth(sk_ctx *ctx, const void *dev_buf, size_t bytes, int launches, void *hip_stream,
                            double *gb_per_s);

/* ---- BGZF block deflate for the -g writer (no counterpart in the reference, whose -g hands the
 * records to gzprintf, src/trim_single.cpp:418).  text: n_blocks blocks at a stride of 65280 bytes,
 * block b holding sizes[b] (<= 65280) bytes (the buffer may end with the last block's bytes); out: n_blocks slots of 65536 bytes; out_sizes[b] = the
 * length of block b's deflate stream in its slot, or 0 when the block does not compress into the
 * slot (the caller then writes it as a stored block).  The caller frames each stream as a gzip
 * member with the BGZF size field, CRC-32 and ISIZE.  Host pointers (pinned ones from
 * sk_bgzf_host_alloc are copied fastest); synchronous; callable from several threads. */
int sk_bgzf_deflate(int device, const uint8_t *text, const uint32_t *sizes, uint32_t n_blocks, uint8_t *out, uint32_t *out_sizes);
void *sk_bgzf_host_alloc(size_t bytes);
void sk_bgzf_host_free(void *p);
const char *sk_bgzf_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SICKLE_AMD_H */
