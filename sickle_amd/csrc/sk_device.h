// sk_device.h -- shared between the kernels (sk_kernels.hip) and the C-ABI layer (sk_capi.hip).
#ifndef SK_DEVICE_H
#define SK_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sickle_amd.h"

#define SK_TILE_THREADS 256
#define SK_TILE_WAVES (SK_TILE_THREADS / 64)
#define SK_TILE_SLACK 128u /* bytes a lane may read past its tile (the lead stream runs ahead) */
#define SK_LDS_PER_CU (160u * 1024u)
#define SK_TILE_NBUF_DEFAULT 1 /* LDS buffers per wave for the quality tile (see sk_kernels.hip) */
#define SK_MAX_READ_LEN_DEV (1u << 24) /* == SK_MAX_READ_LEN of the C ABI */
#define SK_RAG_MAX_LEN 2040            /* longest read the lane-per-read kernel takes from a ragged batch */
#define SK_RAG_BUF_DEFAULT (20u * 1024u) /* LDS bytes per wave for ragged tiles when the caller gives no length hint */
#define SK_RAG_BUF_MAX (40u * 1024u)
#ifndef SK_SORT_WINDOW
#define SK_SORT_WINDOW 8192  /* reads per window of the device-side regrouping of mixed-length ragged batches (sk_sort.hip) */
#endif
#define SK_SORT_MIN_READS 65536u /* ragged batches below this keep the plain tile kernel */

struct sk_cut_dev {
    int32_t five, three;
};

// == sk_tile of the C ABI (include/sickle_amd.h)
struct sk_tile_dev {
    uint64_t byte_off;
    uint32_t slot0;
    uint32_t stride;
    uint16_t rows;
    uint16_t read_len;
    uint32_t reserved;
};

// Scalar arguments of one scan, derived on the host from sk_params (+ the batch shape).
struct sk_scan_args {
    uint64_t n_reads;
    uint32_t stride;   // bytes between reads (fixed-stride layouts)
    uint32_t read_len; // uniform read length (fixed-stride, lengths == NULL)
    int32_t qmin, qmax; // legal char range of the encoding (reference src/sickle.h:85-91)
    int32_t craw;      // qual_threshold + offset: the threshold on raw chars (window: craw * w)
    int32_t cthr;      // min(craw, 128), for the byte-parallel compares
    int32_t cthr_raw;  // == craw (scalar compares of the wave kernel)
    int32_t lthr;      // length_threshold
    int32_t no5;       // -x
    int32_t truncn;    // -n
    uint32_t n_tiles;   // segmented batches: number of tile descriptors
    int32_t tile_order; // diagnostic (SK_TILE_ORDER): 0 = tile t on workgroup t mod G, 1 = contiguous tile ranges per XCD
    uint32_t buf_bytes; // LDS bytes per wave (segmented / rows at any address); general kernel: != 0 = only the tiles that do not fit them
    int32_t slot_order;   // segmented batches: cuts written in slot order, out_index only names erroring reads
    uint64_t scan_id;     // number of this scan on its error word (ragged batches: tile kernel -> general kernel hand-over)
    uint32_t team_rbuf;   // general kernels with resident reads (band, team): LDS bytes of one read's buffer
    uint32_t team_maxlen; // ... and the longest read that goes through LDS
    uint32_t stream_nb;   // streaming general kernel: 1 KiB blocks in a wave's ring
    uint32_t stream_read_cost;  // streaming general kernel: what a read costs beyond its bytes when the batch is cut into spans
    uint32_t stream_tbl;  // streaming general kernel: entries of the prefix table (a power of two)
    uint32_t seg_chunk_shift; // segmented batches: a wave takes 1 << this consecutive tiles at a time
    const uint32_t *band_table; // SK_BAND_WIDTHS band matrices, one per window width (sk_band_dword): what a tile kernel loads when a
                                // tile's window width differs from the one before (segmented batches, regrouped ragged batches)
    const uint32_t *sort_flags; // ragged batches behind the device-side regrouping: {windows of mixed lengths, reads too long for the tiles}; the
                                // plain tile kernel (and the general kernel behind it) return at once when the sorted scan runs, and vice versa
};

// The band matrix of window width wu, as lane `lane` of a wave holds it for v_mfma_i32_32x32x32_i8 (sk_kernels.hip, MFMA path):
// the dword whose four bytes multiply positions p .. p+3 (relative to a 32-position block), 1 where the lane's window
// covers the position.  The lane supplies row m' = lane & 31 of A; the hardware puts that row into accumulator r of lane
// half hh with m' = (r & 3) + 8 * (r >> 2) + 4 * hh, and that slot is to be window 16 * hh + r.
#if defined(__HIPCC__) || defined(__CUDACC__)
__host__ __device__
#endif
static inline uint32_t sk_band_dword(int lane, int wu, int p)
{
    const int mp = lane & 31;
    const int hh = (mp >> 2) & 1, r = (mp & 3) | ((mp >> 3) << 2);
    const int win = 16 * hh + r;
    const int hi = win + wu - p, lo = win - p; // bytes j with lo <= j < hi
    const uint32_t below_hi = hi >= 4 ? 0x01010101u : (hi <= 0 ? 0u : 0x01010101u & ((1u << (8 * hi)) - 1u));
    const uint32_t below_lo = lo >= 4 ? 0x01010101u : (lo <= 0 ? 0u : 0x01010101u & ((1u << (8 * lo)) - 1u));
    return below_hi & ~below_lo;
}
// the table: [width 0 .. SK_BAND_WIDTHS-1][block 0..2][lane 0..63][4 dwords]: dword j of block b of a lane = positions
// 16 * (lane >> 5) + 4 * j + 32 * b
#define SK_BAND_WIDTHS 66u
#define SK_BAND_TABLE_DWORDS (SK_BAND_WIDTHS * 3u * 64u * 4u)

// internal to libsickle_amd.so (not part of the C ABI)
extern "C" __attribute__((visibility("hidden"))) int sk_tile_is_staged(uint32_t stride, uint32_t read_len, int has_seq);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_tile(const uint8_t *qual, const uint8_t *seq, const uint32_t *lengths,
                                     sk_cut_dev *out, unsigned long long *errword, const sk_scan_args *a,
                                     int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_seg(const uint8_t *qual, const uint8_t *seq, const sk_tile_dev *tiles,
                                    const uint32_t *out_index, sk_cut_dev *out, unsigned long long *errword,
                                    const sk_scan_args *a, const sk_seg_class *classes, uint32_t n_classes,
                                    int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_team(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                     const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                     const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_band(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                     const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                     const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_stream(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                       const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                       const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_any(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                    const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                    const sk_scan_args *a, int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) uint32_t sk_wide_lds_bytes(uint32_t read_len);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_wide(const uint8_t *qual, const uint8_t *seq, sk_cut_dev *out, unsigned long long *errword,
                                     const sk_scan_args *a, int cu_count, hipStream_t stream);
// mixed-length ragged batches: the per-window regrouping (sk_sort.hip) and the scan of its tiles (sk_kernels.hip)
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_sort(const uint64_t *offsets, uint64_t n_reads, uint32_t max_len, uint64_t *perm,
                                     unsigned long long *lists, uint32_t list_cap, uint32_t *counts, uint32_t *counts_of_next_scan,
                                     hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_sorted(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets, const uint64_t *perm,
                                       const unsigned long long *lists, const uint32_t *counts, sk_cut_dev *out,
                                       unsigned long long *errword, const sk_scan_args *a, int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_pair_count(const sk_cut_dev *cuts, uint64_t n_pairs, uint8_t *classes,
                                           unsigned long long *counters, int cu_count, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_read_probe(const void *buf, size_t bytes, uint32_t *sink, int cu_count,
                                           hipStream_t stream);
#endif
