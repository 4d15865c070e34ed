// sk_kernels.hip -- the quality-scan kernels for gfx950 (CDNA4, wave64).
//
// What is computed, per read, is exactly Abstract_Trimmer::sliding_window of the
// reference (src/trim.cpp:3-116) with get_quality_num's range check (src/trim.cpp:118-140);
// how it is computed is not: the reference walks the read once with a rolling sum and
// early breaks, the kernels use the closed form
//     S_i   = sum of the w = max(L/10, L if L<10) chars of window i      (0 <= i <= L-w)
//     i0    = first i with S_i >= T        (T = (qthr+offset)*w, on raw chars)   -> 5' window
//     i1    = first i >  i0 with S_i <  T  (i >= 0 with -x)                      -> 3' window
//     five  = first j >= i0 with c[j] >= qthr+offset ; three = first j >= i1 with c[j] < qthr+offset
// (the reference's double-precision average compares exactly like the integers), the N rule
// and the length filter applied afterwards, and the range error raised iff the first bad
// char lies in the part of the read the reference would have touched: [0, i1 + w) if the
// 3' break fired, the whole read otherwise, nothing if L < length_threshold.
//
// Two kernels:
//   sk_scan_tile_kernel   fixed-stride batches.  One LANE per read, one wavefront per 64-read
//                         tile, 16 single-wave workgroups per CU.  The tile (64*stride contiguous
//                         bytes) goes HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA, nt policy:
//                         coalesced, no VGPR round trip).  Uniform-length batches take their window
//                         sums from the integer matrix pipe (band(w) x Q as two
//                         v_mfma_i32_32x32x32_i8 per 32 windows x 32 reads, the B operand read
//                         straight out of the LDS rows) and spend one v_alignbit per window on the
//                         vector ALU to collect the signs of S_i - T; mixed-length batches walk
//                         their rows 4 windows per dword with byte-parallel arithmetic
//                         (v_alignbyte, signed byte differences, v_dot4_i32_i8, v_alignbit).
//                         Range check: two v_sad_u8 per dword.
//   sk_scan_team_kernel   any layout (ragged offsets, odd strides, long reads).  One
//                         WAVEFRONT per read; each lane owns a contiguous run of windows,
//                         seeds its window sum directly and rolls it in a register; the
//                         per-lane first-hits are combined with wave min-reductions.
// HBM-bound byte streaming (algorithmic bytes: L + 8 per read, 2L + 8 with -n); the MFMAs are a
// way to take instructions off the vector ALU, not the bound.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "sk_device.h"

namespace {

constexpr uint32_t H4 = 0x80808080u;
constexpr int INF = 0x7fffffff;

__device__ __forceinline__ uint32_t splat(uint32_t b) { return b * 0x01010101u; }

// bit 7 of each byte set iff that byte (assumed < 128) is >= the byte of c4 (each <= 128)
__device__ __forceinline__ uint32_t ge_flags(uint32_t x, uint32_t c4) { return ((x | H4) - c4) & H4; }

// bit 7 of each byte set iff that byte is outside [min, max] (or >= 128):  the range check of
// reference src/trim.cpp:129, four chars at a time.  min4 = splat(min), hi4 = splat(127 - max).
__device__ __forceinline__ uint32_t bad_flags(uint32_t x, uint32_t min4, uint32_t hi4)
{
    uint32_t lo_ok = (x | H4) - min4;  // bit7 set iff byte >= min
    uint32_t hi_bad = (x & ~H4) + hi4; // bit7 set iff (byte & 127) > max
    return (~lo_ok | hi_bad | x) & H4;
}

// the first n bytes of x (n <= 0: none, n >= 4: all), the others taken from `filler`
__device__ __forceinline__ uint32_t first_bytes(uint32_t x, int n, uint32_t filler)
{
    const uint32_t keep = n >= 4 ? ~0u : (n <= 0 ? 0u : (1u << (8 * n)) - 1u);
    return (x & keep) | (filler & ~keep);
}

// flags of bytes [n, 4) cleared, n in 0..4
__device__ __forceinline__ uint32_t keep_first(uint32_t flags, int n)
{
    return n >= 4 ? flags : (n <= 0 ? 0u : flags & ((1u << (8 * n)) - 1u));
}

// v_ffbh_u32 / v_ffbl_b32: index of the first set bit from the top / from the bottom, and
// 0xFFFFFFFF for an empty mask (what __builtin_clz/ctz leave undefined and would guard with an
// extra instruction).  Combined with saturating adds, "no hit" stays 0xFFFFFFFF through a min().
__device__ __forceinline__ uint32_t ffbh_or_none(uint32_t x)
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t ffbl_or_none(uint32_t x)
{
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
constexpr uint32_t NONE = 0xffffffffu;

// ---- reductions on the data-parallel-primitive path (v_*_dpp: a lane reads its neighbour's register in the
// same instruction, no LDS crossbar, no wait): four steps leave the result of each ROW of 16 lanes in all of
// its lanes; the four rows are then combined through scalar registers (v_readlane + s_min/s_max).
constexpr int DPP_QUAD_SWAP1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
template <int CTRL>
__device__ __forceinline__ int dpp_peer(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int row_min(int v) // min over the 16-lane row, in every lane of the row
{
    v = min(v, dpp_peer<DPP_QUAD_SWAP1>(v));
    v = min(v, dpp_peer<DPP_QUAD_SWAP2>(v));
    v = min(v, dpp_peer<DPP_ROW_HALF_MIRROR>(v));
    v = min(v, dpp_peer<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ int row_max(int v)
{
    v = max(v, dpp_peer<DPP_QUAD_SWAP1>(v));
    v = max(v, dpp_peer<DPP_QUAD_SWAP2>(v));
    v = max(v, dpp_peer<DPP_ROW_HALF_MIRROR>(v));
    v = max(v, dpp_peer<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ uint32_t row_or(uint32_t v)
{
    v |= (uint32_t)dpp_peer<DPP_QUAD_SWAP1>((int)v);
    v |= (uint32_t)dpp_peer<DPP_QUAD_SWAP2>((int)v);
    v |= (uint32_t)dpp_peer<DPP_ROW_HALF_MIRROR>((int)v);
    v |= (uint32_t)dpp_peer<DPP_ROW_MIRROR>((int)v);
    return v;
}
// wave-wide: one value per wave, in scalar registers
__device__ __forceinline__ int wave_min(int v)
{
    v = row_min(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v)
{
    v = row_max(v);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v = row_or(v);
    return (uint32_t)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
                      __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}

__device__ __forceinline__ void report_error(unsigned long long *errword, uint64_t read, int pos, int ch)
{
    // lowest read index wins; ties (same read) resolve to the lowest position
    unsigned long long key = ((unsigned long long)read << 32) | ((unsigned long long)(uint32_t)pos << 8) |
                             (unsigned long long)(uint32_t)(ch & 0xff);
    atomicMin(errword, key);
}

// cache policy of the tile DMA (the aux operand of global_load_lds): 0 = default, 2 = nt.
// Every tile byte is read exactly once, so nt: measured -9 % on the DMA-only floor and -8 % on the
// whole kernel against the default policy (interleaved A/B on one device, tools/ablate.py).
#ifndef SK_TAIL_PRIO
#define SK_TAIL_PRIO 0
#endif
#ifndef SK_DMA_AUX
#define SK_DMA_AUX 2
#endif
using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;

// Copies `bytes` (multiple of 4, wave-uniform) from global `src` (16-byte aligned) to the
// wave-private LDS region `dst` with LDS-DMA; the LDS image is byte-identical to the global one.
// Full 1 KiB pieces are issued four per trip through the instruction's immediate offset (it
// moves the global and the LDS address together), without touching EXEC; only the last, partial
// piece is predicated.
__device__ __forceinline__ void tile_to_lds(const uint8_t *src, uint8_t *dst, uint32_t bytes, int lane)
{
    const uint32_t nfull = bytes >> 4;  // 16-byte chunks
    const uint32_t pieces = nfull >> 6; // full 64-lane pieces
    const uint8_t *sp = src + (size_t)lane * 16;
    uint32_t p = 0;
    for (; p + 4 <= pieces; p += 4) {
        gptr_t g = (gptr_t)(sp + (size_t)p * 1024);
        lptr_t l = (lptr_t)(dst + p * 1024);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, SK_DMA_AUX);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, SK_DMA_AUX);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, SK_DMA_AUX);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, SK_DMA_AUX);
    }
    for (; p < pieces; ++p)
        __builtin_amdgcn_global_load_lds((gptr_t)(sp + (size_t)p * 1024), (lptr_t)(dst + p * 1024), 16, 0, SK_DMA_AUX);
    if ((uint32_t)lane < (nfull & 63u))
        __builtin_amdgcn_global_load_lds((gptr_t)(sp + (size_t)p * 1024), (lptr_t)(dst + p * 1024), 16, 0, SK_DMA_AUX);
    const uint32_t tail = (bytes & 15u) >> 2; // 0..3 dwords after the last full 16-byte chunk
    if ((uint32_t)lane < tail)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)nfull * 16 + lane * 4), (lptr_t)(dst + nfull * 16),
                                         4, 0, 0);
}

} // namespace

// ------------------------------------------------------------------------------------------
// Tiled kernel: lane per read, software-pipelined.
//
// Each wave owns two LDS buffers.  While it scans tile t out of one buffer the LDS-DMA of
// its next tile is already in flight into the other, so HBM latency is hidden inside the
// wave instead of relying on other waves being out of phase (measured: without this the
// whole chip convoys -- every wave loads, then every wave computes -- at ~half the rate).
// The DMA of a tile is retired with a COUNTED s_waitcnt vmcnt(n): n = the vector-memory
// operations issued after it (the next tile's pieces and the cut store), which are allowed
// to stay outstanding.  With -n the sequence tile of the same reads rides the same two
// buffers: Q(t) -> buf0, S(t) -> buf1, Q(t+1) -> buf0, ...
// ------------------------------------------------------------------------------------------
namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt_imm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt takes an immediate; n is wave-uniform and small (pieces of one tile + 1)
__device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (n) {
#define SK_CASE(N) case N: wait_vmcnt_imm<N>(); break;
        SK_CASE(0) SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7)
        SK_CASE(8) SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15)
        SK_CASE(16) SK_CASE(17) SK_CASE(18) SK_CASE(19) SK_CASE(20) SK_CASE(21) SK_CASE(22) SK_CASE(23)
        SK_CASE(24) SK_CASE(25) SK_CASE(26) SK_CASE(27) SK_CASE(28) SK_CASE(29) SK_CASE(30) SK_CASE(31)
        SK_CASE(32) SK_CASE(33) SK_CASE(34)
#undef SK_CASE
    default: wait_vmcnt_imm<0>(); break;
    }
}

// number of vector-memory instructions tile_to_lds issues for `bytes`
__device__ __forceinline__ int tile_pieces(uint32_t bytes)
{
    return (int)(((bytes >> 4) + 63) >> 6) + (((bytes & 15u) >> 2) ? 1 : 0);
}

} // namespace

#define SK_STAGE_MIN 5  /* register-staged kernels exist for tiles of 5..10 KiB: row strides 72..160 */
#define SK_STAGE_MAX 10
typedef int sk_v4i __attribute__((ext_vector_type(4)));
typedef int sk_v16i __attribute__((ext_vector_type(16)));
typedef unsigned sk_v2u __attribute__((ext_vector_type(2)));

// MFMA = true (uniform-length batches; w <= 65, i.e. every uniform length the tiled kernel takes):
// the window sums are taken off the vector ALU.  A box filter is a banded 0/1 matrix, so for 32 windows x 32 reads
//     S[window][read] - T = band(w)[window][pos] x Q[pos][read] + (-T)
// is two v_mfma_i32_32x32x32_i8 (positions 32b..32b+63; three for w > 33), exact in int32.  The B operand of a lane
// is 16 consecutive quality bytes of one read -- two ds_read_b64 from its LDS row, no shuffling;
// the A operand is a per-lane constant.  The rows of `band` are permuted so that a lane's 16
// accumulators are 16 CONSECUTIVE windows (lane half h: windows 16h..16h+15), which leaves the
// vector ALU one v_alignbit per window to collect the sign bits.  The integer matrix pipe is
// otherwise idle in this kernel.  This is not a GEMM reshaping of the problem: the data stay in
// their row layout and every byte is still read from HBM exactly once.
//
// NBUF = LDS buffers per wave for the quality tile: 2 = the DMA of tile t+1 overlaps the scan of
// tile t inside the wave (8 waves per CU); 1 = a tile is loaded, scanned, then replaced, and the
// overlap comes from having 16 waves per CU out of phase.  With -n (HAS_SEQ) the two buffers
// hold the quality and the sequence tile of the same reads and each is refilled as soon as its
// scan is over.
//
// ABLATE (diagnostic launches of tools/ablate.py only; the product always runs 0):
//   1 = DMA + cut store only (no scan), 2 = scan only (tile loaded once, then reused)
//
// SEG = segmented batches (mixed lengths, sorted by length on the host): every tile has its own
// descriptor -- byte offset, row stride, read length, row count -- and is uniform inside, so it
// takes the matrix path like a uniform batch (the band matrix is rebuilt when the length changes,
// which in a sorted batch is rare); cuts are scattered to out[out_index[slot]].
//
// STAGE > 0 (uniform batches without -n whose tile fits STAGE KiB): the tile does not come in by
// LDS-DMA but through the wave's own registers -- STAGE global_load_dwordx4 (nt) of the NEXT tile
// are in flight while this tile is scanned, and are written to the LDS buffer (ds_write_b128, same
// image as the DMA's) once the scan is over.  Plain loads stream faster than LDS-DMA on this
// device (tools/probes/read_bw.hip: 7.0 against 6.5 TB/s), and the registers act as a second
// buffer per wave without costing LDS.
typedef unsigned sk_v4u __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// row pitch of the LDS image the re-striding loader builds for reads of up to `len` bytes.  The loader
// moves 16 bytes per lane, so a multiple of 16, with an ODD number of 16-byte units: the rows' 8-byte
// reads (matrix path) then fall two lanes to a bank pair and their 4-byte reads (vector-ALU path) four
// lanes to a bank -- the best 16-byte granules allow (SQ_LDS_BANK_CONFLICT is 56 % of the LDS cycles on
// packed 150 bp).  Measured alternative: 4 bytes per lane and a pitch of 8 * odd (no conflicts, four
// times the DMA instructions) is slower everywhere -- packed 150 bp 0.44 against 0.33 ms, ragged 150 bp
// 0.56 against 0.42 ms, a 75-301 bp mix 0.59 against 0.46 ms: the loader's instruction count costs
// more than the conflicts.
template <bool UNIFORM>
__device__ __forceinline__ uint32_t rag_pitch(uint32_t len)
{
    return 16u * (((len + 15u) >> 4) | 1u);
}

// one LDS-DMA of the re-striding loader: 16 bytes per lane (uniform lengths) or 4
template <bool WIDE>
__device__ __forceinline__ void dma_piece(const uint8_t *g, uint8_t *l)
{
    if (WIDE) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, SK_DMA_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, SK_DMA_AUX);
}

// One tile = reads [64t, 64t+64) of a batch whose rows start at any byte address (`offsets`, or a
// fixed stride that is not a multiple of 8, with or without `lengths`), as its lanes see it
struct sk_rag_tile {
    uint64_t start;  // wave-uniform: byte offset of the tile's first read
    uint32_t span;   // wave-uniform: bytes from there to the end of its last read (saturated)
    uint32_t rowoff; // per lane: this lane's read starts at start + rowoff
    int len;         // per lane: its length (0 for lanes past the end of the batch)
    int lmax;        // wave-uniform: the longest of them
};

__device__ __forceinline__ sk_rag_tile rag_probe(uint64_t t, int lane, const uint64_t *__restrict__ offsets,
                                                 const uint32_t *__restrict__ lengths, const sk_scan_args &a)
{
    const uint64_t r = (t << 6) + lane;
    const uint64_t rc = min(r, a.n_reads - 1);
    uint64_t o, e;
    if (offsets) {
        o = offsets[rc];
        e = offsets[rc + 1];
    } else {
        o = rc * a.stride;
        e = o + (lengths ? min(lengths[rc], a.stride) : a.read_len);
    }
    const int last = (int)min((uint64_t)63, a.n_reads - 1 - (t << 6));
    sk_rag_tile g;
    g.start = readlane_u64(o, 0);
    const uint64_t end = readlane_u64(e, last);
    const uint64_t span = end >= g.start ? end - g.start : ~0ull;
    g.span = (uint32_t)min(span, (uint64_t)0xffffffffu);
    // offsets that do not ascend give a read no bytes rather than bytes outside its tile
    const bool ok = r < a.n_reads && o >= g.start && e >= o && e <= end;
    g.rowoff = ok ? (uint32_t)(o - g.start) : 0u;
    g.len = ok ? (int)min(e - o, (uint64_t)SK_MAX_READ_LEN_DEV) : 0;
    g.lmax = wave_max(g.len);
    return g;
}

// Is the tile sk_scan_tile_any_kernel's?  Its re-strided image (64 rows at rag_pitch(lmax)) must fit
// the wave's LDS buffer.  sk_scan_team_kernel asks the same question and takes the other tiles.
__device__ __forceinline__ bool rag_tile_fits(const sk_rag_tile &g, uint32_t buf_bytes)
{
    return g.lmax <= SK_RAG_MAX_LEN && 64u * rag_pitch<false>((uint32_t)g.lmax) + SK_TILE_SLACK <= buf_bytes;
}

// end of the batch's bytes (exclusive), for the test above
__device__ __forceinline__ uint64_t rag_batch_end(const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lengths,
                                                  const sk_scan_args &a)
{
    if (offsets) return offsets[a.n_reads];
    return (a.n_reads - 1) * a.stride + (lengths ? min(lengths[a.n_reads - 1], a.stride) : a.read_len);
}

// what a wave needs to know about one tile
struct sk_tile_view {
    uint64_t off;    // wave-uniform: byte offset of the tile (of its first read) in qual / seq
    uint32_t bytes;  // wave-uniform: bytes of the tile in global memory
    uint32_t ts;     // wave-uniform: row pitch of its LDS image
    uint32_t rows;   // wave-uniform: reads in it
    int len;         // read length: one value when UNIFORM, per lane otherwise (0 past the end)
    uint64_t r;      // per lane: where this lane's cut goes in out[]
    uint32_t rowoff; // ragged: per lane, where the lane's read starts, relative to off
    bool take;       // rows at any address: false = left to sk_scan_team_kernel
    bool uni;        // ragged: the tile's 64 reads have one length (their rows are then len apart)
};

} // namespace

// RAG (rows at any byte address: packed fixed-stride batches whose stride is not a multiple of 8 or
// whose base is not 16-byte aligned, and ragged `offsets` batches): the tile is RE-STRIDED on its way
// into LDS.  LDS-DMA takes a per-lane global address, so lane i of piece p fetches the 4 bytes that
// belong at dword 64p+i of an image with rows at pitch rag_pitch(longest read) -- (row, dword) =
// divmod(64p+i, pitch/4), source = that row's start + 4*dword, at whatever alignment -- and the
// image the scan sees is the aligned, bank-friendly one of the strided layouts.  Measured
// (tools/probes/restride_probe.hip): the re-striding DMA streams at the rate of the plain one
// (6.4 TB/s), whereas reading unaligned rows out of LDS costs 8x per ds_read.  Uniform lengths keep
// the matrix path; per-lane lengths walk the vector-ALU path.  A tile whose image does not fit the
// wave's buffer (long reads) is left to sk_scan_team_kernel.
template <bool UNIFORM, bool HAS_SEQ, bool MFMA, int NBUF, int ABLATE, int SEG, int STAGE, bool RAG>
__device__ __forceinline__ void
sk_scan_tile_body(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                  const uint32_t *__restrict__ lengths, sk_cut_dev *__restrict__ out,
                  unsigned long long *errword, const sk_scan_args &a, const sk_tile_dev *__restrict__ tiles,
                  const uint32_t *__restrict__ out_index, const uint64_t *__restrict__ offsets)
{
    static_assert(!RAG || (NBUF == 1 && !SEG && STAGE == 0 && ABLATE == 0), "re-strided tiles: one buffer, LDS-DMA");
    static_assert(!MFMA || UNIFORM || RAG, "the matrix path needs one window width per tile");
    // MIXED (ragged batches): per-lane lengths in general, but a tile whose 64 reads have ONE length --
    // every tile of the usual fixed-length run handed over as offsets -- takes the matrix path and the
    // uniform row walks; the other tiles walk the vector-ALU path.  Decided per tile, wave-uniformly.
    constexpr bool MIXED = MFMA && !UNIFORM;
    static_assert(STAGE == 0 || (UNIFORM && !HAS_SEQ && NBUF == 1 && !SEG), "register staging: uniform batches, one buffer");
    static_assert(!SEG || (UNIFORM && MFMA && NBUF == 1), "segmented batches run the uniform matrix path, one buffer");
    // -n: NBUF == 2 keeps the quality and the sequence tile in two buffers (8 waves per CU);
    // NBUF == 1 runs both through ONE buffer, one after the other (16 waves per CU)
    constexpr int LDS_BUFS = NBUF;
    constexpr bool SEQ_SHARES = HAS_SEQ && NBUF == 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    // readfirstlane: tells the compiler this is one value per wave, so that the tile index and
    // everything derived from it (addresses, piece counts, loop and switch conditions) live in
    // SGPRs and branch on the scalar unit instead of being carried through the vector ALU
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waves_per_block = blockDim.x >> 6;
    const uint32_t stride = a.stride; // SEG: the largest row stride of the batch (sizes the LDS buffers)
    const uint32_t buf_bytes = (RAG || SEG) ? a.buf_bytes : 64u * stride + SK_TILE_SLACK;
    uint8_t *buf0 = lds + (size_t)wave * LDS_BUFS * buf_bytes;
    uint8_t *buf1 = LDS_BUFS > 1 ? buf0 + buf_bytes : buf0;

    const uint64_t n_tiles = SEG ? (uint64_t)a.n_tiles : (a.n_reads + 63) >> 6;
    uint64_t wave_global = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint64_t wave_count = (uint64_t)gridDim.x * waves_per_block;
    if (a.tile_order == 1 && (wave_count & 7) == 0) {
        // workgroups b and b+8 share an XCD (round-robin dispatch): give each XCD group a
        // contiguous run of wave slots, so that it streams a contiguous eighth of every stripe
        const uint64_t per = wave_count >> 3;
        wave_global = (wave_global & 7) * per + (wave_global >> 3);
    }

    const uint32_t min4 = splat((uint32_t)a.qmin), max4 = splat((uint32_t)a.qmax);
    const uint32_t hi4 = splat((uint32_t)(127 - a.qmax));
    const uint32_t cthr4 = splat((uint32_t)a.cthr);
    const int range = a.qmax - a.qmin;

    // uniform-length batches (and each tile of a segmented one): one length, one window width, one
    // window count for every lane
    int Lu = 0, wu = 0;
    bool scan_u = false, three_blocks = false;
    sk_v4i bandA0 = {0, 0, 0, 0}, bandA1 = {0, 0, 0, 0}, bandA2 = {0, 0, 0, 0};
    sk_v16i negT;
    const int half = lane >> 5;
    auto set_length = [&](int len) {
        Lu = len;
        scan_u = Lu > 0 && Lu >= a.lthr;  // reference trim.cpp:21
        wu = Lu / 10 ? Lu / 10 : Lu;      // trim.cpp:8,30
        // windows wider than 33 reach into a third 32-position block (never in the staged kernels: rows
        // <= 160 bytes).  Segmented launches fix it at compile time (SEG = 2: every tile of the launch
        // has w <= 33; SEG = 3: three blocks for every tile, right for any w <= 65): one MFMA loop
        // instead of two in the kernel, fewer registers
        three_blocks = MFMA && STAGE == 0 && (SEG == 2 ? false : SEG == 3 ? true : wu > 33);
        if (MFMA) {
            // ---- constants of the matrix path.  This lane supplies row m' = lane&31 of A; the
            // hardware puts row m' into accumulator reg r of lane half hh with
            // m' = (r&3) + 8*(r>>2) + 4*hh; we want that slot to be window 16*hh + r
            int mp = lane & 31;
            // segmented batches call this inside the tile loop: without the barrier the compiler hoists
            // the 48 per-byte position constants out of the loop and pins a register to each
            if (SEG || MIXED) asm volatile("" : "+v"(mp));
            const int hh = (mp >> 2) & 1, r = (mp & 3) | ((mp >> 3) << 2);
            const int win = 16 * hh + r;
            // band bytes of positions p .. p+3 (relative to 32*b): 1 where win <= position < win + wu
            auto ones_below = [](int n) -> uint32_t { // 0x01 in the bytes j < n of a dword
                return n >= 4 ? 0x01010101u : (n <= 0 ? 0u : 0x01010101u & ((1u << (8 * n)) - 1u));
            };
            auto band4 = [&](int p) -> int { return (int)(ones_below(win + wu - p) & ~ones_below(win - p)); };
            const int k0 = (lane >> 5) * 16; // the first position (relative to 32*b) this lane's bytes multiply
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bandA0[j] = band4(k0 + 4 * j);
                bandA1[j] = band4(k0 + 4 * j + 32);
                bandA2[j] = band4(k0 + 4 * j + 64);
            }
            const int T = a.craw * wu;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int seed = -T;
                // opaque to the compiler: otherwise it keeps this uniform value in SGPRs and copies
                // it into 16 VGPRs again before every MFMA pair (8 v_mov_b64 per 32 windows)
                asm volatile("" : "+v"(seed));
                negT[i] = seed;
            }
        }
    };
    if (!SEG) set_length((int)a.read_len);

    const uint64_t batch_end = RAG ? rag_batch_end(offsets, lengths, a) : 0;
    auto probe = [&](uint64_t tt) -> sk_tile_view {
        sk_tile_view v;
        v.take = true;
        v.uni = false;
        v.rowoff = 0;
        if (SEG) {
            const sk_tile_dev d = tiles[tt];
            v.off = d.byte_off;
            v.ts = d.stride;
            v.rows = d.rows;
            v.bytes = v.rows * v.ts;
            v.len = (int)d.read_len;
            v.r = d.slot0; // probe_index() turns it into this lane's slot of out[]
        } else if (RAG && UNIFORM) {
            v.off = (tt << 6) * stride;
            v.rows = (uint32_t)min((uint64_t)64, a.n_reads - (tt << 6));
            v.ts = rag_pitch<true>(a.read_len);
            v.bytes = (v.rows - 1u) * stride + a.read_len;
            v.len = (int)a.read_len;
            v.r = (tt << 6) + lane;
        } else if (RAG) {
            const sk_rag_tile g = rag_probe(tt, lane, offsets, lengths, a);
            v.off = g.start;
            v.bytes = g.span;
            v.rows = (uint32_t)min((uint64_t)64, a.n_reads - (tt << 6));
            v.ts = rag_pitch<false>((uint32_t)g.lmax);
            v.len = g.len;
            v.rowoff = g.rowoff;
            v.r = (tt << 6) + lane;
            v.take = rag_tile_fits(g, buf_bytes);
            v.uni = v.rows == 64u && __builtin_amdgcn_ballot_w64(g.len != g.lmax) == 0;
        } else {
            v.off = (tt << 6) * stride;
            v.rows = (uint32_t)min((uint64_t)64, a.n_reads - (tt << 6));
            v.ts = stride;
            v.bytes = v.rows * stride;
            v.r = (tt << 6) + lane;
            v.len = UNIFORM ? (int)a.read_len : (v.r < a.n_reads ? (int)min(lengths[v.r], stride) : 0);
        }
        return v;
    };

    // segmented batches scatter their cuts back to the caller's read order.  The descriptor is a scalar
    // load (not counted by vmcnt: it can be issued before the wait for the tile); the index is a vector
    // load that needs the descriptor, issued after that wait, when the descriptor has long arrived
    auto probe_index = [&](sk_tile_view &v) {
        const uint32_t slot = (uint32_t)v.r + min((uint32_t)lane, v.rows - 1u);
        v.r = a.slot_order ? (uint64_t)slot : (uint64_t)out_index[slot]; // slot order: one coalesced stream of cuts
    };

    // the re-striding loader (RAG): image chunk 64p + lane = (row, c) = divmod(64p + lane, chunks per row)
    // comes from the row's start + c chunks.  Chunks beyond a row's end fetch what follows it in the batch (nobody
    // reads them); the clamp keeps those inside the tile (a row's last chunk may still reach up to 15
    // bytes past it).
    auto rag_dma = [&](const uint8_t *base, uint8_t *dst, const sk_tile_view &v) {
        constexpr uint32_t GRAN = 16u; // bytes per lane per DMA (see rag_pitch)
        const uint32_t cpr = v.ts / GRAN;             // chunks per image row == pieces per tile
        const uint32_t qd = 64u / cpr, rd = 64u % cpr;
        uint32_t rr = (uint32_t)lane / cpr, cc = (uint32_t)lane % cpr;
        const uint8_t *src = base + v.off;
        const uint32_t lim = (v.bytes ? v.bytes : 1u) - 1u;
        for (uint32_t p = 0; p < cpr; ++p) {
            uint32_t ro;
            if (UNIFORM) ro = rr * stride;
            else if (v.uni) ro = rr * (uint32_t)__builtin_amdgcn_readfirstlane(v.len); // equal lengths: rows len apart
            else ro = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rr << 2), (int)v.rowoff);
            const uint32_t so = min(ro + GRAN * cc, lim);
            // a chunk may reach past its tile: harmless inside the batch, but the last chunks of the
            // batch's last tile(s) must not leave the caller's buffer -- those few lanes copy their
            // bytes one by one instead
            const bool inside = v.off + so + GRAN <= batch_end;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!inside) == 0, 1)) {
                dma_piece<true>(src + so, dst + p * (64u * GRAN));
            } else {
                if (inside) {
                    dma_piece<true>(src + so, dst + p * (64u * GRAN));
                } else {
                    for (uint32_t j = 0; j < GRAN && v.off + so + j < batch_end; ++j)
                        dst[p * (64u * GRAN) + (uint32_t)lane * GRAN + j] = src[so + j];
                }
            }
            cc += rd;
            rr += qd;
            if (cc >= cpr) {
                cc -= cpr;
                ++rr;
            }
        }
    };
    auto load_tile = [&](const uint8_t *base, uint8_t *dst, const sk_tile_view &v) {
        if (RAG) rag_dma(base, dst, v);
        else tile_to_lds(base + v.off, dst, v.bytes, lane);
    };

    // register stage: piece p of a FULL tile (64 rows; 64*stride bytes, a multiple of 512) is the
    // 16 bytes per lane at p KiB; STAGE = the number of pieces, the last one may be a half (its
    // upper lanes repeat the tile's last 16 bytes: same data to the same place, no predication).
    // The last tile of a batch, if it is not full, comes in by LDS-DMA like in the unstaged kernel.
    sk_v4u stage[STAGE ? STAGE : 1];
    const uint32_t full_bytes = 64u * stride;
    auto stage_off = [&](int p) -> uint32_t {
        const uint32_t off = (uint32_t)p * 1024u + (uint32_t)lane * 16u;
        return p == STAGE - 1 ? min(off, full_bytes - 16u) : off;
    };

    // Which tiles a wave takes: tile t, then t + (number of waves), ... -- except in segmented batches, where a
    // wave takes SEG_CHUNK consecutive tiles at a time.  There the tiles are sorted by length, and with single
    // steps every wave met a new length at every tile (its tiles lie `waves` apart) and rebuilt the band matrix
    // each time (set_length: ~150 instructions against ~600 for the tile's scan); a chunk shares one length.
    constexpr uint64_t SEG_CHUNK = 4;
    auto next_tile = [&](uint64_t tt) -> uint64_t {
        if (SEG) return ((tt + 1) % SEG_CHUNK != 0) ? tt + 1 : tt + 1 + (wave_count - 1) * SEG_CHUNK;
        return tt + wave_count;
    };
    uint64_t t = SEG ? wave_global * SEG_CHUNK : wave_global;
    if (t >= n_tiles) return;

    // prologue: Q(t) [and S(t)] in flight
    sk_tile_view cur = probe(t), nxt = cur;
    if (SEG) probe_index(cur);
    bool cur_staged = STAGE && cur.rows == 64u;
    if (cur_staged) {
#pragma unroll
        for (int p = 0; p < STAGE; ++p)
            stage[p] = __builtin_nontemporal_load(reinterpret_cast<const sk_v4u *>(qual + cur.off + stage_off(p)));
    } else if (cur.take) {
        load_tile(qual, buf0, cur);
    }
    if (HAS_SEQ && !SEQ_SHARES) tile_to_lds(seq + cur.off, buf1, cur.bytes, lane);
    int parity = 0; // NBUF == 2: which buffer holds Q(t)
    // the next tile's view is taken AFTER this tile has landed wherever taking it loads something
    // (descriptor and out_index of a segmented batch, offsets / lengths): the wait for the tile is a
    // vmcnt(0), and a load issued just before it would put its whole latency on every tile
    constexpr bool PROBE_EARLY = NBUF == 2 || STAGE != 0 || ABLATE != 0 || SEG != 0;

    for (; t < n_tiles; t = next_tile(t)) {
        const uint64_t tn = next_tile(t);
        const bool more = tn < n_tiles;
        if (PROBE_EARLY && more) nxt = probe(tn);
        const uint32_t ts = cur.ts;  // this tile's row pitch in LDS
        const uint64_t r = cur.r;
        const uint32_t cur_bytes = cur.bytes;
        if (SEG && cur.len != Lu) set_length(cur.len);
        const uint32_t next_bytes = (PROBE_EARLY && more) ? nxt.bytes : 0u;
        const int next_pieces = (PROBE_EARLY && more) ? tile_pieces(next_bytes) : 0;
        const uint8_t *tile;

        const bool active = (uint32_t)lane < cur.rows;
        const int Lv = UNIFORM ? 0 : cur.len; // mixed lengths: this lane's length (0 past the end of the batch)
        bool tile_u = false; // MIXED: this tile's reads have one length (and a window the matrix path takes)
        if (MIXED) {
            const int l0 = __builtin_amdgcn_readfirstlane(cur.len);
            tile_u = cur.uni && l0 > 0 && l0 / 10 <= 65;
            if (tile_u && l0 != Lu) set_length(l0);
        }

        if (SEQ_SHARES || RAG) {
            tile = buf0;
            wait_vmcnt(0); // Q(t)
        } else if (HAS_SEQ) {
            tile = buf0;
            // outstanding, oldest first: Q(t), S(t) [, store(t-1) before them]
            wait_vmcnt(tile_pieces(cur_bytes));
        } else if (STAGE) {
            tile = buf0;
            const bool next_staged = more && nxt.rows == 64u && ABLATE != 2;
            if (cur_staged && (ABLATE != 2 || t == wave_global)) {
                // piece by piece: into the LDS buffer, and the register is reloaded at once with the
                // same piece of the next tile (of the first KiB of this tile again if there is no full
                // next tile: loads nobody uses keep the code free of branches and its wait counts exact)
                const uint8_t *nsrc = qual + (next_staged ? nxt.off : cur.off);
                const uint32_t keep = next_staged ? ~0u : 1023u; // no next tile: every piece re-reads the first KiB
#pragma unroll
                for (int p = 0; p < STAGE; ++p) {
                    const uint32_t off = stage_off(p);
                    *reinterpret_cast<sk_v4u *>(buf0 + off) = stage[p];
                    stage[p] = __builtin_nontemporal_load(reinterpret_cast<const sk_v4u *>(nsrc + (off & keep)));
                }
            } else if (!cur_staged) {
                wait_vmcnt(0); // the ragged last tile, by DMA
            }
            cur_staged = next_staged;
        } else if (NBUF == 1 || ABLATE == 2) {
            tile = buf0;
            wait_vmcnt(0);
        } else {
            tile = parity ? buf1 : buf0;
            uint8_t *other = parity ? buf0 : buf1;
            if (more) tile_to_lds(qual + nxt.off, other, next_bytes, lane);
            wait_vmcnt(next_pieces); // everything older than Q(t+1) has landed: Q(t), store(t-1)
            parity ^= 1;
        }
        if (!PROBE_EARLY && more) nxt = probe(tn);
        if (SEG && more) probe_index(nxt);
        if (RAG && !cur.take) { // nothing was loaded: this tile is sk_scan_team_kernel's
            // tell it that there is work: the word after the error word takes this scan's number (scans of a
            // stream are ordered and numbered upwards, so the word never needs a reset)
            if (lane == 0) atomicMax(errword + 1, (unsigned long long)a.scan_id);
            if (more && nxt.take) load_tile(qual, buf0, nxt);
            cur = nxt;
            continue;
        }
        const uint32_t *row = reinterpret_cast<const uint32_t *>(tile + (size_t)lane * ts);
        if (ABLATE == 1) {
            const sk_cut_dev dummy{(int)row[0], (int)row[1]};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (NBUF == 1 && more && (!STAGE || !cur_staged)) tile_to_lds(qual + nxt.off, buf0, next_bytes, lane);
            if (active) out[r] = dummy;
            cur = nxt;
            continue;
        }

        // Lanes past the end of the batch scan whatever sits in their LDS row (every read below
        // is bounded) and are dropped at the store; in uniform batches that keeps L, w and the
        // window count in scalar registers.
        const bool scanned = UNIFORM ? (active && scan_u) : (active && Lv > 0 && Lv >= a.lthr);
        const int L = UNIFORM ? (scan_u ? Lu : 0) : (scanned ? Lv : 0);
        int w = L / 10; // trim.cpp:8 (int)(0.1*L) == L/10
        if (w == 0) w = L; // trim.cpp:30
        const int nwin = (UNIFORM ? scan_u : scanned) ? L - w + 1 : 0;
        const int m = w >> 2, sh = w & 3;
        // one value per wave: known (uniform batch), the lanes' common value (uniform tile of a ragged batch), or reduced
        const int Lmax = UNIFORM ? L : tile_u ? __builtin_amdgcn_readfirstlane(L) : wave_max(L);
        const int wmax = UNIFORM ? w : tile_u ? __builtin_amdgcn_readfirstlane(w) : wave_max(w);
        const int nwinmax = UNIFORM ? nwin : tile_u ? __builtin_amdgcn_readfirstlane(nwin) : wave_max(nwin);

        // ---- range check of the whole read in 2 ops per dword: for a char c in [min,max],
        // |c-min| + |c-max| == max-min, and it is larger for every other byte value, so the
        // read is clean iff the two SADs add up to L*(max-min).  (Whether a bad char counts is
        // decided below against the part of the read the reference would have touched.)
        uint32_t sad = 0;
        {
            int k = 0;
            if (UNIFORM || tile_u) {
                const int full = Lmax >> 2; // whole dwords; rows are 8-byte aligned
                const uint64_t *row64 = reinterpret_cast<const uint64_t *>(row);
                for (; k + 8 <= full; k += 8) { // 4 x ds_read_b64 in flight, then 16 SADs
                    uint64_t x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = row64[(k >> 1) + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        sad = __builtin_amdgcn_sad_u8((uint32_t)x[u], min4, sad);
                        sad = __builtin_amdgcn_sad_u8((uint32_t)x[u], max4, sad);
                        sad = __builtin_amdgcn_sad_u8((uint32_t)(x[u] >> 32), min4, sad);
                        sad = __builtin_amdgcn_sad_u8((uint32_t)(x[u] >> 32), max4, sad);
                    }
                }
                for (; k + 2 <= full; k += 2) {
                    const uint64_t x = row64[k >> 1];
                    sad = __builtin_amdgcn_sad_u8((uint32_t)x, min4, sad);
                    sad = __builtin_amdgcn_sad_u8((uint32_t)x, max4, sad);
                    sad = __builtin_amdgcn_sad_u8((uint32_t)(x >> 32), min4, sad);
                    sad = __builtin_amdgcn_sad_u8((uint32_t)(x >> 32), max4, sad);
                }
            }
            for (; 4 * k < Lmax; ++k) { // per-lane masking: the last dword(s) (UNIFORM) / mixed lengths
                const uint32_t x = first_bytes(row[k], L - 4 * k, min4);
                sad = __builtin_amdgcn_sad_u8(x, min4, sad);
                sad = __builtin_amdgcn_sad_u8(x, max4, sad);
            }
            // every dword visited contributes 4*range when clean (fillers are legal chars)
            sad -= (uint32_t)(4 * k * range);
        }
        const bool bad = scanned && sad != 0;

        // ---- all windows, 32 per trip: trim.cpp:34-81 without the breaks.  Branch-free state
        // step, so that the compiler can overlap it with the next trip's loads and MFMAs.
        // i0u / i1u: the 5' and the 3' window, NONE until found (a min() over the trips keeps
        // the first one, because later trips can only offer larger indices)
        uint32_t i0u = NONE, i1u = NONE;
        bool all_have5 = false; // wave-uniform
        // bit (31 - s) of M: window base+s is below the threshold
        auto step32 = [&](uint32_t M, int base) {
            const int nv = nwin - base;
            const uint32_t vmask = nv >= 32 ? ~0u : (nv <= 0 ? 0u : ~(~0u >> nv));
            const uint32_t lt = M & vmask;
            uint32_t cand = lt; // with -x the 3' search starts at window 0 (trim.cpp:62)
            if (!a.no5 && !all_have5) {
                const uint32_t ge = ~M & vmask;
                i0u = min(i0u, __builtin_elementwise_add_sat(ffbh_or_none(ge), (uint32_t)base)); // trim.cpp:42
                // windows of this trip strictly after i0: the low (base+31 - i0) bits, all 32 if
                // i0 lies in an earlier trip, none while it is not found
                const uint32_t width = __builtin_elementwise_sub_sat((uint32_t)(base + 31), i0u);
                const uint32_t low = (1u << (width & 31u)) - 1u;
                cand = lt & (width >= 32u ? ~0u : low);
                // once every lane has its 5' window, later trips need neither the search nor the mask
                all_have5 = __builtin_amdgcn_ballot_w64(i0u == NONE) == 0;
            }
            i1u = min(i1u, __builtin_elementwise_add_sat(ffbh_or_none(cand), (uint32_t)base)); // trim.cpp:61
        };

        if (MFMA && (UNIFORM || tile_u)) {
            // lane (n = lane&31, half): 16 bytes of read 32g+n at positions 32*kb + 16*half
            const uint8_t *frag0 = tile + (size_t)(lane & 31) * ts + 16 * half;
            const uint8_t *frag1 = frag0 + (size_t)32 * ts;
            auto load_frag = [](const uint8_t *p) -> sk_v4i {
                const uint64_t lo = *reinterpret_cast<const uint64_t *>(p);
                const uint64_t hi = *reinterpret_cast<const uint64_t *>(p + 8);
                sk_v4i f = {(int)(uint32_t)lo, (int)(uint32_t)(lo >> 32), (int)(uint32_t)hi, (int)(uint32_t)(hi >> 32)};
                return f;
            };
            // 16 sign bits per accumulator block, window order; lanes 0..31 keep reads 0..31 (group
            // 0), lanes 32..63 reads 32..63 (group 1): after the swap s[0] = windows 0..15 of the
            // lane's own read, s[1] = windows 16..31
            auto collect = [&](const sk_v16i &d0, const sk_v16i &d1, int base) {
                uint32_t p0 = 0, p1 = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) p0 = __builtin_amdgcn_alignbit(p0, (uint32_t)d0[i], 31);
#pragma unroll
                for (int i = 0; i < 16; ++i) p1 = __builtin_amdgcn_alignbit(p1, (uint32_t)d1[i], 31);
                const sk_v2u s = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
                step32((s[0] << 16) | s[1], base);
            };
            if (SEG == 2 || (SEG != 3 && !three_blocks)) { // w <= 33: positions base .. base+63, two trips per turn so that
                                 // the fragment registers alternate instead of being copied
                sk_v4i qa0 = load_frag(frag0), qa1 = load_frag(frag1);
                for (int base = 0; base < nwinmax; base += 64) {
                    const sk_v4i qb0 = load_frag(frag0 + base + 32), qb1 = load_frag(frag1 + base + 32);
                    sk_v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, qa0, negT, 0, 0, 0);
                    sk_v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, qa1, negT, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, qb0, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, qb1, d1, 0, 0, 0);
                    collect(d0, d1, base);
                    if (base + 32 >= nwinmax) break;
                    qa0 = load_frag(frag0 + base + 64);
                    qa1 = load_frag(frag1 + base + 64);
                    sk_v16i e0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, qb0, negT, 0, 0, 0);
                    sk_v16i e1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, qb1, negT, 0, 0, 0);
                    e0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, qa0, e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, qa1, e1, 0, 0, 0);
                    collect(e0, e1, base + 32);
                }
            } else { // w in 34..65: positions base .. base+95
                sk_v4i q0 = load_frag(frag0), q1 = load_frag(frag1);
                sk_v4i q0n = load_frag(frag0 + 32), q1n = load_frag(frag1 + 32);
                for (int base = 0; base < nwinmax; base += 32) {
                    const sk_v4i q0f = load_frag(frag0 + base + 64), q1f = load_frag(frag1 + base + 64);
                    sk_v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, q0, negT, 0, 0, 0);
                    sk_v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, q1, negT, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, q0n, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, q1n, d1, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA2, q0f, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA2, q1f, d1, 0, 0, 0);
                    q0 = q0n;
                    q1 = q1n;
                    q0n = q0f;
                    q1n = q1f;
                    collect(d0, d1, base);
                }
            }
        } else {
            // ---- S_0 - T : trim.cpp:31-33
            uint32_t acc = 0;
            {
                int k = 0;
                if (UNIFORM) {
                    for (; 4 * (k + 1) <= wmax; ++k) acc = __builtin_amdgcn_sad_u8(row[k], 0u, acc);
                }
                for (; 4 * k < wmax; ++k) acc = __builtin_amdgcn_sad_u8(first_bytes(row[k], w - 4 * k, 0u), 0u, acc);
            }
            int v = (int)acc - a.craw * w; // sign bit <=> window average below the threshold
            uint32_t lead_lo = row[m];
            for (int base = 0; base < nwinmax; base += 32) {
                uint32_t M = 0;
                const int dw0 = base >> 2;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t y = row[dw0 + k]; // chars leaving the window
                    const uint32_t lead_hi = row[dw0 + k + m + 1];
                    const uint32_t x = __builtin_amdgcn_alignbyte(lead_hi, lead_lo, (uint32_t)sh); // chars entering
                    lead_lo = lead_hi;
                    const int d = (int)(((x | H4) - y) ^ H4); // per byte: x - y as int8 (both < 128)
                    const int t1 = __builtin_amdgcn_sdot4(d, 0x00000001, v, false);
                    const int t2 = __builtin_amdgcn_sdot4(d, 0x00000101, v, false);
                    const int t3 = __builtin_amdgcn_sdot4(d, 0x00010101, v, false);
                    const int t4 = __builtin_amdgcn_sdot4(d, 0x01010101, v, false);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)v, 31);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t1, 31);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t2, 31);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t3, 31);
                    v = t4;
                }
                step32(M, base);
            }
        }
        const bool found5 = a.no5 || i0u != NONE;
        const bool have5 = !a.no5 && i0u != NONE;
        const bool done = i1u != NONE;
        const int i0 = have5 ? (int)i0u : 0, i1 = done ? (int)i1u : 0;

        // ---- the in-window searches: trim.cpp:46-51 and :65-70.
        // 5': first char >= threshold at or after i0; 3': first char < threshold at or after i1.
        // Both exist inside their window (its average is on that side of the threshold), i.e. in
        // the dwords k..k+trips-1.  Addresses do not depend on the data, so the reads pipeline;
        // a hit is tracked as the bit index of its flag (8*pos + 7), NONE until found.
        int five = 0, three = L;
#if SK_TAIL_PRIO
        // the tail of a tile (hit searches, then the refill DMA) runs at raised priority: the sooner
        // a wave gets its buffer back in flight, the fuller the DMA queue of the CU stays
        __builtin_amdgcn_s_setprio(SK_TAIL_PRIO);
#endif
        {
            const int k5 = i0 >> 2, k3 = i1 >> 2;
            const int trips = (wmax + 3) / 4 + 1;
            const uint32_t *r5 = row + k5, *r3 = row + k3;
            // bit index of the first hit RELATIVE to dword k (8*byte + 7 + 32*trip), NONE until found
            uint32_t h5 = ffbl_or_none(ge_flags(r5[0], cthr4) & (~0u << (8 * (i0 & 3))));
            uint32_t h3 = ffbl_or_none((ge_flags(r3[0], cthr4) ^ H4) & (~0u << (8 * (i1 & 3))));
            for (int it = 1; it < trips; ++it) {
                const uint32_t g5 = ge_flags(r5[it], cthr4);
                const uint32_t g3 = ge_flags(r3[it], cthr4) ^ H4;
                const uint32_t rel = 32u * (uint32_t)it; // wave-uniform
                h5 = min(h5, __builtin_elementwise_add_sat(ffbl_or_none(g5), rel));
                h3 = min(h3, __builtin_elementwise_add_sat(ffbl_or_none(g3), rel));
            }
            if (have5 && h5 != NONE) five = 4 * k5 + (int)(h5 >> 3);
            if (done && h3 != NONE) three = 4 * k3 + (int)(h3 >> 3);
        }

        // ---- range error: only if the first bad char is one the reference would have read
        if (__builtin_amdgcn_ballot_w64(bad)) {
            if (bad) {
                const int touched = done ? i1 + w : L;
                int p = INF;
                for (int k = 0; 4 * k < L; ++k) {
                    uint32_t f = keep_first(bad_flags(row[k], min4, hi4), L - 4 * k);
                    if (f) { p = 4 * k + (__builtin_ctz(f) >> 3); break; }
                }
                // (segmented batches in slot order: the caller's read number comes from out_index here only)
                if (p < touched) report_error(errword, (SEG && a.slot_order) ? (uint64_t)out_index[r] : r, p, (int)(int8_t)(tile[(size_t)lane * ts + p]));
            }
        }

        // ---- the N rule: trim.cpp:86-98 (lowercase n: cut before it; only uppercase N: cut = -2)
        if (HAS_SEQ) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (SEQ_SHARES) {
                // the quality scan is over: the same buffer now takes the sequence tile of these reads
                load_tile(seq, buf0, cur);
                wait_vmcnt(0);
            } else {
                // buf0 is free now: start Q(t+1), then retire S(t)
                if (more) tile_to_lds(qual + nxt.off, buf0, next_bytes, lane);
                wait_vmcnt(next_pieces); // older than Q(t+1): S(t)
            }
            const uint32_t *srow = reinterpret_cast<const uint32_t *>((SEQ_SHARES ? buf0 : buf1) + (size_t)lane * ts);
            // 'n' (0x6e) and 'N' (0x4e) differ in bit 5 only: one zero-byte test on (c | 0x20) ^ 'n'
            // flags both, bit 5 of the original byte tells them apart.  nlo = bit index of the first
            // lowercase n (NONE if none), anyN = whether an uppercase N occurs at all.
            uint32_t nlo = NONE, anyN = 0;
            auto n_step = [&](uint32_t x, uint32_t rel) {
                const uint32_t y = (x | 0x20202020u) ^ 0x6e6e6e6eu;
                // exact zero-byte flags (the shorter (y-0x01..)&~y form also flags a 0x01 byte above
                // a zero byte, i.e. an 'o' right after an 'N')
                const uint32_t either = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & H4;
                const uint32_t lower = either & (x << 2);            // bit 5 of the byte moved onto its flag
                nlo = min(nlo, __builtin_elementwise_add_sat(ffbl_or_none(lower), rel));
                anyN |= either ^ lower;
            };
            {
                int k = 0;
                if (UNIFORM || tile_u) {
                    const int full = Lmax >> 2;
                    const uint64_t *srow64 = reinterpret_cast<const uint64_t *>(srow);
                    for (; k + 8 <= full; k += 8) {
                        uint64_t x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[u] = srow64[(k >> 1) + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            n_step((uint32_t)x[u], 32u * (uint32_t)(k + 2 * u));
                            n_step((uint32_t)(x[u] >> 32), 32u * (uint32_t)(k + 2 * u + 1));
                        }
                    }
                }
                for (; 4 * k < Lmax; ++k) n_step(first_bytes(srow[k], L - 4 * k, 0u), 32u * (uint32_t)k);
            }
            if (nlo != NONE) three = (int)(nlo >> 3) - 1;
            else if (anyN) three = -2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (more) {
                if (SEQ_SHARES) { if (nxt.take) load_tile(qual, buf0, nxt); }   // Q(t+1)
                else tile_to_lds(seq + nxt.off, buf1, next_bytes, lane);      // S(t+1)
            }
        } else if (NBUF == 1 && ABLATE != 2 && (!STAGE || (more && !cur_staged))) {
            // single buffer: every LDS read of this tile is done, refill it now -- the cut store
            // below and the other waves of the CU cover the DMA latency.  (Staged kernel: only the
            // ragged last tile of the batch takes this way.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (more && nxt.take) load_tile(qual, buf0, nxt);
        }
#if SK_TAIL_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif

        // ---- trim.cpp:103-108
        if (!scanned || !found5 || (three - five < a.lthr)) {
            five = -1;
            three = -1;
        }
        if (active) out[r] = sk_cut_dev{five, three};
        // this trip's LDS reads are complete (their results were consumed) before the next
        // trip may overwrite the buffer they came from
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur = nxt;
    }
}

template <bool UNIFORM, bool HAS_SEQ, bool MFMA = false, int NBUF = 2, int ABLATE = 0, int SEG = 0>
__global__ void __launch_bounds__(SK_TILE_THREADS, 2)
sk_scan_tile_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                    const uint32_t *__restrict__ lengths, sk_cut_dev *__restrict__ out,
                    unsigned long long *errword, sk_scan_args a, const sk_tile_dev *__restrict__ tiles = nullptr,
                    const uint32_t *__restrict__ out_index = nullptr)
{
    sk_scan_tile_body<UNIFORM, HAS_SEQ, MFMA, NBUF, ABLATE, SEG, 0, false>(qual, seq, lengths, out, errword, a, tiles, out_index, nullptr);
}

// rows at any byte address (RAG): packed uniform batches (matrix path when MFMA) and ragged ones
template <bool UNIFORM, bool HAS_SEQ, bool MFMA>
__global__ void __launch_bounds__(SK_TILE_THREADS, 2)
sk_scan_tile_any_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                        const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lengths,
                        sk_cut_dev *__restrict__ out, unsigned long long *errword, sk_scan_args a)
{
    sk_scan_tile_body<UNIFORM, HAS_SEQ, MFMA, 1, 0, false, 0, true>(qual, seq, lengths, out, errword, a, nullptr, nullptr, offsets);
}

// the register-staged variant (STAGE = KiB pieces per tile = ceil(stride / 16)): the registers of a
// wave hold the scan state and the next tile, three waves per SIMD (12 per CU) at STAGE = 10
template <int STAGE, int ABLATE = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4)))
sk_scan_tile_staged_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                           const uint32_t *__restrict__ lengths, sk_cut_dev *__restrict__ out,
                           unsigned long long *errword, sk_scan_args a, const sk_tile_dev *__restrict__ tiles = nullptr,
                           const uint32_t *__restrict__ out_index = nullptr)
{
    sk_scan_tile_body<true, false, true, 1, ABLATE, false, STAGE, false>(qual, seq, lengths, out, errword, a, tiles, out_index, nullptr);
}

// ------------------------------------------------------------------------------------------
// General kernel: a TEAM of lanes per read (16 or 64), any layout, any length.
//
// Takes what the lane-per-read kernels cannot: rows too long for a 64-read LDS tile.  The reads of
// a wave (4 with teams of 16, 1 with teams of 64) are staged whole into LDS by the entire wave --
// 16 bytes per lane per LDS-DMA, source address per lane, so the image of each read starts on an
// aligned boundary whatever its address in the batch.  Then lane tl of a team owns the c bytes
// [tl*c, tl*c + c) of its read (c a multiple of 4 with c/4 odd: the lanes' dword walks spread over
// the banks) and the windows that START there:
//   1. range check (two v_sad_u8 per dword) and byte sum of its chunk; inclusive scan of the chunk
//      sums over the team: P(x) for every chunk boundary x;
//   2. S_s - T for its first window from the prefix: P(s + w) - P(s), the first taken from the lane
//      w/c chunks up plus a partial chunk sum -- no lane adds up w bytes;
//   3. the windows, 4 per dword of the trailing and the leading stream with byte-parallel arithmetic
//      (the vector-ALU path of the tile kernel), 32 per trip: first >= T, first < T, first < T after
//      the lane's first >= T;
//   4. team min-reductions give i0 and i1; the two in-window searches and the N rule stride the
//      team over dwords; lane 0 of the team stores the cut.
// A read too long for the wave's LDS buffer is scanned straight from global memory by the whole
// wave (scan_read_global: the same algorithm byte by byte; correctness path).
// With a.buf_bytes != 0 the kernel takes only the 64-read tiles sk_scan_tile_any_kernel left.
// ------------------------------------------------------------------------------------------
namespace {

template <int TEAM>
__device__ __forceinline__ int team_min(int v) // teams of 16 lanes are DPP rows
{
    return TEAM == 16 ? row_min(v) : wave_min(v);
}
template <int TEAM>
__device__ __forceinline__ uint32_t team_or(uint32_t v)
{
    return TEAM == 16 ? row_or(v) : wave_or(v);
}
template <int TEAM>
__device__ __forceinline__ uint32_t team_scan_add(uint32_t v, int tl) // inclusive prefix sum over the team
{
    // within a row: row_shr:n reads the lane n to the left, lanes without one add nothing (bound_ctrl: 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    if (TEAM == 64) { // the rows before this lane's: their totals through scalar registers
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                       t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
        const int row = tl >> 4;
        v += row == 0 ? 0u : row == 1 ? t0 : row == 2 ? t0 + t1 : t0 + t1 + t2;
    }
    return v;
}

// one read, the whole wave, from global memory: reference trim.cpp:3-116 with the closed form of this
// file's header, byte by byte.  Returns the cut (the same value in every lane).
template <bool HAS_SEQ>
__device__ __forceinline__ sk_cut_dev scan_read_global(const uint8_t *__restrict__ q, const uint8_t *__restrict__ sq, int L,
                                                    uint64_t r, int lane, const sk_scan_args &a, unsigned long long *errword)
{
    int five = -1, three = -1;
    if (L > 0 && L >= a.lthr) { // trim.cpp:21
        int w = L / 10;
        if (w == 0) w = L;
        const int nwin = L - w + 1;
        const int T = a.craw * w;

        // first bad char of the whole read (lanes stride the bytes, coalesced)
        int pbad = INF;
        for (int j = lane; j < L; j += 64) {
            const int c = (int)(int8_t)q[j];
            if ((c < a.qmin || c > a.qmax) && pbad == INF) pbad = j;
        }
        pbad = wave_min(pbad);

        // lane owns windows [s, e): seeds the sum, then rolls it (trim.cpp:76-80)
        const int per = (nwin + 63) >> 6;
        const int s = lane * per;
        const int e = min(nwin, s + per);
        int fa = INF, fb = INF, fc = INF; // first >=T, first <T, first <T after fa
        if (s < e) {
            int tot = 0;
            for (int j = 0; j < w; ++j) tot += q[s + j];
            for (int i = s; i < e; ++i) {
                if (tot >= T) {
                    if (fa == INF) fa = i;
                } else {
                    if (fb == INF) fb = i;
                    if (fa != INF && fc == INF) fc = i;
                }
                if (i + 1 < e) tot += (int)q[i + w] - (int)q[i];
            }
        }
        const int i0 = a.no5 ? -1 : wave_min(fa);
        const bool found5 = a.no5 || i0 != INF;
        int cand = INF;
        if (a.no5) cand = fb;
        else if (i0 != INF && s < e) cand = (s > i0) ? fb : (fa == i0 ? fc : INF);
        const int i1 = wave_min(cand);
        const bool done = found5 && i1 != INF;

        five = 0;
        three = L;
        if (!a.no5 && i0 != INF) { // trim.cpp:46-51
            int hit = INF;
            for (int j = lane; j < w && hit == INF; j += 64)
                if ((int)q[i0 + j] >= a.cthr_raw) hit = i0 + j;
            five = wave_min(hit);
            if (five == INF) five = 0;
        }
        if (done) { // trim.cpp:65-70
            int hit = INF;
            for (int j = lane; j < w && hit == INF; j += 64)
                if ((int)q[i1 + j] < a.cthr_raw) hit = i1 + j;
            three = wave_min(hit);
            if (three == INF) three = L;
        }
        const int touched = done ? i1 + w : L;
        if (pbad < touched) {
            if (lane == 0) report_error(errword, r, pbad, (int)(int8_t)q[pbad]);
        }
        if (HAS_SEQ) { // trim.cpp:86-98
            int ni = INF, Ni = INF;
            for (int j = lane; j < L; j += 64) {
                const uint8_t c = sq[j];
                if (c == 'n' && ni == INF) ni = j;
                if (c == 'N' && Ni == INF) Ni = j;
            }
            ni = wave_min(ni);
            Ni = wave_min(Ni);
            if (ni != INF) three = ni - 1;
            else if (Ni != INF) three = -2;
        }
        if (!found5 || (three - five < a.lthr)) { // trim.cpp:103-108
            five = -1;
            three = -1;
        }
    }
    return sk_cut_dev{five, three};
}

} // namespace

template <int TEAM, bool HAS_SEQ>
__global__ void __launch_bounds__(64)
sk_scan_team_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                    const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lengths,
                    sk_cut_dev *__restrict__ out, unsigned long long *errword, sk_scan_args a)
{
    static_assert(TEAM == 16 || TEAM == 64, "teams of 16 or 64 lanes");
    constexpr int RPW = 64 / TEAM; // reads per wave
    constexpr bool SKIP = TEAM == 64; // whole-wave teams: prefix table + skip-ahead window search (see do_slot)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x; // single-wave workgroups
    const int g = lane / TEAM, tl = lane % TEAM;
    // a read's buffer: rbuf bytes (the longest read the LDS path takes + what the lead stream may read
    // past it), the teams' buffers 80 bytes apart on top so that their rows do not share banks
    const uint32_t rbuf = a.team_rbuf, pitch = rbuf + 80u;
    const uint32_t *row32 = reinterpret_cast<const uint32_t *>(lds + (uint32_t)g * pitch);
    const uint8_t *rowb = lds + (uint32_t)g * pitch;
    const uint64_t n_slots = (a.n_reads + RPW - 1) / RPW;
    const uint64_t batch_end = a.n_reads ? rag_batch_end(offsets, lengths, a) : 0;
    const uint32_t min4 = splat((uint32_t)a.qmin), max4 = splat((uint32_t)a.qmax);
    const uint32_t hi4 = splat((uint32_t)(127 - a.qmax));
    const uint32_t cthr4 = splat((uint32_t)a.cthr);
    const int range = a.qmax - a.qmin;

    // known = the caller already holds this lane's read (start, length): the hand-over mode took them from the
    // tile probe; otherwise they are loaded here
    auto do_slot = [&](uint64_t slot, bool known, uint64_t o_known, int L_known) {
        const uint64_t r = slot * RPW + g;
        const bool valid = r < a.n_reads;
        uint64_t o;
        int L;
        if (known) {
            o = o_known;
            L = valid ? L_known : 0;
        } else {
            const uint64_t rc = min(r, a.n_reads - 1);
            uint64_t e;
            if (offsets) {
                o = offsets[rc];
                e = offsets[rc + 1];
            } else {
                o = rc * a.stride;
                e = o + (lengths ? lengths[rc] : a.read_len);
            }
            L = (valid && e >= o) ? (int)min(e - o, (uint64_t)SK_MAX_READ_LEN_DEV) : 0;
        }
        const bool big = L > (int)a.team_maxlen; // not through LDS
        const bool scan = L > 0 && L >= a.lthr && !big; // trim.cpp:21

        // ---- the reads of this wave into LDS, read after read, the whole wave copying: lane i of a
        // piece fetches the 16 bytes at read offset 16*(c0 + i), wherever they are in the batch
        auto stage = [&](const uint8_t *base) {
#pragma unroll
            for (int gg = 0; gg < RPW; ++gg) {
                const int Lg = __builtin_amdgcn_readlane(scan ? L : 0, gg * TEAM);
                if (Lg == 0) continue;
                const uint64_t og = readlane_u64(o, gg * TEAM);
                uint8_t *dst = lds + (uint32_t)gg * pitch;
                const uint8_t *src = base + og;
                const uint32_t nch = ((uint32_t)Lg + 15u) >> 4;
                const bool all_inside = og + 16u * nch <= batch_end; // wave-uniform: no chunk of this read can leave the batch
                for (uint32_t c0 = 0; c0 < nch; c0 += 64u) {
                    const uint32_t so = 16u * (c0 + (uint32_t)lane);
                    if (c0 + (uint32_t)lane < nch) {
                        if (all_inside || og + so + 16u <= batch_end) {
                            __builtin_amdgcn_global_load_lds((gptr_t)(src + so), (lptr_t)(dst + c0 * 16u), 16, 0, SK_DMA_AUX);
                        } else { // the batch ends inside this chunk: byte by byte
                            for (uint32_t j = 0; j < 16u && og + so + j < batch_end; ++j) dst[so + j] = src[so + j];
                        }
                    }
                }
            }
            wait_vmcnt(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        stage(qual);

        int five = -1, three = -1;
        int w = L / 10; // trim.cpp:8
        if (w == 0) w = L; // trim.cpp:30
        const int nwin = scan ? L - w + 1 : 0;
        const int T = a.craw * w;
        // this lane's chunk of the read: c bytes, c/4 odd (conflict-free dword walks); SKIP: c/16 odd
        // (conflict-free 16-byte walks)
        const int c4 = !scan ? 1 : SKIP ? 4 * (((((L + TEAM - 1) / TEAM) + 15) >> 4) | 1) : ((((L + TEAM - 1) / TEAM + 3) >> 2) | 1);
        const int c = 4 * c4;
        const int s = tl * c;
        const int sdw = s >> 2;

        const bool has_win = scan && s < nwin;
        const int we = min(s + c, nwin); // this lane's windows: [s, we)
        uint32_t fa = NONE, fb = NONE, fc = NONE; // first >= T, first < T, first < T after fa (window indices)
        bool bad = false;
        if (SKIP) {
            // ---- whole-wave teams (long reads): prefix sums + skip-ahead instead of walking every window.
            // P16[k] = sum of the read's bytes before position 16k, so S_i = P(i + w) - P(i) for ANY i costs two
            // table reads and two partial 16-byte sums; and since one step changes a window sum by at most 255,
            // a lane at S_i - T = v < 0 can jump ceil(-v / 255) windows ahead without missing the first
            // S >= T (and v / 255 + 1 ahead when looking for the first S < T).  Window sums of long reads sit
            // far from the threshold almost everywhere (w * |Q - q|), so a lane evaluates a few dozen windows
            // instead of its whole chunk: 6 instead of 48 VALU per dword of the read, exact for every input.
            uint32_t *P16 = reinterpret_cast<uint32_t *>(lds + rbuf);
            const sk_v4u *row128 = reinterpret_cast<const sk_v4u *>(rowb);
            const int sg = s >> 4;
            const int ngroups = scan ? max(0, min(c4 >> 2, (L - s + 15) >> 4)) : 0;
            uint32_t sad = 0, run = 0;
            for (int gi = 0; gi < ngroups; ++gi) { // 1. range check (trim.cpp:129) + local prefix of the chunk
                const sk_v4u x = row128[sg + gi];
                const int nval = L - (s + 16 * gi);
                P16[sg + gi] = run;
                if (nval >= 16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        sad = __builtin_amdgcn_sad_u8(x[u], min4, sad);
                        sad = __builtin_amdgcn_sad_u8(x[u], max4, sad);
                        run = __builtin_amdgcn_sad_u8(x[u], 0u, run);
                    }
                } else { // the read ends inside this group
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t xq = first_bytes(x[u], nval - 4 * u, min4);
                        sad = __builtin_amdgcn_sad_u8(xq, min4, sad);
                        sad = __builtin_amdgcn_sad_u8(xq, max4, sad);
                        run = __builtin_amdgcn_sad_u8(first_bytes(x[u], nval - 4 * u, 0u), 0u, run);
                    }
                }
            }
            bad = scan && sad != (uint32_t)(16 * ngroups * range); // fillers are legal chars
            const uint32_t incl = team_scan_add<TEAM>(run, tl);
            const uint32_t excl = incl - run;
            for (int gi = 0; gi < ngroups; ++gi) atomicAdd(&P16[sg + gi], excl); // local -> global prefix (ds_add_u32)
            // P(L) when L is a multiple of 16: one entry past the last group
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (tl == 0 && scan) P16[(L + 15) >> 4] = total;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto P = [&](int x) -> int { // sum of the bytes before position x, 0 <= x <= L
                const int gi = x >> 4, r = x & 15;
                uint32_t p = P16[gi];
                const sk_v4u d = row128[gi];
#pragma unroll
                for (int u = 0; u < 4; ++u) p = __builtin_amdgcn_sad_u8(first_bytes(d[u], r - 4 * u, 0u), 0u, p);
                return (int)p;
            };
            auto SmT = [&](int i) -> int { return P(i + w) - P(i) - T; }; // sign bit <=> window average below the threshold
            if (has_win) { // 2./3. trim.cpp:34-81: at most two searches per lane
                int j = s, v = SmT(s);
                if (v >= 0) {
                    fa = (uint32_t)s;
                    for (;;) { // the first window below the threshold after it
                        j += v / 255 + 1;
                        if (j >= we) break;
                        v = SmT(j);
                        if (v < 0) { fb = fc = (uint32_t)j; break; }
                    }
                } else {
                    fb = (uint32_t)s;
                    for (;;) { // the first window at or above the threshold
                        j += (-v + 254) / 255;
                        if (j >= we) break;
                        v = SmT(j);
                        if (v >= 0) { fa = (uint32_t)j; break; }
                    }
                    if (fa != NONE) {
                        for (;;) { // and the first one below it again
                            j += v / 255 + 1;
                            if (j >= we) break;
                            v = SmT(j);
                            if (v < 0) { fc = (uint32_t)j; break; }
                        }
                    }
                }
            }
        } else {
        // ---- 1. range check + chunk sum (trim.cpp:129 and the prefix of 31-33)
        uint32_t sad = 0, csum = 0;
        {
            // whole dwords of the chunk that lie inside the read need no masking: four loads in flight
            const int inner = scan ? min(c4, max(0, (L - s) >> 2)) : 0;
            int k = 0;
            for (; k + 4 <= inner; k += 4) {
                uint32_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = row32[sdw + k + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sad = __builtin_amdgcn_sad_u8(x[u], min4, sad);
                    sad = __builtin_amdgcn_sad_u8(x[u], max4, sad);
                    csum = __builtin_amdgcn_sad_u8(x[u], 0u, csum);
                }
            }
            for (; k < c4; ++k) { // the rest of the chunk: the read may end inside it
                uint32_t x = min4;
                int nval = 0;
                if (scan) {
                    x = row32[sdw + k];
                    nval = L - (s + 4 * k);
                }
                const uint32_t xq = first_bytes(x, nval, min4);
                sad = __builtin_amdgcn_sad_u8(xq, min4, sad);
                sad = __builtin_amdgcn_sad_u8(xq, max4, sad);
                csum = __builtin_amdgcn_sad_u8(first_bytes(x, nval, 0u), 0u, csum);
            }
        }
        // every dword visited contributes 4 * range when clean (fillers are legal chars)
        bad = scan && sad != (uint32_t)(4 * c4 * range);
        const uint32_t incl = team_scan_add<TEAM>(csum, tl);

        // ---- 2. S_s - T for this lane's first window: P(s + w) - P(s) - T
        const int dq = w / c, rem = w - dq * c;
        const int kq = tl + dq; // the chunk position s + w lies in
        const uint32_t below = (uint32_t)__shfl((int)incl, g * TEAM + min(max(kq - 1, 0), TEAM - 1), 64);
        uint32_t part = 0;
        {
            const int remmax = __builtin_amdgcn_readfirstlane(wave_max(has_win ? rem : 0));
            const int bdw = (kq * c) >> 2;
            for (int j = 0; 4 * j < remmax; ++j)
                if (has_win && 4 * j < rem) part = __builtin_amdgcn_sad_u8(first_bytes(row32[bdw + j], rem - 4 * j, 0u), 0u, part);
        }
        int v = (int)((kq >= 1 ? below : 0u) + part) - (int)(incl - csum) - T; // sign bit <=> window average below the threshold

        // ---- 3. the lane's windows [s, we), 32 per trip: trim.cpp:34-81 without the breaks
        {
            const int mytrips = has_win ? (we - s + 31) >> 5 : 0;
            const int tripmax = __builtin_amdgcn_readfirstlane(wave_max(mytrips));
            const int ldw = (s + w) >> 2;
            const uint32_t sh = (uint32_t)(w & 3); // s is a multiple of 4
            uint32_t lead_lo = has_win ? row32[ldw] : 0u;
            for (int tr = 0; tr < tripmax; ++tr) {
                if (tr < mytrips) {
                    uint32_t M = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int dwi = tr * 8 + k;
                        const uint32_t y = row32[sdw + dwi];              // chars leaving the window
                        const uint32_t lead_hi = row32[ldw + dwi + 1];
                        const uint32_t x = __builtin_amdgcn_alignbyte(lead_hi, lead_lo, sh); // chars entering
                        lead_lo = lead_hi;
                        const int d = (int)(((x | H4) - y) ^ H4); // per byte: x - y as int8 (both < 128)
                        const int t1 = __builtin_amdgcn_sdot4(d, 0x00000001, v, false);
                        const int t2 = __builtin_amdgcn_sdot4(d, 0x00000101, v, false);
                        const int t3 = __builtin_amdgcn_sdot4(d, 0x00010101, v, false);
                        const int t4 = __builtin_amdgcn_sdot4(d, 0x01010101, v, false);
                        M = __builtin_amdgcn_alignbit(M, (uint32_t)v, 31);
                        M = __builtin_amdgcn_alignbit(M, (uint32_t)t1, 31);
                        M = __builtin_amdgcn_alignbit(M, (uint32_t)t2, 31);
                        M = __builtin_amdgcn_alignbit(M, (uint32_t)t3, 31);
                        v = t4;
                    }
                    // bit (31 - j) of M: window base + j is below the threshold
                    const int base = s + 32 * tr;
                    const int nv = we - base;
                    const uint32_t vmask = nv >= 32 ? ~0u : ~(~0u >> nv); // nv >= 1 here
                    const uint32_t lt = M & vmask, ge = ~M & vmask;
                    fa = min(fa, __builtin_elementwise_add_sat(ffbh_or_none(ge), (uint32_t)base));
                    fb = min(fb, __builtin_elementwise_add_sat(ffbh_or_none(lt), (uint32_t)base));
                    // windows of this trip strictly after fa: the low (base + 31 - fa) bits
                    const uint32_t width = __builtin_elementwise_sub_sat((uint32_t)(base + 31), fa);
                    const uint32_t low = (1u << (width & 31u)) - 1u;
                    const uint32_t after = lt & (width >= 32u ? ~0u : low);
                    fc = min(fc, __builtin_elementwise_add_sat(ffbh_or_none(after), (uint32_t)base));
                }
            }
        }

        }

        // ---- 4. the team's windows: trim.cpp:42 and :61
        const int fai = fa == NONE ? INF : (int)fa, fbi = fb == NONE ? INF : (int)fb, fci = fc == NONE ? INF : (int)fc;
        const int i0 = a.no5 ? INF : team_min<TEAM>(fai);
        const bool have5 = !a.no5 && i0 != INF;
        const bool found5 = a.no5 || i0 != INF;
        int cand = INF;
        if (a.no5) cand = fbi;
        else if (i0 != INF && has_win) cand = (s > i0) ? fbi : (fai == i0 ? fci : INF);
        const int i1 = team_min<TEAM>(cand);
        const bool done = found5 && i1 != INF;

        five = 0;
        three = L;
        if (have5) { // trim.cpp:46-51: the first char >= threshold at or after i0 (one exists inside the window)
            int hit = INF;
            const int d0 = i0 >> 2, ndw = ((i0 & 3) + w + 3) >> 2;
            for (int d = tl; d < ndw && hit == INF; d += TEAM) {
                uint32_t f = ge_flags(row32[d0 + d], cthr4);
                if (d == 0) f &= ~0u << (8 * (i0 & 3));
                if (f) hit = 4 * (d0 + d) + (__builtin_ctz(f) >> 3);
            }
            hit = team_min<TEAM>(hit);
            five = hit == INF ? 0 : hit;
        }
        if (done) { // trim.cpp:65-70
            int hit = INF;
            const int d0 = i1 >> 2, ndw = ((i1 & 3) + w + 3) >> 2;
            for (int d = tl; d < ndw && hit == INF; d += TEAM) {
                uint32_t f = ge_flags(row32[d0 + d], cthr4) ^ H4;
                if (d == 0) f &= ~0u << (8 * (i1 & 3));
                if (f) hit = 4 * (d0 + d) + (__builtin_ctz(f) >> 3);
            }
            hit = team_min<TEAM>(hit);
            three = hit == INF ? L : hit;
        }

        // ---- range error: only if the first bad char is one the reference would have read
        if (__builtin_amdgcn_ballot_w64(bad)) {
            int pb = INF;
            if (bad) {
                for (int k = 0; k < c4 && s + 4 * k < L && pb == INF; ++k) {
                    const uint32_t f = keep_first(bad_flags(row32[sdw + k], min4, hi4), L - (s + 4 * k));
                    if (f) pb = s + 4 * k + (__builtin_ctz(f) >> 3);
                }
            }
            pb = team_min<TEAM>(pb);
            const int touched = done ? i1 + w : L;
            if (scan && pb < touched && tl == 0) report_error(errword, r, pb, (int)(int8_t)rowb[pb]);
        }

        // ---- the N rule: trim.cpp:86-98, the sequences through the same buffers
        if (HAS_SEQ) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stage(seq);
            uint32_t nlo = NONE, anyN = 0; // bit index of the first lowercase n; any uppercase N
            const int c4m = __builtin_amdgcn_readfirstlane(wave_max(scan ? c4 : 0));
            for (int k = 0; k < c4m; ++k) {
                if (scan && k < c4 && s + 4 * k < L) {
                    const uint32_t x = first_bytes(row32[sdw + k], L - (s + 4 * k), 0u);
                    const uint32_t y = (x | 0x20202020u) ^ 0x6e6e6e6eu;
                    const uint32_t either = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & H4;
                    const uint32_t lower = either & (x << 2); // bit 5 of the byte moved onto its flag
                    nlo = min(nlo, __builtin_elementwise_add_sat(ffbl_or_none(lower), (uint32_t)(8 * (s + 4 * k))));
                    anyN |= either ^ lower;
                }
            }
            const int nl = team_min<TEAM>(nlo == NONE ? INF : (int)(nlo >> 3));
            anyN = team_or<TEAM>(anyN);
            if (nl != INF) three = nl - 1;
            else if (anyN) three = -2;
        }
        if (!scan || !found5 || (three - five < a.lthr)) { // trim.cpp:103-108
            five = -1;
            three = -1;
        }
        if (valid && !big && tl == 0) out[r] = sk_cut_dev{five, three};
        // every LDS read of this slot is done before the next slot's DMA may overwrite the buffers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // ---- reads too long for the LDS buffer: the whole wave, one after the other, from global memory
        if (__builtin_amdgcn_ballot_w64(big)) {
#pragma unroll
            for (int gg = 0; gg < RPW; ++gg) {
                if (!__builtin_amdgcn_readlane((int)big, gg * TEAM)) continue;
                const int Lg = __builtin_amdgcn_readlane(L, gg * TEAM);
                const uint64_t og = readlane_u64(o, gg * TEAM);
                const uint64_t rg = slot * RPW + gg;
                const sk_cut_dev cut = scan_read_global<HAS_SEQ>(qual + og, HAS_SEQ ? seq + og : nullptr, Lg, rg, lane, a, errword);
                if (lane == 0) out[rg] = cut;
            }
        }
    };

    if (a.buf_bytes) {
        // only the 64-read tiles sk_scan_tile_any_kernel left (the same test as there) -- if it left any:
        // it has put this scan's number into the word after the error word for every tile it skipped
        if (*reinterpret_cast<volatile unsigned long long *>(errword + 1) != a.scan_id) return;
        // Runs of 8 consecutive reads are dealt to the waves (so that the reads of one left-over tile spread
        // over the device).  A wave asks the question for the tile its run lies in; the probe leaves read
        // 64*tile + l's start and length in lane l, so the run's reads need no further offset loads.
        // (A ticket counter in global memory instead of the fixed deal, one atomic per read, measured slower:
        // 0.82 against 0.69 ms on 64 200 reads of 1-30 kb.)
        constexpr uint64_t RUN = 8;
        const uint64_t n_runs = (a.n_reads + RUN - 1) / RUN;
        for (uint64_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
            const sk_rag_tile pr = rag_probe((run * RUN) >> 6, lane, offsets, lengths, a);
            if (rag_tile_fits(pr, a.buf_bytes)) continue;
            for (uint64_t slot = run * RUN / RPW; slot < (run + 1) * RUN / RPW && slot < n_slots; ++slot) {
                const int idx = (int)((slot * RPW + (uint64_t)g) & 63u); // this lane's read within the tile
                const uint32_t ro = (uint32_t)__shfl((int)pr.rowoff, idx, 64);
                const int len = __shfl(pr.len, idx, 64);
                do_slot(slot, true, pr.start + ro, len);
            }
        }
    } else {
        for (uint64_t slot = blockIdx.x; slot < n_slots; slot += gridDim.x) do_slot(slot, false, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// launchers (host side of this translation unit)
// ------------------------------------------------------------------------------------------
namespace {

// The dynamic-LDS ceiling of a kernel is a property of (device, function) in the HIP runtime; it is
// raised ONCE per pair to the CU's 160 KiB and every launch then passes its own size.  (Setting it per
// launch to that launch's size raced between host threads scanning batches of different strides on one
// device: A sets 80 KiB, B sets 10 KiB, A's launch fails.)  Also caches the kernel's register count.
struct kernel_facts {
    hipError_t status = hipSuccess;
    int regs = 0;
};
template <typename K>
kernel_facts prepare_kernel(K kern)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, kernel_facts> seen;
    int device = 0;
    (void)hipGetDevice(&device);
    const void *fn = reinterpret_cast<const void *>(kern);
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find({device, fn});
    if (it != seen.end()) return it->second;
    kernel_facts f;
    f.status = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_PER_CU);
    hipFuncAttributes fa;
    if (f.status == hipSuccess && hipFuncGetAttributes(&fa, fn) == hipSuccess) f.regs = fa.numRegs;
    if (f.status == hipSuccess) seen[{device, fn}] = f; // a failure is retried by the next launch
    return f;
}

int tile_nbuf_default()
{
    // diagnostic override (tools/ablate.py, A/B runs): SK_TILE_NBUF=1|2
    static const int v = [] {
        const char *e = getenv("SK_TILE_NBUF");
        return (e && (*e == '1' || *e == '2')) ? *e - '0' : SK_TILE_NBUF_DEFAULT;
    }();
    return v;
}

int tile_stage_default()
{
    // diagnostic override (A/B runs): SK_TILE_STAGE=0 keeps uniform batches on the LDS-DMA kernel
    static const int v = [] {
        const char *e = getenv("SK_TILE_STAGE");
        return (e && *e == '0') ? 0 : 1;
    }();
    return v;
}

template <typename K>
hipError_t launch_tile_kernel(K kern, int bufs, const uint8_t *qual, const uint8_t *seq, const uint32_t *lengths,
                              sk_cut_dev *out, unsigned long long *errword, const sk_scan_args *a, int cu_count,
                              int per_cu_cap, hipStream_t stream, bool by_registers = false)
{
    // single-wave workgroups (waves never synchronise with each other); as many per CU as the
    // 160 KiB of LDS and the 32-wave limit allow
    const uint32_t lds_bytes = (uint32_t)bufs * (64u * a->stride + SK_TILE_SLACK);
    if (lds_bytes > SK_LDS_PER_CU) return hipErrorInvalidValue;
    int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
    if (per_cu > 16) per_cu = 16;
    const kernel_facts facts = prepare_kernel(kern);
    if (facts.status != hipSuccess) return facts.status;
    if (by_registers) {
        // the staged kernels are bounded by their registers, not by LDS (the grid is persistent, so
        // workgroups beyond what fits would only run as a second round): waves per SIMD = the 512
        // registers of a lane's file over the kernel's count (allocated in eights), four SIMDs
        const int regs = facts.regs > 0 ? facts.regs : 160;
        int fits = 4 * (512 / ((regs + 7) & ~7));
        if (fits < 4) fits = 4;
        if (getenv("SK_DEBUG_LAUNCH")) fprintf(stderr, "[sk] staged kernel: %d registers -> %d workgroups per CU\n", regs, fits);
        if (per_cu > fits) per_cu = fits;
    }
    if (per_cu_cap > 0 && per_cu > per_cu_cap) per_cu = per_cu_cap;
    const uint64_t n_tiles = (a->n_reads + 63) >> 6;
    uint64_t grid = (uint64_t)cu_count * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, qual, seq, lengths, out, errword, *a,
                       (const sk_tile_dev *)nullptr, (const uint32_t *)nullptr);
    return hipGetLastError();
}

} // namespace

// does a uniform batch of this shape take the register-staged kernel?
extern "C" __attribute__((visibility("hidden"))) int sk_tile_is_staged(uint32_t stride, uint32_t read_len, int has_seq)
{
    const uint32_t wu = read_len / 10 ? read_len / 10 : read_len;
    const uint32_t pieces = (64u * stride + 1023u) >> 10;
    return !has_seq && read_len > 0 && wu <= 65 && tile_nbuf_default() == 1 && tile_stage_default() && stride >= 16 &&
           stride % 8 == 0 && pieces >= SK_STAGE_MIN && pieces <= SK_STAGE_MAX;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_tile(const uint8_t *qual, const uint8_t *seq, const uint32_t *lengths,
                                     sk_cut_dev *out, unsigned long long *errword, const sk_scan_args *a,
                                     int cu_count, hipStream_t stream)
{
    const bool uniform = lengths == nullptr;
    const bool has_seq = a->truncn != 0;
    // the matrix path: one window width for the whole batch, band within three 32-position blocks
    const uint32_t wu = a->read_len / 10 ? a->read_len / 10 : a->read_len;
    const bool mfma = uniform && wu <= 65 && a->read_len > 0;
    const int nbuf = tile_nbuf_default();
#define SK_GO(KERN, BUFS) launch_tile_kernel(KERN, BUFS, qual, seq, lengths, out, errword, a, cu_count, 0, stream)
    if (has_seq && nbuf == 1) {
        if (mfma) return SK_GO((sk_scan_tile_kernel<true, true, true, 1>), 1);
        if (uniform) return SK_GO((sk_scan_tile_kernel<true, true, false, 1>), 1);
        return SK_GO((sk_scan_tile_kernel<false, true, false, 1>), 1);
    }
    if (has_seq) {
        if (mfma) return SK_GO((sk_scan_tile_kernel<true, true, true, 2>), 2);
        if (uniform) return SK_GO((sk_scan_tile_kernel<true, true, false, 2>), 2);
        return SK_GO((sk_scan_tile_kernel<false, true, false, 2>), 2);
    }
    if (nbuf == 1) {
        const uint32_t pieces = (64u * a->stride + 1023u) >> 10;
        if (mfma && sk_tile_is_staged(a->stride, a->read_len, 0)) {
#define SK_STAGED(P) case P: return launch_tile_kernel(sk_scan_tile_staged_kernel<P>, 1, qual, seq, lengths, out, errword, a, cu_count, 0, stream, true)
            switch (pieces) {
                SK_STAGED(5); SK_STAGED(6); SK_STAGED(7); SK_STAGED(8); SK_STAGED(9); SK_STAGED(10);
            default: break;
            }
#undef SK_STAGED
        }
        if (mfma) return SK_GO((sk_scan_tile_kernel<true, false, true, 1>), 1);
        if (uniform) return SK_GO((sk_scan_tile_kernel<true, false, false, 1>), 1);
        return SK_GO((sk_scan_tile_kernel<false, false, false, 1>), 1);
    }
    if (mfma) return SK_GO((sk_scan_tile_kernel<true, false, true, 2>), 2);
    if (uniform) return SK_GO((sk_scan_tile_kernel<true, false, false, 2>), 2);
    return SK_GO((sk_scan_tile_kernel<false, false, false, 2>), 2);
#undef SK_GO
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_seg(const uint8_t *qual, const uint8_t *seq, const sk_tile_dev *tiles,
                                    const uint32_t *out_index, sk_cut_dev *out, unsigned long long *errword,
                                    const sk_scan_args *a, const sk_seg_class *classes, uint32_t n_classes,
                                    int cu_count, hipStream_t stream)
{
    // One launch per class of tiles (a run of the tile array): the LDS buffer of a wave is sized for
    // the widest row of THAT run, so the short reads of a mixed batch get their 16 waves per CU, and
    // runs without windows wider than 33 run the two-block matrix loop.  No class table: one run.
    const sk_seg_class whole = {0u, a->n_tiles, a->stride, 1u};
    if (!classes || n_classes == 0) {
        classes = &whole;
        n_classes = 1;
    }
    for (uint32_t c = 0; c < n_classes; ++c) {
        const sk_seg_class &k = classes[c];
        if (k.n_tiles == 0) continue;
        if ((uint64_t)k.first_tile + k.n_tiles > a->n_tiles || k.max_stride == 0 || k.max_stride % 8 != 0) return hipErrorInvalidValue;
        const uint32_t lds_bytes = 64u * k.max_stride + SK_TILE_SLACK;
        if (lds_bytes > SK_LDS_PER_CU) return hipErrorInvalidValue;
        int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
        if (per_cu > 16) per_cu = 16;
        uint64_t grid = (uint64_t)cu_count * per_cu;
        const uint64_t chunks = ((uint64_t)k.n_tiles + 3) / 4; // a wave takes 4 consecutive tiles at a time (SEG_CHUNK)
        if (grid > chunks) grid = chunks;
        sk_scan_args as = *a;
        as.buf_bytes = lds_bytes;
        as.n_tiles = k.n_tiles;
        auto launch = [&](auto kern) {
            const kernel_facts facts = prepare_kernel(kern);
            if (facts.status != hipSuccess) return facts.status;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, qual, seq,
                               (const uint32_t *)nullptr, out, errword, as, tiles + k.first_tile, out_index);
            return hipGetLastError();
        };
        hipError_t e;
        if (a->truncn) e = k.wide ? launch(sk_scan_tile_kernel<true, true, true, 1, 0, 3>) : launch(sk_scan_tile_kernel<true, true, true, 1, 0, 2>);
        else e = k.wide ? launch(sk_scan_tile_kernel<true, false, true, 1, 0, 3>) : launch(sk_scan_tile_kernel<true, false, true, 1, 0, 2>);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Batches whose rows start at any byte address.  offsets == lengths == nullptr: packed uniform
// batch (a->stride any value >= a->read_len).  Otherwise ragged (offsets, or stride + lengths).
// a->buf_bytes = the LDS bytes of a wave (sized by the caller for the longest read it expects); the
// tiles this kernel leaves (image too large for the buffer; never in a packed uniform batch) are taken
// by sk_launch_wave with the same a->buf_bytes.
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_any(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                    const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                    const sk_scan_args *a, int cu_count, hipStream_t stream)
{
    const bool uniform = offsets == nullptr && lengths == nullptr;
    const uint32_t lds_bytes = a->buf_bytes;
    if (lds_bytes == 0 || lds_bytes > SK_LDS_PER_CU) return hipErrorInvalidValue;
    int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
    if (per_cu > 16) per_cu = 16;
    const uint64_t n_tiles = (a->n_reads + 63) >> 6;
    uint64_t grid = (uint64_t)cu_count * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid == 0) return hipSuccess;
    auto launch = [&](auto kern) {
        const kernel_facts facts = prepare_kernel(kern);
        if (facts.status != hipSuccess) return facts.status;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, qual, seq, offsets, lengths, out,
                           errword, *a);
        return hipGetLastError();
    };
    const uint32_t wu = a->read_len / 10 ? a->read_len / 10 : a->read_len;
    const bool mfma = uniform && wu <= 65 && a->read_len > 0;
    if (a->truncn) {
        if (mfma) return launch(sk_scan_tile_any_kernel<true, true, true>);
        if (uniform) return launch(sk_scan_tile_any_kernel<true, true, false>);
        return launch(sk_scan_tile_any_kernel<false, true, true>);
    }
    if (mfma) return launch(sk_scan_tile_any_kernel<true, false, true>);
    if (uniform) return launch(sk_scan_tile_any_kernel<true, false, false>);
    return launch(sk_scan_tile_any_kernel<false, false, true>);
}

// diagnostic: the uniform, no-seq tile kernel with part of its work removed (tools/ablate.py).
// mode = ablate + 10*(vector-ALU path instead of matrix path) + 100*(single buffer)
extern "C" hipError_t sk_launch_tile_ablate(int mode, const uint8_t *qual, sk_cut_dev *out, unsigned long long *errword,
                                            const sk_scan_args *a, int cu_count, int waves, int per_cu, hipStream_t stream)
{
    (void)waves;
#define SK_GO(KERN, BUFS) launch_tile_kernel(KERN, BUFS, qual, nullptr, nullptr, out, errword, a, cu_count, per_cu, stream)
    switch (mode) {
    case 0: return SK_GO((sk_scan_tile_kernel<true, false, true, 2, 0>), 2);
    case 1: return SK_GO((sk_scan_tile_kernel<true, false, true, 2, 1>), 2);
    case 2: return SK_GO((sk_scan_tile_kernel<true, false, true, 2, 2>), 2);
    case 10: return SK_GO((sk_scan_tile_kernel<true, false, false, 2, 0>), 2);
    case 12: return SK_GO((sk_scan_tile_kernel<true, false, false, 2, 2>), 2);
    case 100: return SK_GO((sk_scan_tile_kernel<true, false, true, 1, 0>), 1);
    case 101: return SK_GO((sk_scan_tile_kernel<true, false, true, 1, 1>), 1);
    case 102: return SK_GO((sk_scan_tile_kernel<true, false, true, 1, 2>), 1);
    case 110: return SK_GO((sk_scan_tile_kernel<true, false, false, 1, 0>), 1);
    case 200: return SK_GO((sk_scan_tile_staged_kernel<10, 0>), 1);
    case 201: return SK_GO((sk_scan_tile_staged_kernel<10, 1>), 1);
    case 202: return SK_GO((sk_scan_tile_staged_kernel<10, 2>), 1);
    default: return hipErrorInvalidValue;
    }
#undef SK_GO
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_team(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                     const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                     const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream)
{
    // max_len = the longest read the caller expects (0 = unknown).  Teams of 16 lanes (4 reads per
    // wave) up to 2 KiB, the whole wave beyond; reads longer than the buffer sized here still come out
    // right, from global memory (scan_read_global).
    if (a->n_reads == 0) return hipSuccess;
    if (max_len == 0) max_len = 32768;
    static const uint64_t team16_max = [] { const char *e = getenv("SK_TEAM16_MAX"); return e ? (uint64_t)atoll(e) : 4096ull; }();
    const int team = max_len <= team16_max ? 16 : 64;
    const int rpw = 64 / team;
    uint64_t cap = max_len;
    const uint64_t cap_max = (uint64_t)(SK_LDS_PER_CU / 2) / rpw - 1024; // at least two waves per CU
    if (cap > cap_max) cap = cap_max;
    sk_scan_args at = *a;
    at.team_maxlen = (uint32_t)cap;
    // what the lead stream and the 32-window trips may read past the read: a chunk + 32 windows + slack
    at.team_rbuf = (uint32_t)((cap + cap / team + 4 + 32 + SK_TILE_SLACK + 15) & ~(uint64_t)15);
    // whole-wave teams keep a prefix table beside the read: 4 bytes per 16 (sk_scan_team_kernel, SKIP)
    const uint32_t lds_bytes = (uint32_t)rpw * (at.team_rbuf + 80u) + (team == 64 ? (at.team_rbuf >> 2) + 64u : 0u);
    int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
    static const int wave_cap = [] { const char *e = getenv("SK_TEAM_WAVES"); return e ? atoi(e) : 16; }();
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) return hipErrorInvalidValue;
    const uint64_t n_slots = (a->n_reads + rpw - 1) / rpw;
    uint64_t grid = (uint64_t)cu_count * per_cu;
    if (grid > n_slots) grid = n_slots;
    if (grid == 0) return hipSuccess;
    auto launch = [&](auto kern) {
        const kernel_facts facts = prepare_kernel(kern);
        if (facts.status != hipSuccess) return facts.status;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, qual, seq, offsets, lengths, out,
                           errword, at);
        return hipGetLastError();
    };
    if (team == 16) return a->truncn ? launch(sk_scan_team_kernel<16, true>) : launch(sk_scan_team_kernel<16, false>);
    return a->truncn ? launch(sk_scan_team_kernel<64, true>) : launch(sk_scan_team_kernel<64, false>);
}

// ------------------------------------------------------------------------------------------
// measurement aid: what a read-only stream of this buffer gets on this device (sk_probe_read_bandwidth)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sk_read_probe_kernel(const sk_v4u *__restrict__ src, size_t n16, uint32_t *sink)
{
    constexpr int UNROLL = 4;
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256 * UNROLL;
    sk_v4u acc = {0, 0, 0, 0};
    for (; i + 256 * (UNROLL - 1) < n16; i += step) {
        sk_v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    // keeps the loads alive; quality bytes never fold to this value
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && n16 == 1) sink[threadIdx.x & 1] = 1;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_read_probe(const void *buf, size_t bytes, uint32_t *sink, int cu_count,
                                           hipStream_t stream)
{
    hipLaunchKernelGGL(sk_read_probe_kernel, dim3((unsigned)cu_count * 32u), dim3(256), 0, stream,
                       reinterpret_cast<const sk_v4u *>(buf), bytes / 16, sink);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// pair classification: reference src/trim_paired.cpp:543-567 over the cuts of a scan (mates at 2k, 2k+1).
// One 16-byte load per pair and lane; the four class counts of a wave come from ballots, one lane adds them.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sk_pair_count_kernel(const sk_v4i *__restrict__ cuts, uint64_t n_pairs, uint8_t *__restrict__ classes,
                                                            unsigned long long *counters)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long c_both = 0, c_first = 0, c_second = 0, c_none = 0; // per wave, kept by every lane
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256; k0 < n_pairs; k0 += stride) {
        const uint64_t k = k0 + threadIdx.x;
        int cls = -1;
        if (k < n_pairs) {
            const sk_v4i c = __builtin_nontemporal_load(cuts + k); // {five1, three1, five2, three2}
            const bool r1 = c[1] >= 0, r2 = c[3] >= 0;             // src/trim_paired.cpp:500,502
            cls = r1 ? (r2 ? 0 : 1) : (r2 ? 2 : 3);
            if (classes) classes[k] = (uint8_t)cls;
        }
        c_both += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 0));
        c_first += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 1));
        c_second += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 2));
        c_none += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 3));
    }
    if ((threadIdx.x & 63) == 0) {
        if (c_both) atomicAdd(counters + 0, c_both);
        if (c_first) atomicAdd(counters + 1, c_first);
        if (c_second) atomicAdd(counters + 2, c_second);
        if (c_none) atomicAdd(counters + 3, c_none);
    }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_pair_count(const sk_cut_dev *cuts, uint64_t n_pairs, uint8_t *classes,
                                           unsigned long long *counters, int cu_count, hipStream_t stream)
{
    if (n_pairs == 0) return hipSuccess;
    uint64_t grid = (n_pairs + 255) / 256;
    if (grid > (uint64_t)cu_count * 16) grid = (uint64_t)cu_count * 16;
    hipLaunchKernelGGL(sk_pair_count_kernel, dim3((unsigned)grid), dim3(256), 0, stream, reinterpret_cast<const sk_v4i *>(cuts), n_pairs,
                       classes, counters);
    return hipGetLastError();
}
