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
// The kernels, by translation unit:
//   sk_kernels.hip (this file)  the LANE-per-read tile kernels.  One wavefront per 64-read tile, single-wave
//                         workgroups, up to 16 per CU.  The tile goes HBM -> LDS with global_load_lds_dwordx4
//                         (LDS-DMA, nt policy: coalesced, no VGPR round trip), through the wave's registers
//                         (sk_scan_tile_staged_kernel: rows of 72..160 bytes), or RE-STRIDED by a per-lane
//                         source address (sk_scan_tile_any_kernel: packed and ragged batches).  Uniform-length
//                         tiles take their window sums from the integer matrix pipe (band(w) x Q as two
//                         v_mfma_i32_32x32x32_i8 per 32 windows x 32 reads, the B operand read straight out of
//                         the LDS rows) and spend one v_alignbit per window on the vector ALU to collect the
//                         signs of S_i - T; mixed-length tiles walk their rows 4 windows per dword with
//                         byte-parallel arithmetic (v_alignbyte, signed byte differences, v_dot4_i32_i8,
//                         v_alignbit).  Range check: two v_sad_u8 per dword.
//   sk_team.hip           the general kernel: a TEAM of 16 lanes or the whole wave per read, the read staged
//                         whole into LDS; long reads, whatever the lane-per-read tiles cannot hold.
//   sk_aux.hip            read-bandwidth probe, pair classification.
//   sk_kernel_common.h    what they share (byte-parallel helpers, DPP reductions, the ragged-tile probe ...).
// HBM-bound byte streaming (algorithmic bytes: L + 8 per read, 2L + 8 with -n); the MFMAs are a
// way to take instructions off the vector ALU, not the bound.

#include "sk_kernel_common.h"

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
#ifndef SK_WIDE_QUIET
#define SK_WIDE_QUIET 1
#endif
#define SK_STAGE_MIN 5  /* register-staged kernels exist for tiles of 5..20 KiB: row strides 72..320 */
#define SK_STAGE_MAX 20

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
// SORT (a ragged batch regrouped by sk_sort.hip; RAG with a row start per lane): the tiles come from the list of this
// workgroup's XCD -- {window, first slot, rows, window width} -- and a tile's rows are the reads perm[] names: equal in
// window width, not in length, anywhere inside their window of SK_SORT_WINDOW consecutive reads.  One band per tile,
// so the matrix path; lengths, window counts and the masks of the range check stay per lane.  (tiles = the lists,
// out_index = perm, lengths = the lists' tile counts.)
// WIDE (uniform MEDIUM reads, 505 bytes to a few thousand: rows too long for 64 of them to share a wave's LDS buffer
// with enough other waves on the CU): tiles of 32 reads, a PAIR of lanes per read -- lane n and lane n + 32 hold the two
// halves of read n's 16-byte operand fragments, which is how one v_mfma_i32_32x32x32_i8 wants 32 reads anyway, and each
// gets 16 of the read's 32 window sums; a v_permlane32_swap hands both lanes all 32 sign bits and both run the read's
// state (everything after the matrix path is per read and done by both lanes alike; lane n stores).  Windows of any
// width from 32 up: the positions a window of 32 consecutive windows covers fall into a first block (triangular band),
// dm = w / 32 - 1 blocks that EVERY one of the 32 windows covers whole, and two last blocks (the band's other edge).
// The whole blocks need no band -- an all-ones matrix -- and their sum is carried from step to step in the accumulator
// that also holds -T: entering block in (+ones), leaving block out (-ones).  Five MFMAs per 32 windows x 32 reads
// whatever the width; two new fragments per step.
// WIDE == 2: tiles of 16 reads, FOUR lanes per read -- the 32 columns of the matrix are 16 reads x the two halves of
// their windows (column n: read n mod 16, windows [0, H) for n < 16 and [H, ..) for the others, H a multiple of 32),
// each half with its own state, the two joined when the windows are through.  Half the LDS per wave, so twice the
// waves on a CU (a wave's MFMAs run under another wave's vector work), for the same MFMA count per byte.
template <bool UNIFORM, bool HAS_SEQ, bool MFMA, int NBUF, int ABLATE, int SEG, int STAGE, bool RAG, bool SORT = false, int WIDE = 0, int WSTAGE = 0>
__device__ __forceinline__ void
sk_scan_tile_body(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                  const uint32_t *__restrict__ lengths, sk_cut_dev *__restrict__ out,
                  unsigned long long *errword, const sk_scan_args &a, const sk_tile_dev *__restrict__ tiles,
                  const uint32_t *__restrict__ out_index, const uint64_t *__restrict__ offsets)
{
    static_assert(!SORT || (RAG && !UNIFORM && MFMA), "regrouped ragged batches: re-strided tiles, matrix path");
    static_assert(!WIDE || (RAG && UNIFORM && MFMA && !SORT), "medium reads: uniform batches, re-strided tiles, matrix path");
    static_assert(!WSTAGE || (WIDE && !HAS_SEQ), "the medium-read tiles' register stage: without -n");
    constexpr uint32_t ROWS = WIDE == 2 ? 16u : WIDE ? 32u : 64u; // reads per tile
    constexpr int RSHIFT = WIDE == 2 ? 4 : WIDE ? 5 : 6;
    static_assert(!RAG || (NBUF == 1 && !SEG && STAGE == 0 && ABLATE == 0), "re-strided tiles: one buffer, LDS-DMA");
    static_assert(!MFMA || UNIFORM || RAG, "the matrix path needs one window width per tile");
    // MIXED (ragged batches): per-lane lengths in general, but a tile whose 64 reads have ONE length --
    // every tile of the usual fixed-length run handed over as offsets -- takes the matrix path and the
    // uniform row walks; the other tiles walk the vector-ALU path.  Decided per tile, wave-uniformly.
    constexpr bool MIXED = MFMA && !UNIFORM;
    static_assert(STAGE == 0 || (UNIFORM && !HAS_SEQ && NBUF == 1 && (SEG == 0 || SEG == 2)), "register staging: uniform tiles, one buffer");
    static_assert(!SEG || (UNIFORM && MFMA && NBUF == 1), "segmented batches run the uniform matrix path, one buffer");
    // -n: NBUF == 2 keeps the quality and the sequence tile in two buffers (8 waves per CU);
    // NBUF == 1 runs both through ONE buffer, one after the other (16 waves per CU)
    constexpr int LDS_BUFS = NBUF;
    constexpr bool SEQ_SHARES = HAS_SEQ && NBUF == 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    const int rlane = WIDE == 2 ? (lane & 15) : WIDE ? (lane & 31) : lane; // the tile row (read) this lane works on
    const int part = WIDE == 2 ? (lane >> 4) & 1 : 0;                          // WIDE == 2: which half of the read's windows
    // readfirstlane: tells the compiler this is one value per wave, so that the tile index and
    // everything derived from it (addresses, piece counts, loop and switch conditions) live in
    // SGPRs and branch on the scalar unit instead of being carried through the vector ALU
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waves_per_block = blockDim.x >> 6;
    const uint32_t stride = a.stride; // SEG: the largest row stride of the batch (sizes the LDS buffers)
    const uint32_t buf_bytes = (RAG || SEG) ? a.buf_bytes : 64u * stride + SK_TILE_SLACK;
    uint8_t *buf0 = lds + (size_t)wave * LDS_BUFS * buf_bytes;
    uint8_t *buf1 = LDS_BUFS > 1 ? buf0 + buf_bytes : buf0;

    // SORT: the list of XCD x = blockIdx mod 8 (single-wave workgroups), shared by the workgroups b = x, x + 8, ...
    const uint32_t xcd = SORT ? blockIdx.x & 7u : 0u;
    const unsigned long long *const slist = reinterpret_cast<const unsigned long long *>(tiles) + (size_t)xcd * a.n_tiles * 4u;
    const uint64_t *const perm = reinterpret_cast<const uint64_t *>(out_index) + (size_t)xcd * a.n_tiles * 64u;
    const uint64_t n_tiles = SORT ? (uint64_t)min(scalar_load(lengths + xcd), a.n_tiles) : SEG ? (uint64_t)a.n_tiles : (a.n_reads + ROWS - 1) >> RSHIFT;
    uint64_t wave_global = SORT ? (uint64_t)(blockIdx.x >> 3) : (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint64_t wave_count = SORT ? (uint64_t)((gridDim.x - xcd + 7u) >> 3) : (uint64_t)gridDim.x * waves_per_block;
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
    // TABLE (segmented batches, regrouped ragged batches: the window width changes from tile to tile): the band matrix
    // is not computed (~150 vector instructions) but LOADED from the context's table of all widths (three 16-byte
    // loads per lane, from L2) -- a tile ahead, into `band_next`, while the tile before is scanned; installed at the top
    // of the turn that needs it.  Invariant: whenever a turn's width differs from wu, band_next holds that width's band.
    constexpr bool TABLE = SEG != 0 || SORT;
    struct band3 {
        sk_v4i b0, b1, b2;
    };
    band3 band_next{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto band_fetch = [&](int w) -> band3 {
        typedef const __attribute__((address_space(1))) sk_v4i *gv4;
        gv4 tb = (gv4)(uintptr_t)a.band_table + ((uint32_t)w * 3u * 64u + (uint32_t)lane);
        band3 b;
        b.b0 = tb[0];
        b.b1 = tb[64];
        b.b2 = (SEG == 2) ? b.b1 : tb[128]; // (SEG == 2: no window reaches into a third block)
        return b;
    };
    auto width_of = [](int len) -> int { return len / 10 ? len / 10 : len; }; // trim.cpp:8,30
    // the scalars of a length; the threshold vector when the width changed
    auto set_scalars = [&](int len) {
        const int w_old = wu;
        Lu = len;
        scan_u = Lu > 0 && Lu >= a.lthr;  // reference trim.cpp:21
        wu = width_of(Lu);
        // windows wider than 33 reach into a third 32-position block (never in the staged kernels: rows
        // <= 320 bytes).  Segmented launches fix it at compile time (SEG = 2: every tile of the launch
        // has w <= 33; SEG = 3: three blocks for every tile, right for any w <= 65): one MFMA loop
        // instead of two in the kernel, fewer registers
        three_blocks = MFMA && STAGE == 0 && (SEG == 2 ? false : SEG == 3 ? true : wu > 33);
        if (MFMA && (!TABLE || wu != w_old)) {
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
    auto install_band = [&](const band3 &b) {
        bandA0 = b.b0;
        bandA1 = b.b1;
        bandA2 = b.b2;
    };
    auto set_length = [&](int len) {
        set_scalars(len);
        if (MFMA) {
            // ---- constants of the matrix path, computed (uniform batches: once per launch; ragged batches in input
            // order: at the rare change of length): sk_band_dword says which byte is which
            int ln = lane;
            // inside the tile loop: without the barrier the compiler hoists the 48 per-byte position constants out
            // of the loop and pins a register to each
            if (MIXED) asm volatile("" : "+v"(ln));
            const int k0 = (ln >> 5) * 16; // the first position (relative to 32*b) this lane's bytes multiply
            // WIDE: the band's far edge lies in the blocks behind the dm whole ones
            const int far = WIDE ? 32 * (wu / 32) : 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bandA0[j] = (int)sk_band_dword(ln, wu, k0 + 4 * j);
                bandA1[j] = (int)sk_band_dword(ln, wu, k0 + 4 * j + far);
                bandA2[j] = (int)sk_band_dword(ln, wu, k0 + 4 * j + far + 32);
            }
        }
    };
    if (!TABLE) set_length((int)a.read_len);

    const uint64_t batch_end = RAG ? rag_batch_end(offsets, lengths, a) : 0;
    // (segmented batches: the descriptor of a tile is loaded TWO turns ahead, and by a VECTOR load -- every lane the
    // same address, the values back into scalar registers a turn later.  A scalar load shares its counter with the LDS
    // reads of the scan, so wherever it is issued the next wait for an LDS read waits for it as well, and a
    // descriptor that comes from HBM costs the wave ~1 us per tile: 5-8 % measured against the uniform kernel on the
    // same data.  The vector load goes out once the turn's tile has landed and is collected, in order, by the next
    // turn's wait for its tile.)
    auto fetch_desc = [&](uint64_t tt) -> sk_tile_dev {
        uint32_t at = (uint32_t)tt * (uint32_t)sizeof(sk_tile_dev); // (< 2^26 tiles: n_reads < 2^32)
        asm volatile("" : "+v"(at)); // a vector offset: global_load, counted by vmcnt
        return *reinterpret_cast<const sk_tile_dev *>(reinterpret_cast<const uint8_t *>(tiles) + at);
    };
    auto uniform_desc = [&](const sk_tile_dev &d) -> sk_tile_dev {
        sk_tile_dev u;
        u.byte_off = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(d.byte_off >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)d.byte_off);
        u.slot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.slot0);
        u.stride = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.stride);
        const uint32_t rl = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)d.rows | ((uint32_t)d.read_len << 16)));
        u.rows = (uint16_t)rl;
        u.read_len = (uint16_t)(rl >> 16);
        u.reserved = 0;
        return u;
    };
    auto probe = [&](uint64_t tt, const sk_tile_dev *ahead = nullptr) -> sk_tile_view {
        sk_tile_view v;
        v.take = true;
        v.uni = false;
        v.rowoff = 0;
        if (SEG) {
            const sk_tile_dev d = ahead ? uniform_desc(*ahead) : tiles[tt];
            v.off = d.byte_off;
            v.ts = d.stride;
            v.rows = d.rows;
            v.bytes = v.rows * v.ts;
            v.len = (int)d.read_len;
            v.r = d.slot0; // probe_index() turns it into this lane's slot of out[]
        } else if (RAG && UNIFORM) {
            v.off = (tt << RSHIFT) * stride;
            v.rows = (uint32_t)min((uint64_t)ROWS, a.n_reads - (tt << RSHIFT));
            v.ts = rag_pitch<true>(a.read_len);
            v.bytes = (v.rows - 1u) * stride + a.read_len;
            v.len = (int)a.read_len;
            v.r = (tt << RSHIFT) + rlane;
        } else if (SORT) {
            // (sort_issue / sort_finish below: the loads go out a tile ahead)
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

    // SORT: a tile's descriptor (32 bytes, every lane the same address) and its 64 entries (8 bytes per lane) lie where
    // the tile's number says.  Both are VECTOR loads, issued once the turn's tile has landed and looked at when the
    // turn's scan is over -- the scan covers them.  (The first version took descriptor, window bounds and entries one
    // after the other at the start of every scan, 2-3 us of exposed latency per tile; the second loaded the descriptor
    // through the scalar cache at the top of the turn and waited for it there: a scalar load shares its counter with
    // the LDS reads, there is no place in a scan where it could stay in flight.)
    struct sort_raw {
        sk_v4u d0, d1; // the descriptor's eight dwords (d1[3] unused)
        uint64_t e;    // this lane's entry
    };
    auto sort_issue = [&](uint64_t tt) -> sort_raw {
        sort_raw q;
        uint32_t at = (uint32_t)tt * 32u; // (< 2^27 tiles per list)
        asm volatile("" : "+v"(at));     // a vector offset: global_load, counted by vmcnt
        const uint8_t *dp = reinterpret_cast<const uint8_t *>(slist) + at;
        q.d0 = *reinterpret_cast<const sk_v4u *>(dp);
        q.d1 = *reinterpret_cast<const sk_v4u *>(dp + 16);
        q.e = perm[tt * 64u + (uint32_t)lane];
        return q;
    };
    auto first = [](uint32_t x) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    int sort_lmax = 0, sort_lmin = 0, sort_w = 0; // of the tile being scanned (wave-uniform, from its descriptor)
    int next_lmax = 0, next_lmin = 0, next_w = 0;
    auto sort_finish = [&](const sort_raw &q) -> sk_tile_view {
        sk_tile_view v;
        const uint32_t d00 = first(q.d0[0]), d01 = first(q.d0[1]), d02 = first(q.d0[2]), d03 = first(q.d0[3]);
        const uint32_t d10 = first(q.d1[0]), d11 = first(q.d1[1]), d12 = first(q.d1[2]);
        const uint64_t widx = d00;
        v.rows = (d01 >> 16) & 0xffu;
        next_w = (int)(d01 >> 24);
        next_lmax = (int)(d12 & 0xffffu);
        next_lmin = (int)(d12 >> 16);
        v.off = ((uint64_t)d03 << 32) | d02;
        v.bytes = d11 ? 0xffffffffu : d10;
        v.rowoff = (uint32_t)lane < v.rows ? (uint32_t)q.e : 0u;
        v.len = (uint32_t)lane < v.rows ? (int)((uint32_t)(q.e >> 32) & 0xffffu) : 0;
        v.r = widx * SK_SORT_WINDOW + (uint32_t)(q.e >> 48);
        v.ts = rag_pitch<false>((uint32_t)next_lmax);
        v.take = true; // (the sort sends a batch with a read too long for the tiles to the other kernels)
        v.uni = true;  // here: ONE window width
        return v;
    };

    // segmented batches: a tile's descriptor is a scalar load (not counted by vmcnt: it can be issued before the
    // wait for the tile); v.r becomes this lane's slot.  Cuts go out in slot order (one coalesced stream), or are
    // scattered back to the caller's read order through out_index: that index is a vector load issued once the tile
    // has landed and collected when the scan is over, just before the refill goes out -- the scan covers its latency.
    // (It is an asm load because the compiler, given the choice, waits for it on the spot: 3.45 against 4.2 TB/s.)
    auto probe_index = [&](sk_tile_view &v) { v.r = (uint64_t)((uint32_t)v.r + min((uint32_t)lane, v.rows - 1u)); };
    const bool scatter = SEG && !a.slot_order;
    auto index_issue = [&](uint64_t slot) -> uint32_t {
        uint32_t idx;
        const uint32_t *src = out_index + slot;
        asm volatile("global_load_dword %0, %1, off" : "=v"(idx) : "v"(src) : "memory");
        return idx;
    };
    auto index_settle = [&](uint32_t &idx) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(idx)::"memory"); };

    // WIDE: every full tile of a launch has the same shape, so a lane's source offsets relative to its tile -- what the
    // re-striding loader spends ~12 vector instructions a piece on -- are computed ONCE: WOFF of them (the register stage
    // loads from them, the LDS-DMA loader of the other medium-read kernels issues its pieces from them).
    typedef uint32_t sk_v4u_a4 __attribute__((ext_vector_type(4), aligned(1)));
    constexpr int WOFF = WIDE ? (WSTAGE ? WSTAGE : 32) : 0;
    uint32_t wso[WOFF ? WOFF : 1];
    sk_v4u wst[WSTAGE ? WSTAGE : 1];
    const uint32_t wpieces = WIDE ? (ROWS * (rag_pitch<true>(a.read_len) / 16u) + 63u) >> 6 : 0u;
    if (WIDE) {
        const uint32_t cpr = rag_pitch<true>(a.read_len) / 16u, qd = 64u / cpr, rd = 64u % cpr;
        const uint32_t lim = (ROWS - 1u) * stride + a.read_len - 1u;
        uint32_t rr = (uint32_t)lane / cpr, cc = (uint32_t)lane % cpr;
#pragma unroll
        for (int p = 0; p < WOFF; ++p) {
            wso[p] = min(__umul24(rr, stride) + 16u * cc, lim);
            cc += rd;
            rr += qd;
            if (cc >= cpr) {
                cc -= cpr;
                ++rr;
            }
        }
    }

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
        // (a tile whose last byte is followed by 16 more of the batch: no chunk of it can leave the caller's buffer)
        const bool all_safe = v.off + (uint64_t)lim + 1u + GRAN <= batch_end; // wave-uniform
        if (WIDE && !WSTAGE && all_safe && v.rows == ROWS && wpieces <= (uint32_t)WOFF) {
            // a full tile inside the batch: the pieces from the offsets computed at the start of the launch
#pragma unroll
            for (int p = 0; p < WOFF; ++p)
                if ((uint32_t)p < wpieces) dma_piece<true, false>(src + wso[p], dst + (uint32_t)p * (64u * GRAN));
            return;
        }
        // one piece: its chunks start `ro` bytes into the tile (a row start per lane) + GRAN * cc
        auto piece = [&](uint32_t p, uint32_t ro, uint32_t cc_p) {
            const uint32_t so = min(ro + GRAN * cc_p, lim);
            // a chunk may reach past its tile: harmless inside the batch, but the last chunks of the
            // batch's last tile(s) must not leave the caller's buffer -- those few lanes copy their
            // bytes one by one instead
            const bool inside = all_safe || v.off + so + GRAN <= batch_end;
            if (__builtin_expect(all_safe || __builtin_amdgcn_ballot_w64(!inside) == 0, 1)) {
                dma_piece<true, SORT>(src + so, dst + p * (64u * GRAN));
            } else {
                if (inside) {
                    dma_piece<true, SORT>(src + so, dst + p * (64u * GRAN));
                } else {
                    for (uint32_t j = 0; j < GRAN && v.off + so + j < batch_end; ++j)
                        dst[p * (64u * GRAN) + (uint32_t)lane * GRAN + j] = src[so + j];
                }
            }
        };
        auto advance = [&]() {
            cc += rd;
            rr += qd;
            if (cc >= cpr) {
                cc -= cpr;
                ++rr;
            }
        };
        if (SORT) {
            // a row start per lane, fetched from the lane that holds the row: FOUR pieces' worth of ds_bpermute go out
            // before the first is waited for (one at a time, each piece sat out the LDS round trip: 19 of them for a
            // 301-base tile)
            for (uint32_t p = 0; p < cpr; p += 4) {
                uint32_t ro[4], cq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    ro[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(min(rr, 63u) << 2), (int)v.rowoff);
                    cq[u] = cc;
                    advance();
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (p + (uint32_t)u < cpr) piece(p + (uint32_t)u, ro[u], cq[u]);
            }
        } else {
            const uint32_t pieces = WIDE == 2 ? (cpr + 3u) >> 2 : WIDE ? (cpr + 1u) >> 1 : cpr; // (32 / 16 rows: the last piece's upper lanes fetch the tile's last bytes again)
            for (uint32_t p = 0; p < pieces; ++p) {
                uint32_t ro;
                if (UNIFORM) ro = WIDE ? __umul24(rr, stride) : rr * stride; // (WIDE: stride < 2^24, wide_takes())
                else if (v.uni) ro = rr * (uint32_t)__builtin_amdgcn_readfirstlane(v.len); // equal lengths: rows len apart
                else ro = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rr << 2), (int)v.rowoff);
                piece(p, ro, cc);
                advance();
            }
        }
    };
    auto load_tile = [&](const uint8_t *base, uint8_t *dst, const sk_tile_view &v) {
        if (RAG) rag_dma(base, dst, v);
        else tile_to_lds(base + v.off, dst, v.bytes, lane);
    };

    // WSTAGE (medium-read tiles without -n whose image has at most WSTAGE 1 KiB pieces; rows at any byte address -- the
    // device's unaligned access mode takes 16-byte loads anywhere, as the LDS-DMA loader's do): the NEXT
    // tile waits in the wave's registers -- plain 16-byte loads at the re-striding loader's addresses, in flight while
    // this tile is scanned, written to LDS (same image) at the top of the next turn.  Every full tile of the launch has
    // the same shape, so a lane's source offsets relative to its tile are computed ONCE (the DMA loader spends ~12 vector
    // instructions a piece on them, tile after tile); the wave no longer waits for its tile after every scan (43 % of
    // its cycles at 1 000 bases), and the loads are plain ones.  The batch's last tiles (fewer rows, or chunks that could
    // leave the caller's buffer) come in by LDS-DMA as before.
    auto wstage_ok = [&](const sk_tile_view &v) -> bool { return WSTAGE && v.rows == ROWS && v.off + (uint64_t)v.bytes + 16u <= batch_end; };
    auto wstage_load = [&](const sk_tile_view &v) {
        const uint8_t *src = qual + v.off;
#pragma unroll
        for (int p = 0; p < WSTAGE; ++p) wst[p] = __builtin_nontemporal_load(reinterpret_cast<const sk_v4u_a4 *>(src + wso[p]));
    };
    auto wstage_write = [&]() {
#pragma unroll
        for (int p = 0; p < WSTAGE; ++p)
            if ((uint32_t)p < wpieces) *reinterpret_cast<sk_v4u *>(buf0 + (uint32_t)p * 1024u + (uint32_t)lane * 16u) = wst[p];
    };
    bool w_cur = false, w_nxt = false;

    // register stage: piece p of a FULL tile (64 rows; 64*stride bytes, a multiple of 512) is the
    // 16 bytes per lane at p KiB; STAGE = the number of pieces, the last one may be a half (its
    // upper lanes repeat the tile's last 16 bytes: same data to the same place, no predication).
    // The last tile of a batch, if it is not full, comes in by LDS-DMA like in the unstaged kernel.
    // Segmented batches (SEG): STAGE is the most a tile of the launch needs; the pieces past a tile's end repeat its
    // last 16 bytes like the half piece above (no branches: with them the compiler no longer counts its waits and every
    // write waits for the loads just issued -- 2.4 against 4.8 TB/s); tiles with fewer than 64 rows come in by LDS-DMA.
    sk_v4u stage[STAGE ? STAGE : 1];
    const uint32_t full_bytes = 64u * stride;
    auto stage_off = [&](int p, uint32_t bytes) -> uint32_t {
        const uint32_t off = (uint32_t)p * 1024u + (uint32_t)lane * 16u;
        return (SEG || p == STAGE - 1) ? min(off, bytes - 16u) : off;
    };

    // Which tiles a wave takes: tile t, then t + (number of waves), ...: the launch sweeps the batch front to back (taking
    // a length-sorted list from both ends at once, for balanced tile sizes per CU, is 10-15 % slower).  Segmented batches
    // can take SEG_CHUNK consecutive tiles at a time (seg_chunk_shift; round 2's default was 4: with single steps every
    // wave met a new length at every tile and REBUILT the band matrix, ~150 instructions against ~600 for the scan).
    // Since the band comes from a table, prefetched, single steps are the default again: chunks cost 3-5 % at long rows.
    // (SORT: one tile at a time -- with chunks of 4 a list's 256 waves have 7 windows open at once, 10 MB, and the XCD's
    // L2 no longer holds the lines two neighbours share: 1187 against 946 MB fetched per scan of 790 MB)
    const uint64_t SEG_CHUNK = SORT ? 1ull : 1ull << a.seg_chunk_shift;
    auto next_tile = [&](uint64_t tt) -> uint64_t {
        if (SEG || SORT) return ((tt + 1) & (SEG_CHUNK - 1)) != 0 ? tt + 1 : tt + 1 + (wave_count - 1) * SEG_CHUNK;
        return tt + wave_count;
    };
    uint64_t t = (SEG || SORT) ? wave_global * SEG_CHUNK : wave_global;
    if (t >= n_tiles) return;

    // prologue: Q(t) [and S(t)] in flight
    sk_tile_view cur = SORT ? sort_finish(sort_issue(t)) : probe(t), nxt = cur;
    if (TABLE) band_next = band_fetch(SORT ? next_w : width_of(cur.len)); // (wu == 0: the first turn installs it)
    sort_raw raw{{0, 0, 0, 0}, {0, 0, 0, 0}, 0};
    if (SEG) probe_index(cur);
    bool cur_staged = STAGE && cur.rows == 64u;
    if (cur_staged) {
        const uint32_t cb = SEG ? cur.bytes : full_bytes;
#pragma unroll
        for (int p = 0; p < STAGE; ++p)
            stage[p] = __builtin_nontemporal_load(reinterpret_cast<const sk_v4u *>(qual + cur.off + stage_off(p, cb)));
    } else if (wstage_ok(cur)) {
        wstage_load(cur);
        w_cur = true;
    } else if (cur.take) {
        load_tile(qual, buf0, cur);
    }
    if (HAS_SEQ && !SEQ_SHARES) tile_to_lds(seq + cur.off, buf1, cur.bytes, lane);
    int parity = 0; // NBUF == 2: which buffer holds Q(t)
    // the next tile's view is taken AFTER this tile has landed wherever taking it loads something
    // (descriptor and out_index of a segmented batch, offsets / lengths): the wait for the tile is a
    // vmcnt(0), and a load issued just before it would put its whole latency on every tile
    constexpr bool PROBE_EARLY = NBUF == 2 || STAGE != 0 || ABLATE != 0 || SEG != 0;

    sk_tile_dev ahead{};
    if (SEG) ahead = fetch_desc(min(next_tile(t), n_tiles - 1));
    for (; t < n_tiles; t = next_tile(t)) {
        const uint64_t tn = next_tile(t);
        const bool more = tn < n_tiles;
        if (PROBE_EARLY && more) nxt = probe(tn, SEG ? &ahead : nullptr);
        const uint32_t ts = cur.ts;  // this tile's row pitch in LDS
        const uint64_t r = cur.r;
        const uint32_t cur_bytes = cur.bytes;
        if (SEG && cur.len != Lu) {
            if (width_of(cur.len) != wu) install_band(band_next);
            set_scalars(cur.len);
        }
        const uint32_t next_bytes = (PROBE_EARLY && more) ? nxt.bytes : 0u;
        const int next_pieces = (PROBE_EARLY && more) ? tile_pieces(next_bytes) : 0;
        const uint8_t *tile;

        const bool active = (uint32_t)rlane < cur.rows;
        const int Lv = UNIFORM ? 0 : cur.len; // mixed lengths: this lane's length (0 past the end of the batch)
        bool tile_u = false; // MIXED: this tile's reads have one length (and a window the matrix path takes)
        if (MIXED && SORT) {
            // one window width w for the tile's reads, in its descriptor: any length with that width names the band
            sort_lmax = next_lmax;
            sort_lmin = next_lmin;
            sort_w = next_w;
            tile_u = sort_w > 0;
            if (tile_u && sort_w != wu) {
                install_band(band_next);
                set_scalars(sort_lmax);
            }
        } else if (MIXED) {
            const int l0 = __builtin_amdgcn_readfirstlane(cur.len);
            tile_u = cur.uni && l0 > 0 && l0 / 10 <= 65;
            if (tile_u && l0 != Lu) set_length(l0);
        }

        if (SORT) {
            tile = buf0;
            wait_vmcnt(0); // Q(t)
            if (more) raw = sort_issue(tn); // looked at when this turn's scan is over
        } else if (SEQ_SHARES || RAG) {
            tile = buf0;
            if (WSTAGE && w_cur) wstage_write(); // (the compiler waits for the registers)
            else wait_vmcnt(0);                  // Q(t)
        } else if (HAS_SEQ) {
            tile = buf0;
            // outstanding, oldest first: Q(t), S(t) [, store(t-1) before them]
            wait_vmcnt(tile_pieces(cur_bytes));
        } else if (STAGE) {
            tile = buf0;
            const bool next_staged = more && nxt.rows == 64u && ABLATE != 2;
            if (SEG) {
                // all the writes, then all the loads: with the branches in between the compiler no longer counts its
                // waits, and a piece's write would wait for the loads just issued (measured: 2.4 against 4.8 TB/s).
                // A tile of fewer than 64 rows (the last of its length) came in by LDS-DMA; the tile after it may
                // be a staged one again.
                if (cur_staged && next_staged) {
                    const uint8_t *nsrc = qual + nxt.off;
#pragma unroll
                    for (int p = 0; p < STAGE; ++p) {
                        *reinterpret_cast<sk_v4u *>(buf0 + stage_off(p, cur_bytes)) = stage[p];
                        stage[p] = __builtin_nontemporal_load(reinterpret_cast<const sk_v4u *>(nsrc + stage_off(p, next_bytes)));
                    }
                } else {
                    if (cur_staged) {
#pragma unroll
                        for (int p = 0; p < STAGE; ++p) *reinterpret_cast<sk_v4u *>(buf0 + stage_off(p, cur_bytes)) = stage[p];
                    } else {
                        wait_vmcnt(0);
                    }
                    if (next_staged) {
                        const uint8_t *nsrc = qual + nxt.off;
#pragma unroll
                        for (int p = 0; p < STAGE; ++p)
                            stage[p] = __builtin_nontemporal_load(reinterpret_cast<const sk_v4u *>(nsrc + stage_off(p, next_bytes)));
                    }
                }
            } else if (cur_staged && (ABLATE != 2 || t == wave_global)) {
                // piece by piece: into the LDS buffer, and the register is reloaded at once with the
                // same piece of the next tile (of the first KiB of this tile again if there is no full
                // next tile: loads nobody uses keep the code free of branches and its wait counts exact)
                const uint8_t *nsrc = qual + (next_staged ? nxt.off : cur.off);
                const uint32_t keep = next_staged ? ~0u : 1023u; // no next tile: every piece re-reads the first KiB
#pragma unroll
                for (int p = 0; p < STAGE; ++p) {
                    const uint32_t off = stage_off(p, full_bytes);
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
        if (!PROBE_EARLY && !SORT && more) nxt = probe(tn);
        if (WSTAGE) {
            w_nxt = more && wstage_ok(nxt);
            if (w_nxt) wstage_load(nxt);
        }
        if (SEG) ahead = fetch_desc(min(next_tile(tn), n_tiles - 1));
        if (SEG && more) probe_index(nxt);
        if (SEG && more && width_of(nxt.len) != wu) band_next = band_fetch(width_of(nxt.len));
        uint32_t oidx = 0;
        if (scatter) oidx = index_issue(r);
        if (RAG && !cur.take) { // nothing was loaded: this tile is sk_scan_team_kernel's
            // tell it that there is work: the word after the error word takes this scan's number (scans of a
            // stream are ordered and numbered upwards, so the word never needs a reset)
            // ... and counts the tiles left, under the scan's number (word 6: number << 32 | count), so that the general
            // kernel can tell a batch that was left to it whole
            if (lane == 0) {
                atomicMax(errword + 1, (unsigned long long)a.scan_id);
                atomicMax(errword + 6, (unsigned long long)(a.scan_id & 0xffffffffull) << 32);
                atomicAdd(errword + 6, 1ull);
            }
            if (more && nxt.take) load_tile(qual, buf0, nxt);
            cur = nxt;
            continue;
        }
        const uint32_t *row = reinterpret_cast<const uint32_t *>(tile + (size_t)rlane * ts);
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
        // (SORT: the longest and the shortest read of the tile and their one window width come with the descriptor)
        const int Lmax = UNIFORM ? L : SORT ? sort_lmax : tile_u ? __builtin_amdgcn_readfirstlane(L) : wave_max(L);
        const int wmax = UNIFORM ? w : SORT ? sort_w : tile_u ? __builtin_amdgcn_readfirstlane(w) : wave_max(w);
        const int nwinmax = UNIFORM ? nwin : SORT ? (sort_w > 0 ? sort_lmax - sort_w + 1 : 0) : tile_u ? __builtin_amdgcn_readfirstlane(nwin) : wave_max(nwin);
        // the dwords every read of the tile has in full (SORT: lengths differ by < 10 inside a tile)
        const int Lfull = SORT ? sort_lmin : Lmax;

        // ---- range check of the whole read in 2 ops per dword: for a char c in [min,max],
        // |c-min| + |c-max| == max-min, and it is larger for every other byte value, so the
        // read is clean iff the two SADs add up to L*(max-min).  (Whether a bad char counts is
        // decided below against the part of the read the reference would have touched.)
        uint32_t sad = 0;
        {
            const uint64_t *row64 = reinterpret_cast<const uint64_t *>(row);
            auto sad8 = [&](int k, uint32_t &acc) { // dwords k .. k+7: 4 x ds_read_b64 (re-strided images: 2 x ds_read_b128) in flight, then 16 SADs
                uint64_t x[4];
                if (RAG) {
                    const sk_v4u *r128 = reinterpret_cast<const sk_v4u *>(__builtin_assume_aligned(row + k, 16));
                    const sk_v4u lo = r128[0], hi = r128[1];
                    x[0] = ((uint64_t)lo[1] << 32) | lo[0];
                    x[1] = ((uint64_t)lo[3] << 32) | lo[2];
                    x[2] = ((uint64_t)hi[1] << 32) | hi[0];
                    x[3] = ((uint64_t)hi[3] << 32) | hi[2];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = row64[(k >> 1) + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = __builtin_amdgcn_sad_u8((uint32_t)x[u], min4, acc);
                    acc = __builtin_amdgcn_sad_u8((uint32_t)x[u], max4, acc);
                    acc = __builtin_amdgcn_sad_u8((uint32_t)(x[u] >> 32), min4, acc);
                    acc = __builtin_amdgcn_sad_u8((uint32_t)(x[u] >> 32), max4, acc);
                }
            };
            int k = 0;
            if (UNIFORM || tile_u) {
                const int full = (SORT ? min(Lfull, Lmax) : Lmax) >> 2; // whole dwords; rows are 8-byte aligned
                if (WIDE) {
                    // the two lanes of a read take its 8-dword groups in turns and add up what they found
                    int groups = 0;
                    for (int g = WIDE == 2 ? lane >> 4 : half; 8 * g + 8 <= full; g += WIDE == 2 ? 4 : 2, ++groups) sad8(8 * g, sad);
                    sad -= (uint32_t)(32 * groups * range);
                    const sk_v2u both = __builtin_amdgcn_permlane32_swap(sad, sad, false, false);
                    sad = both[0] + both[1];
                    if (WIDE == 2) sad += (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)sad);
                    k = full & ~7; // (the rest, below, by both alike)
                } else {
                    for (; k + 8 <= full; k += 8) sad8(k, sad);
                }
                for (; k + 2 <= full; k += 2) {
                    const uint64_t x = row64[k >> 1];
                    sad = __builtin_amdgcn_sad_u8((uint32_t)x, min4, sad);
                    sad = __builtin_amdgcn_sad_u8((uint32_t)x, max4, sad);
                    sad = __builtin_amdgcn_sad_u8((uint32_t)(x >> 32), min4, sad);
                    sad = __builtin_amdgcn_sad_u8((uint32_t)(x >> 32), max4, sad);
                }
            }
            const int k_rest = WIDE ? (((SORT ? min(Lfull, Lmax) : Lmax) >> 2) & ~7) : 0; // dwords already accounted for
            for (; 4 * k < Lmax; ++k) { // per-lane masking: the last dword(s) (UNIFORM) / mixed lengths
                const uint32_t x = first_bytes(row[k], L - 4 * k, min4);
                sad = __builtin_amdgcn_sad_u8(x, min4, sad);
                sad = __builtin_amdgcn_sad_u8(x, max4, sad);
            }
            // every dword visited contributes 4*range when clean (fillers are legal chars)
            sad -= (uint32_t)(4 * (k - k_rest) * range);
        }
        const bool bad = scanned && sad != 0;

        // ---- all windows, 32 per trip: trim.cpp:34-81 without the breaks.  Branch-free state
        // step, so that the compiler can overlap it with the next trip's loads and MFMAs.
        // i0u / i1u: the 5' and the 3' window, NONE until found (a min() over the trips keeps
        // the first one, because later trips can only offer larger indices)
        uint32_t i0u = NONE, i1u = NONE;
        uint32_t ltu = NONE; // WIDE == 2: the first window below the threshold in this lane's half of the windows, 5' window or not
        bool all_have5 = false; // wave-uniform
        // WIDE == 2: this lane's windows are the nwin_l from window wofs on (the loop's `base` counts from there)
        const int whalf = WIDE == 2 ? ((nwin + 63) >> 6) << 5 : 0;
        const int wofs = WIDE == 2 ? part * whalf : 0;
        const int nwin_l = WIDE == 2 ? (part ? max(nwin - whalf, 0) : min(nwin, whalf)) : nwin;
        // bit (31 - s) of M: window base+s is below the threshold
        auto step32 = [&](uint32_t M, int base) {
            const int nv = nwin_l - base;
            const uint32_t vmask = nv >= 32 ? ~0u : (nv <= 0 ? 0u : ~(~0u >> nv));
            const uint32_t lt = M & vmask;
            const uint32_t at = (uint32_t)(base + wofs);
            uint32_t cand = lt; // with -x the 3' search starts at window 0 (trim.cpp:62)
            if (!a.no5 && !all_have5) {
                const uint32_t ge = ~M & vmask;
                i0u = min(i0u, __builtin_elementwise_add_sat(ffbh_or_none(ge), at)); // trim.cpp:42
                // windows of this trip strictly after i0: the low (base+31 - i0) bits, all 32 if
                // i0 lies in an earlier trip, none while it is not found
                const uint32_t width = __builtin_elementwise_sub_sat(at + 31u, i0u);
                const uint32_t low = (1u << (width & 31u)) - 1u;
                cand = lt & (width >= 32u ? ~0u : low);
                // once every lane has its 5' window, later trips need neither the search nor the mask
                all_have5 = __builtin_amdgcn_ballot_w64(i0u == NONE) == 0;
            }
            if (WIDE == 2 && !a.no5) ltu = min(ltu, __builtin_elementwise_add_sat(ffbh_or_none(lt), at));
            i1u = min(i1u, __builtin_elementwise_add_sat(ffbh_or_none(cand), at)); // trim.cpp:61
        };
        // WIDE == 2: the two halves of a read's windows joined (lanes n and n ^ 16 hold them)
        auto join_halves = [&]() {
            const int other = (lane ^ 16) << 2;
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_ds_bpermute(other, (int)i0u), o1 = (uint32_t)__builtin_amdgcn_ds_bpermute(other, (int)i1u),
                           ol = (uint32_t)__builtin_amdgcn_ds_bpermute(other, (int)ltu);
            const uint32_t a0 = part ? o0 : i0u, a1 = part ? o1 : i1u;                          // the first half's
            const uint32_t b0 = part ? i0u : o0, b1 = part ? i1u : o1, bl = part ? ltu : ol;   // the second half's
            if (a.no5) {
                i1u = a1 != NONE ? a1 : b1;
            } else {
                i0u = a0 != NONE ? a0 : b0;
                i1u = a0 != NONE ? (a1 != NONE ? a1 : bl) : b1;
            }
        };

        if (MFMA && (UNIFORM || tile_u)) {
            // lane (n = lane&31, half): 16 bytes of read 32g+n at positions 32*kb + 16*half
            const uint8_t *frag0 = tile + (size_t)(WIDE == 2 ? rlane : lane & 31) * ts + 16 * half + wofs;
            const uint8_t *frag1 = frag0 + (size_t)32 * ts;
            auto load_frag = [](const uint8_t *p) -> sk_v4i {
                // re-strided images have rows at a pitch of 16 x odd bytes: ONE 16-byte read, 16 lanes to the 64 banks
                // (two 8-byte reads fall two lanes to a bank pair there); the strided layouts' rows are 8-byte aligned
                if (RAG) return *reinterpret_cast<const sk_v4i *>(__builtin_assume_aligned(p, 16));
                const uint64_t lo = *reinterpret_cast<const uint64_t *>(p);
                const uint64_t hi = *reinterpret_cast<const uint64_t *>(p + 8);
                sk_v4i f = {(int)(uint32_t)lo, (int)(uint32_t)(lo >> 32), (int)(uint32_t)hi, (int)(uint32_t)(hi >> 32)};
                return f;
            };
            // 16 sign bits per accumulator block, window order; lanes 0..31 keep reads 0..31 (group
            // 0), lanes 32..63 reads 32..63 (group 1): after the swap s[0] = windows 0..15 of the
            // lane's own read, s[1] = windows 16..31
            auto collect = [&](const sk_v16i &d0, const sk_v16i &d1, int base) {
                // (the quiet-step skip of the medium-read tiles, tried here for segmented and regrouped batches: with 64 reads to
                // a tile a trip without a single sum below the threshold is rare, the test costs more than it saves)
                uint32_t p0 = 0, p1 = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) p0 = __builtin_amdgcn_alignbit(p0, (uint32_t)d0[i], 31);
#pragma unroll
                for (int i = 0; i < 16; ++i) p1 = __builtin_amdgcn_alignbit(p1, (uint32_t)d1[i], 31);
                const sk_v2u s = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
                step32((s[0] << 16) | s[1], base);
            };
            if (WIDE) {
                // one group of 32 reads; lane (n, half): 16 bytes of read n at positions 32 * block + 16 * half
                const int dm = wu / 32 - 1;      // blocks every window of a step covers whole (wu >= 32)
                const sk_v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101}, minus = {-1, -1, -1, -1};
                sk_v16i mid = negT; // -T + the whole blocks of the step
                auto frag = [&](int blk) -> sk_v4i { return load_frag(frag0 + 32 * blk); };
                for (int d = 1; d <= dm; ++d) mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, frag(d), mid, 0, 0, 0);
                // the four fragments a step works on: blocks st, st + 1 and the two with the band's far edge
                struct frags {
                    sk_v4i q0, q1, qf, qg;
                };
                auto fetch = [&](int st) -> frags { return frags{frag(st), frag(st + 1), frag(st + dm + 1), frag(st + dm + 2)}; };
                // the sums of a step, and `mid` moved on to the step after it: block st + dm + 1 in, block st + 1 out
                auto sums = [&](const frags &f) -> sk_v16i {
                    sk_v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA0, f.q0, mid, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA1, f.qf, d0, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bandA2, f.qg, d0, 0, 0, 0);
                    if (dm > 0) {
                        mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, f.qf, mid, 0, 0, 0);
                        mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(minus, f.q1, mid, 0, 0, 0);
                    }
                    return d0;
                };
                auto signs = [&](const sk_v16i &d0, int base) {
                    if (SK_WIDE_QUIET && (a.no5 || all_have5)) {
                        // every lane has its 5' window: a step without a single sum below the threshold -- the usual
                        // step between the head of a read and its 3' decline -- changes nothing for anyone
                        uint32_t any = 0;
#pragma unroll
                        for (int i = 0; i < 16; i += 2) any |= (uint32_t)d0[i] | (uint32_t)d0[i + 1];
                        if (__builtin_amdgcn_ballot_w64((int)any < 0) == 0) return;
                    }
                    uint32_t p0 = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) p0 = __builtin_amdgcn_alignbit(p0, (uint32_t)d0[i], 31);
                    // lanes n and n + 32 both get windows 0..15 (from lane n) and 16..31 (from lane n + 32)
                    const sk_v2u sw = __builtin_amdgcn_permlane32_swap(p0, p0, false, false);
                    step32((sw[0] << 16) | sw[1], base);
                };
                // Two steps per turn; the fragments are read out of LDS two steps ahead and the matrix pipe runs one step
                // ahead of the vector ALU: a step's five MFMAs are two dependent chains of 64 cycles a link, the sign
                // collection of the step before runs underneath them.
                // (Tried: leaving the walk once every lane has its 3' window.  Every variant of it -- break, a shrinking bound,
                // with and without the next step's MFMAs in flight -- gave wrong cuts in tests/soak_wide.py.  A build that
                // only PRINTS the state after every turn shows it right (no lane with a 3' window, cuts equal to the
                // oracle's); the same build with the exit taken shows window 14 "below the threshold" for every read after
                // the FIRST turn: the exit changes the code generated for the steps before it.  Compiler or hazard, not
                // the algorithm -- not pursued, not done.)
                const int nsteps = WIDE == 2 ? min(nwinmax, whalf) : nwinmax; // windows a lane walks
                if (WSTAGE) {
                    // (the register stage leaves no room for the second set of fragments: read one step ahead only)
                    frags fa = fetch(0);
                    sk_v16i da = sums(fa), db = da;
                    for (int base = 0; base < nsteps; base += 64) {
                        const int st = base >> 5;
                        const bool second = base + 32 < nsteps, third = base + 64 < nsteps;
                        if (second) {
                            fa = fetch(st + 1);
                            db = sums(fa);
                        }
                        signs(da, base);
                        if (!second) break;
                        if (third) {
                            fa = fetch(st + 2);
                            da = sums(fa);
                        }
                        signs(db, base + 32);
                    }
                } else {
                frags fa = fetch(0), fb = fa;
                if (32 < nsteps) fb = fetch(1);
                sk_v16i da = sums(fa), db = da;
                for (int base = 0; base < nsteps; base += 64) {
                    const int st = base >> 5;
                    const bool second = base + 32 < nsteps, third = base + 64 < nsteps;
                    if (third) fa = fetch(st + 2);
                    if (second) db = sums(fb);
                    signs(da, base);
                    if (!second) break;
                    if (base + 96 < nsteps) fb = fetch(st + 3);
                    if (third) da = sums(fa);
                    signs(db, base + 32);
                }
                }
                if (WIDE == 2) join_halves();
            } else if (SEG == 2 || (SEG != 3 && !three_blocks)) { // w <= 33: positions base .. base+63, two trips per turn so that
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
            // (WIDE: the two / four lanes of a read take the dwords behind the first in turns and keep the earliest hit)
            constexpr int SHARE = WIDE == 2 ? 4 : WIDE ? 2 : 1;
            for (int it = 1 + (WIDE == 2 ? lane >> 4 : WIDE ? half : 0); it < trips; it += SHARE) {
                const uint32_t g5 = ge_flags(r5[it], cthr4);
                const uint32_t g3 = ge_flags(r3[it], cthr4) ^ H4;
                const uint32_t rel = 32u * (uint32_t)it;
                h5 = min(h5, __builtin_elementwise_add_sat(ffbl_or_none(g5), rel));
                h3 = min(h3, __builtin_elementwise_add_sat(ffbl_or_none(g3), rel));
            }
            if (WIDE) {
                const sk_v2u s5 = __builtin_amdgcn_permlane32_swap(h5, h5, false, false), s3 = __builtin_amdgcn_permlane32_swap(h3, h3, false, false);
                h5 = min(s5[0], s5[1]);
                h3 = min(s3[0], s3[1]);
                if (WIDE == 2) {
                    h5 = min(h5, (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)h5));
                    h3 = min(h3, (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)h3));
                }
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
                // (segmented batches: r is the slot; the caller's read number is out_index[slot])
                if (p < touched) report_error(errword, SEG ? (uint64_t)out_index[r] : r, p, (int)(int8_t)(tile[(size_t)rlane * ts + p]));
            }
        }

        // ---- the N rule: trim.cpp:86-98 (lowercase n: cut before it; only uppercase N: cut = -2)
        if (HAS_SEQ) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (SEQ_SHARES) {
                // the quality scan is over: the same buffer now takes the sequence tile of these reads
                load_tile(seq, buf0, cur);
                wait_vmcnt(0);
                if (scatter) index_settle(oidx);
            } else {
                // buf0 is free now: start Q(t+1), then retire S(t)
                if (more) tile_to_lds(qual + nxt.off, buf0, next_bytes, lane);
                wait_vmcnt(next_pieces); // older than Q(t+1): S(t)
            }
            const uint32_t *srow = reinterpret_cast<const uint32_t *>((SEQ_SHARES ? buf0 : buf1) + (size_t)rlane * ts);
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
                    const int full = (SORT ? min(Lfull, Lmax) : Lmax) >> 2;
                    const uint64_t *srow64 = reinterpret_cast<const uint64_t *>(srow);
                    // (WIDE: the two / four lanes of a read take the 8-dword groups in turns; the rest below by all alike)
                    constexpr int SHARE = WIDE == 2 ? 4 : WIDE ? 2 : 1;
                    for (k = 8 * (WIDE == 2 ? lane >> 4 : WIDE ? half : 0); k + 8 <= full; k += 8 * SHARE) {
                        uint64_t x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[u] = srow64[(k >> 1) + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            n_step((uint32_t)x[u], 32u * (uint32_t)(k + 2 * u));
                            n_step((uint32_t)(x[u] >> 32), 32u * (uint32_t)(k + 2 * u + 1));
                        }
                    }
                    if (WIDE) k = full & ~7;
                }
                for (; 4 * k < Lmax; ++k) n_step(first_bytes(srow[k], L - 4 * k, 0u), 32u * (uint32_t)k);
                if (WIDE) {
                    const sk_v2u sn = __builtin_amdgcn_permlane32_swap(nlo, nlo, false, false), sa = __builtin_amdgcn_permlane32_swap(anyN, anyN, false, false);
                    nlo = min(sn[0], sn[1]);
                    anyN = sa[0] | sa[1];
                    if (WIDE == 2) {
                        nlo = min(nlo, (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)nlo));
                        anyN |= (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)anyN);
                    }
                }
            }
            if (nlo != NONE) three = (int)(nlo >> 3) - 1;
            else if (anyN) three = -2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (SORT && more) {
                nxt = sort_finish(raw);
                if (next_w != wu && next_w != 0) band_next = band_fetch(next_w);
            }
            if (more) {
                if (SEQ_SHARES) { if (nxt.take) load_tile(qual, buf0, nxt); }   // Q(t+1)
                else tile_to_lds(seq + nxt.off, buf1, next_bytes, lane);      // S(t+1)
            }
        } else if (NBUF == 1 && ABLATE != 2 && (!STAGE || (more && !cur_staged))) {
            // single buffer: every LDS read of this tile is done, refill it now -- the cut store
            // below and the other waves of the CU cover the DMA latency.  (Staged kernel: only the
            // ragged last tile of the batch takes this way.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (scatter) index_settle(oidx);
            if (SORT && more) {
                nxt = sort_finish(raw);
                if (next_w != wu && next_w != 0) band_next = band_fetch(next_w);
            }
            if (more && nxt.take && !(WSTAGE && w_nxt)) load_tile(qual, buf0, nxt);
        }
        if (WSTAGE) w_cur = w_nxt;
#if SK_TAIL_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif

        // ---- trim.cpp:103-108
        if (!scanned || !found5 || (three - five < a.lthr)) {
            five = -1;
            three = -1;
        }
        // (staged kernels: no refill above, the next tile's pieces went out before the scan.  Tried and dropped: the cuts
        // stored a turn late, so that the next turn's wait for its pieces does not sit out this store -- 1-7 % slower)
        if (STAGE && scatter) index_settle(oidx);
        if (active && (!WIDE || lane < (int)ROWS)) out[scatter ? (uint64_t)oidx : r] = sk_cut_dev{five, three};
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
    // a ragged batch that went through the regrouping (sk_sort.hip) and turned out mixed, without reads too long for
    // the tiles, is the sorted scan's (enqueued right behind; it asks the opposite question)
    if (!UNIFORM && a.sort_flags && scalar_load(a.sort_flags) != 0 && scalar_load(a.sort_flags + 1) == 0) return;
    sk_scan_tile_body<UNIFORM, HAS_SEQ, MFMA, 1, 0, false, 0, true>(qual, seq, lengths, out, errword, a, nullptr, nullptr, offsets);
}

// the tiles of a regrouped ragged batch (sk_sort.hip): lists = 8 tile lists of a.n_tiles entries, counts = their lengths
template <bool HAS_SEQ>
__global__ void __launch_bounds__(SK_TILE_THREADS, 2)
sk_scan_tile_sorted_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                           const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ perm,
                           const unsigned long long *__restrict__ lists, const uint32_t *__restrict__ counts,
                           sk_cut_dev *__restrict__ out, unsigned long long *errword, sk_scan_args a)
{
    if (scalar_load(a.sort_flags) == 0 || scalar_load(a.sort_flags + 1) != 0) return; // a uniform batch, or one with long reads
    sk_scan_tile_body<false, HAS_SEQ, true, 1, 0, false, 0, true, true>(qual, seq, counts, out, errword, a, reinterpret_cast<const sk_tile_dev *>(lists),
                                                                         reinterpret_cast<const uint32_t *>(perm), offsets);
}

// uniform medium reads (WIDE): tiles of 32 reads, a pair of lanes per read, windows of any width from 32 up
template <bool HAS_SEQ, int WIDE, int WSTAGE = 0>
__global__ void __launch_bounds__(SK_TILE_THREADS, 2)
sk_scan_tile_wide_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq, sk_cut_dev *__restrict__ out,
                         unsigned long long *errword, sk_scan_args a)
{
    sk_scan_tile_body<true, HAS_SEQ, true, 1, 0, false, 0, true, false, WIDE, WSTAGE>(qual, seq, nullptr, out, errword, a, nullptr, nullptr, nullptr);
}

// the register-staged variant (STAGE = KiB pieces per tile = ceil(stride / 16)): the registers of a
// wave hold the scan state and the next tile, three waves per SIMD (12 per CU) at STAGE = 10
template <int STAGE, int ABLATE = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STAGE > 10 ? 2 : 3, STAGE > 10 ? 3 : 4)))
sk_scan_tile_staged_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                           const uint32_t *__restrict__ lengths, sk_cut_dev *__restrict__ out,
                           unsigned long long *errword, sk_scan_args a, const sk_tile_dev *__restrict__ tiles = nullptr,
                           const uint32_t *__restrict__ out_index = nullptr)
{
    sk_scan_tile_body<true, false, true, 1, ABLATE, false, STAGE, false>(qual, seq, lengths, out, errword, a, tiles, out_index, nullptr);
}

// segmented batches without -n whose rows fit 20 KiB tiles (strides up to 320): the register-staged variant, tile
// by tile as many pieces as the tile has (STAGE = the most: 10 or 20)
template <int STAGE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STAGE > 10 ? 2 : 3, STAGE > 10 ? 2 : 3)))
sk_scan_seg_staged_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                          const uint32_t *__restrict__ lengths, sk_cut_dev *__restrict__ out,
                          unsigned long long *errword, sk_scan_args a, const sk_tile_dev *__restrict__ tiles,
                          const uint32_t *__restrict__ out_index)
{
    sk_scan_tile_body<true, false, true, 1, 0, 2, STAGE, false>(qual, seq, lengths, out, errword, a, tiles, out_index, nullptr);
}

// ------------------------------------------------------------------------------------------
// launchers (host side of this translation unit)
// ------------------------------------------------------------------------------------------
namespace {

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
    {
        // the staged kernels are bounded by their registers, not by LDS (the grid is persistent, so
        // workgroups beyond what fits would only run as a second round): waves per SIMD = the 512
        // registers of a lane's file over the kernel's count (allocated in eights), four SIMDs
        const int fits = reg_fit(facts);
        if (by_registers && getenv("SK_DEBUG_LAUNCH")) fprintf(stderr, "[sk] staged kernel: %d registers -> %d workgroups per CU\n", facts.regs, fits);
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
                SK_STAGED(11); SK_STAGED(12); SK_STAGED(13); SK_STAGED(14); SK_STAGED(15); SK_STAGED(16);
                SK_STAGED(17); SK_STAGED(18); SK_STAGED(19); SK_STAGED(20);
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
        const uint64_t chunk = 1ull << a->seg_chunk_shift; // a wave takes this many consecutive tiles at a time (SEG_CHUNK)
        const uint64_t chunks = ((uint64_t)k.n_tiles + chunk - 1) / chunk;
        if (grid > chunks) grid = chunks;
        sk_scan_args as = *a;
        as.buf_bytes = lds_bytes;
        as.n_tiles = k.n_tiles;
        auto launch = [&](auto kern, int wg_per_cu = 0) {
            const kernel_facts facts = prepare_kernel(kern);
            if (facts.status != hipSuccess) return facts.status;
            uint64_t g = grid;
            const int fit = wg_per_cu > 0 ? std::min(wg_per_cu, reg_fit(facts)) : reg_fit(facts);
            if (fit < per_cu) { // bounded by registers, not by LDS
                g = (uint64_t)cu_count * fit;
                if (g > chunks) g = chunks;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64), lds_bytes, stream, qual, seq,
                               (const uint32_t *)nullptr, out, errword, as, tiles + k.first_tile, out_index);
            return hipGetLastError();
        };
        hipError_t e;
        // no -n, windows <= 33, rows up to 320 bytes: the next tile can wait in the wave's registers (SK_SEG_STAGE=1).  Not the
        // default: on the mixed batch with the realistic quality model it is 3 % faster than the LDS-DMA kernel (162 against
        // 167.5 us), on the bench's `seg` batch 10 % slower (0.180 against 0.162 ms) -- see DESIGN 4.1.1
        static const bool seg_stage = [] { const char *v = getenv("SK_SEG_STAGE"); return v && *v == '1'; }();
        if (seg_stage && !a->truncn && !k.wide && k.max_stride <= 320u)
            e = k.max_stride <= 160u ? launch(sk_scan_seg_staged_kernel<10>, 12) : launch(sk_scan_seg_staged_kernel<20>, 8);
        else if (a->truncn) e = k.wide ? launch(sk_scan_tile_kernel<true, true, true, 1, 0, 3>) : launch(sk_scan_tile_kernel<true, true, true, 1, 0, 2>);
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
        uint64_t g = (uint64_t)cu_count * std::min(per_cu, reg_fit(facts)); // (LDS and registers)
        if (g > n_tiles) g = n_tiles;
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64), lds_bytes, stream, qual, seq, offsets, lengths, out,
                           errword, *a);
        return hipGetLastError();
    };
    const uint32_t wu = a->read_len / 10 ? a->read_len / 10 : a->read_len;
    const bool mfma = uniform && wu <= 65 && a->read_len > 0;
    // (Tried for packed short reads, reads up to 240 bases back to back: the register stage of the medium-read tiles --
    // next tile in registers, source offsets computed once.  0.56-0.60 against 0.55-0.62 of peak without it on
    // packed150: no gain where 14 waves' images fit a CU anyway; not kept.)
    if (a->truncn) {
        if (mfma) return launch(sk_scan_tile_any_kernel<true, true, true>);
        if (uniform) return launch(sk_scan_tile_any_kernel<true, true, false>);
        return launch(sk_scan_tile_any_kernel<false, true, true>);
    }
    if (mfma) return launch(sk_scan_tile_any_kernel<true, false, true>);
    if (uniform) return launch(sk_scan_tile_any_kernel<true, false, false>);
    return launch(sk_scan_tile_any_kernel<false, false, true>);
}

// Uniform medium reads: the LDS bytes a wave needs for its image of reads of `read_len` bytes (0: not for this kernel --
// windows narrower than 32, or fewer than two waves to a CU)
static uint32_t wide_image_bytes(uint32_t rows, uint32_t read_len)
{
    const uint32_t cpr = ((read_len + 15u) >> 4) | 1u;           // 16-byte chunks per image row (rag_pitch)
    return ((rows * cpr + 63u) >> 6) * 1024u + SK_TILE_SLACK; // whole pieces of the loader
}

#define SK_WIDE_STAGE 20 /* pieces the register stage of the medium-read tiles holds */
// 32 reads to a tile while five waves' images still fit a CU (reads up to ~940 bases), 16 beyond (measured: 600 bases 3.5
// against 2.9 TB/s, 1 000 bases 2.2 against 2.7); SK_WIDE_ROWS=16|32 forces one (A/B runs)
static uint32_t wide_rows(uint32_t read_len)
{
    static const uint32_t forced = [] { const char *e = getenv("SK_WIDE_ROWS"); return e ? (uint32_t)atoi(e) : 0u; }();
    if (forced == 16u || forced == 32u) return forced;
    return wide_image_bytes(32u, read_len) * 5u <= SK_LDS_PER_CU ? 32u : 16u;
}
// does the next tile wait in registers?  (no -n, an image of at most SK_WIDE_STAGE pieces;
// SK_WIDE_STAGE=0 in the environment: never -- A/B runs)
static bool wide_staged(const uint8_t *qual, const sk_scan_args *a, uint32_t rows)
{
    static const bool on = [] { const char *e = getenv("SK_WIDE_STAGE"); return !(e && *e == '0'); }();
    const uint32_t pieces = (wide_image_bytes(rows, a->read_len) - SK_TILE_SLACK) >> 10;
    (void)qual; // (rows at any byte address: the device's unaligned access mode takes 16-byte loads anywhere, as the LDS-DMA loader's do)
    return on && !a->truncn && pieces <= SK_WIDE_STAGE;
}

extern "C" __attribute__((visibility("hidden"))) uint32_t sk_wide_lds_bytes(uint32_t read_len)
{
    if (read_len / 10u < 32u || read_len > 8192u) return 0;
    const uint32_t bytes = wide_image_bytes(wide_rows(read_len), read_len);
    return bytes <= SK_LDS_PER_CU / 2u ? bytes : 0u;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_wide(const uint8_t *qual, const uint8_t *seq, sk_cut_dev *out, unsigned long long *errword,
                                     const sk_scan_args *a, int cu_count, hipStream_t stream)
{
    sk_scan_args aw = *a;
    if (sk_wide_lds_bytes(a->read_len) == 0) return hipErrorInvalidValue;
    // 16-read tiles with the next tile in registers where that is to be had (reads up to ~1 000 bases without -n),
    // else 32- or 16-read tiles by LDS-DMA
    static const uint32_t forced_rows = [] { const char *e = getenv("SK_WIDE_ROWS"); return e ? (uint32_t)atoi(e) : 0u; }();
    // (where seven waves' 32-read images fit a CU -- reads up to ~700 bases -- those win: 600 bases 3.4 against 3.2 TB/s;
    // beyond, the staged 16-read tiles: 800 bases 3.26 against 2.90, 1 000 bases 3.56 against 3.30)
    // (SK_WIDE_STAGE32=0: never the 32-read stage -- A/B runs)
    static const bool stage32_on = [] { const char *e = getenv("SK_WIDE_STAGE32"); return !(e && *e == '0'); }();
    const bool staged32 = stage32_on && forced_rows != 16u && wide_staged(qual, a, 32u); // reads up to ~624 bases: 32-read tiles of up to 20 pieces
    const bool staged = !staged32 && forced_rows != 32u && wide_staged(qual, a, 16u) && (forced_rows == 16u || wide_image_bytes(32u, a->read_len) * 7u > SK_LDS_PER_CU);
    const uint32_t rows = staged32 ? 32u : staged ? 16u : wide_rows(a->read_len);
    aw.buf_bytes = wide_image_bytes(rows, a->read_len);
    int per_cu = (int)(SK_LDS_PER_CU / aw.buf_bytes);
    if (per_cu > 16) per_cu = 16;
    const uint64_t n_tiles = (a->n_reads + rows - 1) / rows;
    uint64_t grid = (uint64_t)cu_count * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    if (grid == 0) return hipSuccess;
    auto launch = [&](auto kern) {
        const kernel_facts facts = prepare_kernel(kern);
        if (facts.status != hipSuccess) return facts.status;
        // the grid is persistent: workgroups beyond what a CU's registers hold would run as a second round, a wave to a CU
        uint64_t g = (uint64_t)cu_count * std::min(per_cu, reg_fit(facts));
        if (g > n_tiles) g = n_tiles;
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64), aw.buf_bytes, stream, qual, seq, out, errword, aw);
        return hipGetLastError();
    };
    if (staged32) return launch(sk_scan_tile_wide_kernel<false, 1, SK_WIDE_STAGE>);
    if (staged) return launch(sk_scan_tile_wide_kernel<false, 2, SK_WIDE_STAGE>);
    if (rows == 32) return a->truncn ? launch(sk_scan_tile_wide_kernel<true, 1>) : launch(sk_scan_tile_wide_kernel<false, 1>);
    return a->truncn ? launch(sk_scan_tile_wide_kernel<true, 2>) : launch(sk_scan_tile_wide_kernel<false, 2>);
}

// The sorted scan of a regrouped ragged batch.  a->buf_bytes = LDS bytes of a wave (sized for the longest read the
// tiles take), a->n_tiles = entries per tile list, a->sort_flags = the sort's verdict (device).
extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_sorted(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets, const uint64_t *perm,
                                       const unsigned long long *lists, const uint32_t *counts, sk_cut_dev *out,
                                       unsigned long long *errword, const sk_scan_args *a, int cu_count, hipStream_t stream)
{
    const uint32_t lds_bytes = a->buf_bytes;
    if (lds_bytes == 0 || lds_bytes > SK_LDS_PER_CU) return hipErrorInvalidValue;
    int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
    if (per_cu > 16) per_cu = 16;
    uint64_t grid = ((uint64_t)cu_count * per_cu) & ~7ull; // the same number of workgroups for every list
    const uint64_t most = ((a->n_reads + 63) >> 6) + 8;
    if (grid > most) grid = (most + 7) & ~7ull;
    if (grid < 8) grid = 8;
    auto launch = [&](auto kern) {
        const kernel_facts facts = prepare_kernel(kern);
        if (facts.status != hipSuccess) return facts.status;
        uint64_t g = ((uint64_t)cu_count * std::min(per_cu, reg_fit(facts))) & ~7ull; // (LDS and registers; the same number of workgroups for every list)
        if (g > grid) g = grid;
        if (g < 8) g = 8;
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(64), lds_bytes, stream, qual, seq, offsets, perm, lists, counts, out, errword, *a);
        return hipGetLastError();
    };
    return a->truncn ? launch(sk_scan_tile_sorted_kernel<true>) : launch(sk_scan_tile_sorted_kernel<false>);
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
