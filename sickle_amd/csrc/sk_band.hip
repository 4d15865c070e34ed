// sk_band.hip -- the general kernel for MEDIUM reads (beyond a 64-read LDS tile, up to a few kilobases): the reads
// resident in LDS, the window sums of a whole read taken from the integer matrix pipe, two reads per wave at a time;
// see the block comment below.
#include "sk_kernel_common.h"

// ------------------------------------------------------------------------------------------
// sk_scan_band_kernel: reference src/trim.cpp:3-116, a wave per read or per pair of reads.
//
// The lane-per-read tile kernels stop at rows of 512 bytes (64 rows must fit a wave's LDS buffer), the streaming
// kernel (sk_stream.hip) only pays for itself from ~4 kb on (its cost per READ is several hundred scalar
// instructions).  Between them round 2 ran teams of 16 lanes per read (sk_team.hip) on the vector ALU: 12 vector
// instructions per 4 windows per lane, 0.18-0.31 of the HBM peak.  Here the windows of ONE read are the columns of
// a matrix product:
//     S[32 n + m] - T = sum over blocks blk of  band_blk[m][k] * c[32 (n + blk) + k]  - T      (m, k in 0..31)
// i.e. D = sum_blk A_blk x B_blk + (-T) with B_blk = the read's bytes from 32 blk on, taken as 32 columns of 32
// consecutive bytes -- a lane's B operand is 16 CONSECUTIVE bytes of the read (one ds_read_b128, the wave's 64
// lanes together read one contiguous KiB), and one CHAIN of v_mfma_i32_32x32x32_i8 gives 1024 windows.  The band
// of a window of width w covers (w + 30) / 32 + 1 blocks: the first and the last two are partial (per-lane
// constants, rebuilt when w changes, which in a batch of equal lengths is never), the ones in between are all
// ones.  Exact in int32 (a window sums at most a few hundred bytes).  Rows of A are permuted as in the tile kernel,
// so that a lane's 16 accumulators are 16 consecutive windows and one v_alignbit per window collects the signs.
//
// A turn runs TWO chains and one v_permlane32_swap: lanes 0..31 then hold the 32-window masks of the first chain's
// columns, lanes 32..63 those of the second's -- window order = lane order, so "the first window at/above the
// threshold" and "the first one below it after that" (trim.cpp:42, :61) are a ballot, a find-first and a v_readlane
// per chain.  The two chains are
//   * two READS, when both have at most 1024 windows (reads up to ~1.1 kb): everything after the chains -- the two
//     in-window searches (trim.cpp:46-51, :65-70), the range check, the N rule, the store -- is done for both reads
//     at once, a read per half of the wave; each chain has its own band, so the two may differ in length;
//   * windows [2048 p, +1024) and the next 1024 of ONE longer read, p = 0, 1, ... until both windows are found.
// Cost per KiB of read: (w + 30) / 32 + 1 MFMAs (5 at 1 kb, 15 at 4 kb; the matrix pipe is otherwise idle) and a
// few dozen vector instructions, whatever the data -- averages hovering at the threshold cost nothing extra.
//
// A wave keeps a RING of reads in LDS (a.stream_nb slots; LDS-DMA, 16 bytes per lane, source address per lane: the
// image of a read starts aligned wherever it lies in the batch): the reads after the ones being scanned are in
// flight, ~8 KiB per wave.  Loads return in order, so the wait for a read is a counted s_waitcnt vmcnt(n).
// With a.buf_bytes != 0 the kernel takes only the 64-read tiles sk_scan_tile_any_kernel left.
// A read longer than a slot (a.team_maxlen) is scanned from global memory (scan_read_global: correctness path).
// ------------------------------------------------------------------------------------------
namespace {

struct sk_band {      // the band matrix of one window width, as this lane supplies it to the chains
    sk_v4i A0, Aa, Ab; // block 0, block w >> 5, the block after it (the blocks in between are all ones)
    sk_v16i negT;      // -T in every accumulator slot
};

struct sk_band_read { // one read of the ring (everything wave-uniform)
    uint32_t r;       // its number in the batch
    uint32_t buf;     // LDS byte offset of its quality bytes (the sequence bytes follow a.team_rbuf later)
    int L, w, nwin;
    int phase;        // 0: looking for the first S >= T, 1: for the first S < T after it, 2: both found
    int i0, i1;
};

// a read that does not go through the ring: nothing to scan (trim.cpp:21), longer than a slot, or the batch's last
// read when the 16-byte chunks of its image would reach past the batch.  (Out of line it was tried: the call makes
// the kernel spill around it, 0.64 against 1.29 TB/s at 1 kb.)
template <bool HAS_SEQ>
__device__ __forceinline__ void band_other(const uint8_t *qual, const uint8_t *seq, uint64_t o, int L, uint64_t r,
                                                     sk_cut_dev *out, unsigned long long *errword, const sk_scan_args &a)
{
    const int lane = threadIdx.x;
    sk_cut_dev cut{-1, -1};
    if (L > 0 && L >= a.lthr) cut = scan_read_global<HAS_SEQ>(qual + o, HAS_SEQ ? seq + o : nullptr, L, r, lane, a, errword);
    if (lane == 0) out[r] = cut;
}

} // namespace

// LEFT: only the reads of the 64-read tiles sk_scan_tile_any_kernel left (a.buf_bytes != 0).
// UNI: a batch of equal lengths at a fixed stride (no offsets, no lengths; every read goes through the ring): the
// read's place, its window width, the band and the number of loads per read are the same for all, so nothing is
// looked up, no slot has a header, one band serves both chains -- the per-read bookkeeping of the general form is
// most of what it spends on a 1 kb read (the scalar unit, one per CU, is its bound).
template <bool HAS_SEQ, bool LEFT, bool UNI>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
sk_scan_band_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                    const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lengths,
                    sk_cut_dev *__restrict__ out, unsigned long long *errword, sk_scan_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x; // single-wave workgroups
    const int half = lane >> 5, l32 = lane & 31;
    // A slot of the ring: a read's quality bytes [, its sequence bytes], a.team_rbuf bytes each (the longest read the
    // ring takes, rounded up).  The matrix chains read up to a KiB + the band's reach past a read's last window: into
    // the next slot (whatever lies there: those windows are masked), or into the overhang behind the last slot.
    const uint32_t rb = a.team_rbuf;
    constexpr uint32_t REGIONS = HAS_SEQ ? 2u : 1u;
    const uint32_t NSLOT = a.stream_nb;
    const uint32_t slot_bytes = REGIONS * rb;
    // behind the ring: one header per slot {read number, length, offset in the batch}, written when the read is staged
    uint32_t *const headers = reinterpret_cast<uint32_t *>(lds + a.stream_tbl);
    const uint64_t batch_end = a.n_reads ? rag_batch_end(offsets, lengths, a) : 0;
    const uint32_t min4 = splat((uint32_t)a.qmin), max4 = splat((uint32_t)a.qmax);
    const uint32_t hi4 = splat((uint32_t)(127 - a.qmax));
    const uint32_t cthr4 = splat((uint32_t)a.cthr);
    const uint32_t clean16 = (uint32_t)(16 * (a.qmax - a.qmin)); // the range-check sum of 16 chars in range

    auto locate = [&](uint64_t r, uint64_t &o, int &L) { // wave-uniform
        uint64_t e;
        if (offsets) {
            scalar_load_pair(offsets + r, o, e);
        } else {
            o = r * a.stride;
            e = o + (lengths ? min(scalar_load(lengths + r), a.stride) : a.read_len);
        }
        L = e >= o ? (int)min(e - o, (uint64_t)SK_MAX_READ_LEN_DEV) : 0;
    };
    auto window_of = [](int L) { const int w = L / 10; return w ? w : L; }; // trim.cpp:8, :30
    // through the ring: a read with something to scan (trim.cpp:21) that fits a slot and whose image -- whole 16-byte
    // chunks -- stays inside the batch (the batch's last read may fail that: band_other takes it)
    auto staged = [&](int L, uint64_t o) {
        return L > 0 && L >= a.lthr && L <= (int)a.team_maxlen && o + (((uint64_t)L + 15u) & ~15ull) <= batch_end;
    };

    // ---- the read into LDS: chunk c (16 bytes) by lane c mod 64
    auto stage = [&](const uint8_t *base, uint64_t o, int L, uint8_t *dst) {
        const uint32_t nch = ((uint32_t)L + 15u) >> 4;
        const uint8_t *src = base + o;
        for (uint32_t c0 = 0; c0 < nch; c0 += 64u)
            if (c0 + (uint32_t)lane < nch)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + 16u * (c0 + (uint32_t)lane)), (lptr_t)(dst + c0 * 16u), 16, 0, SK_DMA_AUX);
    };

    // ---- the bands of the two chains: rebuilt when a chain's window width changes
    sk_band band0, band1;
    auto build_band = [&](sk_band &b, int w) {
        const int wq5 = w >> 5;
        // This lane supplies row m' = lane & 31 of A; the hardware puts row m' into accumulator register r of
        // lane half hh with m' = (r & 3) + 8 (r >> 2) + 4 hh; that slot is to be window 16 hh + r of the column
        int mp = l32;
        asm volatile("" : "+v"(mp)); // (keeps the compiler from hoisting 48 per-byte constants out of the read loop)
        const int hh = (mp >> 2) & 1, r = (mp & 3) | ((mp >> 3) << 2);
        const int win = 16 * hh + r;
        auto ones_below = [](int n) -> uint32_t { // 0x01 in the bytes j < n of a dword
            return n >= 4 ? 0x01010101u : (n <= 0 ? 0u : 0x01010101u & ((1u << (8 * n)) - 1u));
        };
        // bytes of positions p .. p + 3 (relative to the column's first byte): 1 where win <= position < win + w
        auto band4 = [&](int p) -> int { return (int)(ones_below(win + w - p) & ~ones_below(win - p)); };
        const int k0 = half * 16; // the first position (within a block) this lane's bytes multiply
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b.A0[j] = band4(k0 + 4 * j);
            b.Aa[j] = band4(k0 + 4 * j + 32 * wq5);
            b.Ab[j] = band4(k0 + 4 * j + 32 * (wq5 + 1));
        }
        const int T = a.craw * w;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int seed = -T;
            asm volatile("" : "+v"(seed)); // one register per accumulator slot, not a scalar copied 16 times per chain
            b.negT[i] = seed;
        }
    };
    const sk_v4i ONES = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    // the 16 sign bits of S - T of a lane's windows: 1024 windows from LDS offset `at` on, window 32 n + 16 hh + i in
    // accumulator i of lane (n, hh); bit (15 - i) of the result: that window is below the threshold
    auto chain = [&](const sk_band &b, int w, uint32_t at) -> uint32_t {
        const int wq5 = w >> 5, bmax = (w + 30) >> 5; // the last block a window of the column's 32 reaches into (<= wq5 + 1)
        const uint8_t *fb = lds + at + 32 * l32 + 16 * half;
        auto frag = [](const uint8_t *p) -> sk_v4i { return *reinterpret_cast<const sk_v4i *>(p); };
        sk_v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.A0, frag(fb), b.negT, 0, 0, 0);
        for (int blk = 1; blk < wq5; ++blk) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(ONES, frag(fb + 32 * blk), d, 0, 0, 0);
        if (wq5 >= 1) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.Aa, frag(fb + 32 * wq5), d, 0, 0, 0);
        if (bmax > wq5) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(b.Ab, frag(fb + 32 * (wq5 + 1)), d, 0, 0, 0);
        uint32_t p = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) p = __builtin_amdgcn_alignbit(p, (uint32_t)d[i], 31);
        return p;
    };
    // one turn: chain 0 (if on0) over the windows from LDS offset at0, chain 1 (if on1) from at1 -> this lane's 32
    // windows (its half's chain, column lane & 31): lt / ge = below / at-or-above the threshold, of the windows that
    // exist (mybase = index of the lane's first window in its read, mynwin = windows of that read; 0 = chain off)
    auto turn = [&](bool on0, int w0, uint32_t at0, bool on1, int w1, uint32_t at1, int mybase, int mynwin, uint32_t &lt, uint32_t &ge) {
        uint32_t p0 = 0, p1 = 0;
        if (on0) p0 = chain(band0, w0, at0);
        if (on1) p1 = chain(UNI ? band0 : band1, w1, at1);
        // lanes 0..31 keep chain 0's columns, lanes 32..63 chain 1's: s[0] = windows 0..15 of the lane's column,
        // s[1] = windows 16..31; bit (31 - s) of M: window mybase + s is below the threshold
        const sk_v2u sw = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
        const uint32_t M = (sw[0] << 16) | sw[1];
        const int nv = mynwin - mybase;
        const uint32_t vmask = nv >= 32 ? ~0u : (nv <= 0 ? 0u : ~(~0u >> nv));
        lt = M & vmask;
        ge = ~M & vmask;
    };
    // what the windows of half h's chain (its first window: number wbase of the read) say about their read:
    // trim.cpp:42 and :61 over them in order
    auto segment = [&](int h, int wbase, uint32_t lt, uint32_t ge, int mybase, sk_band_read &x) {
        if (x.phase == 0) {
            const uint32_t m = (uint32_t)(__builtin_amdgcn_ballot_w64(ge != 0) >> (32 * h));
            if (m) {
                const int t = __builtin_ctz(m);
                x.i0 = wbase + 32 * t + (int)__builtin_amdgcn_readlane((int)ffbh_or_none(ge), 32 * h + t);
                x.phase = 1;
            }
        }
        if (x.phase == 1) { // the first window below the threshold strictly after i0 (from 0 with -x: i0 = -1)
            const int rel = mybase + 31 - x.i0; // how many of this lane's windows, counted from its last, lie after i0
            const uint32_t after = rel >= 32 ? ~0u : (rel <= 0 ? 0u : (1u << rel) - 1u);
            const uint32_t cand = lt & after;
            const uint32_t m = (uint32_t)(__builtin_amdgcn_ballot_w64(cand != 0) >> (32 * h));
            if (m) {
                const int t = __builtin_ctz(m);
                x.i1 = wbase + 32 * t + (int)__builtin_amdgcn_readlane((int)ffbh_or_none(cand), 32 * h + t);
                x.phase = 2;
            }
        }
    };

    // ---- two in-window searches at once, one per half of the wave: the first char at/above (above_) or below the
    // threshold from position from_ on in the read at LDS offset buf_, inside the window of width w_ that starts there
    // (one exists: the window's average is on that side): trim.cpp:46-51, :65-70.  INF if the half is off.
    auto first2 = [&](bool on0, uint32_t buf0, int from0, int w0, bool above0, bool on1, uint32_t buf1, int from1, int w1, bool above1,
                      int &hit0, int &hit1) {
        hit0 = INF;
        hit1 = INF;
        const int nd0 = on0 ? ((from0 & 3) + w0 + 3) >> 2 : 0, nd1 = on1 ? ((from1 & 3) + w1 + 3) >> 2 : 0;
        const int from = half ? from1 : from0;
        const int ndw = half ? nd1 : nd0;
        const uint32_t flip = (half ? above1 : above0) ? 0u : H4;
        const uint32_t *row32 = reinterpret_cast<const uint32_t *>(lds + (half ? buf1 : buf0)) + (from >> 2);
        const int ndmax = max(nd0, nd1);
        for (int it = 0; it < ndmax; it += 32) {
            const int d = it + l32;
            uint32_t f = 0;
            if (d < ndw) {
                f = ge_flags(row32[d], cthr4) ^ flip;
                if (d == 0) f &= ~0u << (8 * (from & 3));
            }
            const uint64_t m = __builtin_amdgcn_ballot_w64(f != 0);
            const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
            if (hit0 == INF && m0) {
                const int t = __builtin_ctz(m0);
                hit0 = 4 * ((from0 >> 2) + it + t) + (__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)f, t)) >> 3);
            }
            if (hit1 == INF && m1) {
                const int t = __builtin_ctz(m1);
                hit1 = 4 * ((from1 >> 2) + it + t) + (__builtin_ctz((uint32_t)__builtin_amdgcn_readlane((int)f, 32 + t)) >> 3);
            }
            if ((hit0 != INF || !on0) && (hit1 != INF || !on1)) break;
        }
    };

    // ---- the rest of trim.cpp:3-116 for one read (two == false: x alone, on the lower half of the wave) or for two
    // (x on the lower half, y on the upper): the cut positions, the range check, the N rule, the length filter, the store
    auto finish = [&](const sk_band_read &x, bool two, const sk_band_read &y) {
        const bool have5x = !a.no5 && x.i0 != INF, have5y = two && !a.no5 && y.i0 != INF;
        const bool found5x = a.no5 || x.i0 != INF, found5y = a.no5 || y.i0 != INF;
        const bool donex = found5x && x.i1 != INF, doney = two && found5y && y.i1 != INF;
        int fivex = 0, threex = x.L, fivey = 0, threey = y.L;
        {
            // two reads: their 5' searches side by side, then their 3' searches; one read: its two searches side by side
            int h0, h1;
            const bool q1on = two ? have5y : donex;
            if (have5x || q1on) {
                first2(have5x, x.buf, have5x ? x.i0 : 0, x.w, true, q1on, y.buf, q1on ? (two ? y.i0 : x.i1) : 0, y.w, two, h0, h1);
                if (have5x && h0 != INF) fivex = h0;
                if (q1on && h1 != INF) {
                    if (two) fivey = h1;
                    else threex = h1;
                }
            }
            if (two && (donex || doney)) {
                first2(donex, x.buf, donex ? x.i1 : 0, x.w, false, doney, y.buf, doney ? y.i1 : 0, y.w, false, h0, h1);
                if (donex && h0 != INF) threex = h0;
                if (doney && h1 != INF) threey = h1;
            }
        }

        // this lane's read: the lower half of the wave has x, the upper half y (or nothing)
        const bool mine = half ? two : true;
        const int myL = half ? (two ? y.L : 0) : x.L;
        const uint32_t mybuf = half ? y.buf : x.buf;
        const int nch = mine ? (myL + 15) >> 4 : 0;
        const int nchmax = max((x.L + 15) >> 4, two ? (y.L + 15) >> 4 : 0);

        // ---- range check (trim.cpp:129): two v_sad_u8 per dword over the whole read, 32 chunks of 16 bytes per trip;
        // only a read with a char out of range looks for where, and whether the reference would have read it
        {
            const sk_v4u *row128 = reinterpret_cast<const sk_v4u *>(lds + mybuf);
            uint32_t sad = 0, visited = 0;
            for (int c0 = 0; c0 < nchmax; c0 += 32) {
                const int c = c0 + l32;
                if (c < nch) {
                    const sk_v4u q = row128[c];
                    const int n = myL - 16 * c;
                    if (n >= 16) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sad = __builtin_amdgcn_sad_u8(q[u], min4, sad);
                            sad = __builtin_amdgcn_sad_u8(q[u], max4, sad);
                        }
                    } else { // the read ends inside this chunk
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t xq = first_bytes(q[u], n - 4 * u, min4); // fillers are legal chars
                            sad = __builtin_amdgcn_sad_u8(xq, min4, sad);
                            sad = __builtin_amdgcn_sad_u8(xq, max4, sad);
                        }
                    }
                    ++visited;
                }
            }
            const bool bad = sad != visited * clean16;
            const uint64_t mb = __builtin_amdgcn_ballot_w64(bad);
            if (mb) {
                int p = INF;
                if (bad) {
                    for (int c = l32; c < nch && p == INF; c += 32) {
                        const sk_v4u q = row128[c];
#pragma unroll
                        for (int u = 3; u >= 0; --u) {
                            const uint32_t f = keep_first(bad_flags(q[u], min4, hi4), myL - 16 * c - 4 * u);
                            if (f) p = 16 * c + 4 * u + (__builtin_ctz(f) >> 3);
                        }
                    }
                }
                p = row_min(p); // the first bad position of each row of 16 lanes, in its lanes
                const int pbx = min(__builtin_amdgcn_readlane(p, 0), __builtin_amdgcn_readlane(p, 16));
                const int pby = min(__builtin_amdgcn_readlane(p, 32), __builtin_amdgcn_readlane(p, 48));
                const int touchedx = donex ? x.i1 + x.w : x.L, touchedy = doney ? y.i1 + y.w : y.L;
                if (pbx < touchedx && lane == 0) report_error(errword, x.r, pbx, (int)(int8_t)lds[x.buf + (uint32_t)pbx]);
                if (two && pby < touchedy && lane == 0) report_error(errword, y.r, pby, (int)(int8_t)lds[y.buf + (uint32_t)pby]);
            }
        }

        // ---- the N rule: trim.cpp:86-98 (lowercase n: cut before it; only uppercase N: cut = -2)
        if (HAS_SEQ) {
            const sk_v4u *srow = reinterpret_cast<const sk_v4u *>(lds + mybuf + rb);
            uint32_t nlo = NONE, anyN = 0; // bit index of the first lowercase n; any uppercase N
            for (int c0 = 0; c0 < nchmax; c0 += 32) {
                const int c = c0 + l32;
                if (c < nch) {
                    const sk_v4u q = srow[c];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t xx = first_bytes(q[u], myL - 16 * c - 4 * u, 0u);
                        const uint32_t yy = (xx | 0x20202020u) ^ 0x6e6e6e6eu;
                        const uint32_t either = ~(((yy & 0x7f7f7f7fu) + 0x7f7f7f7fu) | yy) & H4; // exact zero-byte flags
                        const uint32_t lower = either & (xx << 2); // bit 5 of the byte moved onto its flag
                        nlo = min(nlo, __builtin_elementwise_add_sat(ffbl_or_none(lower), (uint32_t)(8 * (16 * c + 4 * u))));
                        anyN |= either ^ lower;
                    }
                }
            }
            const int nl = row_min(nlo == NONE ? INF : (int)(nlo >> 3));
            const int nlx = min(__builtin_amdgcn_readlane(nl, 0), __builtin_amdgcn_readlane(nl, 16));
            const int nly = min(__builtin_amdgcn_readlane(nl, 32), __builtin_amdgcn_readlane(nl, 48));
            const uint64_t mN = __builtin_amdgcn_ballot_w64(anyN != 0);
            if (nlx != INF) threex = nlx - 1;
            else if ((uint32_t)mN) threex = -2;
            if (nly != INF) threey = nly - 1;
            else if ((uint32_t)(mN >> 32)) threey = -2;
        }
        if (!found5x || (threex - fivex < a.lthr)) { // trim.cpp:103-108
            fivex = -1;
            threex = -1;
        }
        if (!found5y || (threey - fivey < a.lthr)) {
            fivey = -1;
            threey = -1;
        }
        if (l32 == 0 && mine) out[half ? y.r : x.r] = half ? sk_cut_dev{fivey, threey} : sk_cut_dev{fivex, threex};
        // every LDS read of these reads is done before a later read's DMA may overwrite their slots
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // ---- which reads: every read of the batch, dealt one by one (read blockIdx.x, + gridDim.x, ...); or
    // (a.buf_bytes != 0) only the reads of the 64-read tiles sk_scan_tile_any_kernel left (the same test as there) --
    // if it left any: it has put this scan's number into the word after the error word for every tile it skipped.
    // Then runs of 8 consecutive reads are dealt to the waves, and a wave asks the question for the tile its run
    // lies in; the probe leaves read 64 tile + l's start and length in lane l.
    constexpr bool leftovers = LEFT;
    if (leftovers && scalar_load(errword + 1) != a.scan_id) return;
    const uint64_t G = gridDim.x;
    const uint64_t n_runs = (a.n_reads + 7) / 8;
    uint64_t run = blockIdx.x; // leftovers: the run being dealt out
    int k = 8;                 // ... and the next read of it (8: probe the next run first)
    bool first = true;
    sk_rag_tile pr;
    pr.start = 0, pr.span = 0, pr.rowoff = 0, pr.len = 0, pr.lmax = 0;
    uint64_t rr = blockIdx.x;  // every read: the next one
    auto advance = [&](uint64_t &r, uint64_t &o, int &L) -> bool { // the wave's next read, false when there is none
        if (!leftovers) {
            if (rr >= a.n_reads) return false;
            r = rr;
            rr += G;
            locate(r, o, L);
            return true;
        }
        for (;;) {
            if (k == 8) {
                if (!first) run += G;
                first = false;
                if (run >= n_runs) return false;
                pr = rag_probe((run * 8) >> 6, lane, offsets, lengths, a);
                if (rag_tile_fits(pr, a.buf_bytes)) continue; // the tile kernel took this tile
                k = 0;
            }
            r = run * 8 + (uint64_t)k;
            ++k;
            if (r >= a.n_reads) {
                k = 8;
                continue;
            }
            const int idx = (int)(r & 63u);
            o = pr.start + (uint32_t)__builtin_amdgcn_readlane((int)pr.rowoff, idx);
            L = __builtin_amdgcn_readlane(pr.len, idx);
            return true;
        }
    };

    // ---- the ring.  `inring` reads are issued and not yet scanned, the oldest in slot cslot, the next one goes into
    // slot islot; `inflight` = vector-memory instructions of those reads.
    auto pieces_of = [&](int L) -> int { return (int)REGIONS * (int)(((((uint32_t)L + 15u) >> 4) + 63u) >> 6); };
    uint32_t islot = 0, cslot = 0, inring = 0;
    int inflight = 0;
    bool drained = false;
    // UNI: reads n_ring and beyond do not go through the ring (the batch's last read when the 16-byte chunks of its
    // image would reach past the batch): wave 0 takes them at the end
    const uint64_t n_ring = UNI ? a.n_reads - ((a.n_reads && (uint64_t)a.stride * (a.n_reads - 1) + (((uint64_t)a.read_len + 15u) & ~15ull) > batch_end) ? 1u : 0u) : 0;
    uint64_t r_issue = blockIdx.x, r_head = blockIdx.x; // UNI: the next read to issue / the read at the head of the ring
    auto issue = [&]() {
        uint64_t r = 0, o = 0;
        int L = 0;
        if (UNI) {
            if (r_issue >= n_ring) {
                drained = true;
                return;
            }
            uint8_t *nb = lds + islot * slot_bytes;
            stage(qual, r_issue * a.stride, (int)a.read_len, nb);
            if (HAS_SEQ) stage(seq, r_issue * a.stride, (int)a.read_len, nb + rb);
            r_issue += gridDim.x;
        } else {
            if (!advance(r, o, L)) {
                drained = true;
                return;
            }
            if (lane == 0) *reinterpret_cast<sk_v4u *>(headers + 4u * islot) = sk_v4u{(uint32_t)r, (uint32_t)L, (uint32_t)o, (uint32_t)(o >> 32)};
            if (staged(L, o)) {
                uint8_t *nb = lds + islot * slot_bytes;
                stage(qual, o, L, nb);
                if (HAS_SEQ) stage(seq, o, L, nb + rb);
                inflight += pieces_of(L);
            }
        }
        islot = islot + 1u == NSLOT ? 0u : islot + 1u;
        ++inring;
    };
    auto fresh = [&](sk_band_read &x, uint32_t r, int L, uint32_t slot) {
        x.r = r;
        x.L = L;
        x.buf = slot * slot_bytes;
        x.w = window_of(L);
        x.nwin = L - x.w + 1;
        x.phase = a.no5 ? 1 : 0;
        x.i0 = a.no5 ? -1 : INF;
        x.i1 = INF;
    };
    auto header = [&](uint32_t slot, sk_band_read &x, uint64_t &o) {
        const sk_v4u h = *reinterpret_cast<const sk_v4u *>(headers + 4u * slot);
        o = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)h[3]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)h[2]);
        fresh(x, (uint32_t)__builtin_amdgcn_readfirstlane((int)h[0]), __builtin_amdgcn_readfirstlane((int)h[1]), slot);
    };
    // the wave's next turn: x (and, two == true, y) are the reads at the head of the ring, already taken off its
    // counters; `pending` = vector-memory instructions of the reads still behind them.  A short read (at most 1024
    // windows) shares its turn with the next one if that is short too.  Reads that do not go through the ring are
    // dealt with on the way.  false: nothing left.
    sk_band_read x, y;
    bool two = false;
    int pending = 0;
    const int uni_pieces = UNI ? pieces_of((int)a.read_len) : 0;
    auto fetch = [&]() -> bool {
        for (;;) {
            while (!drained && inring < NSLOT) issue();
            if (inring == 0) return false;
            const uint32_t nslot = cslot + 1u == NSLOT ? 0u : cslot + 1u;
            if (UNI) {
                fresh(x, (uint32_t)r_head, (int)a.read_len, cslot);
                y = x;
                two = x.nwin <= 1024 && inring >= 2;
                if (two) fresh(y, (uint32_t)(r_head + gridDim.x), (int)a.read_len, nslot);
                r_head += (two ? 2u : 1u) * (uint64_t)gridDim.x;
                inring -= two ? 2u : 1u;
                pending = (int)inring * uni_pieces;
                cslot = two ? (nslot + 1u == NSLOT ? 0u : nslot + 1u) : nslot;
                return true;
            }
            uint64_t ox, oy;
            header(cslot, x, ox);
            if (!staged(x.L, ox)) {
                band_other<HAS_SEQ>(qual, seq, ox, x.L, x.r, out, errword, a);
                cslot = nslot;
                --inring;
                continue;
            }
            inflight -= pieces_of(x.L);
            two = false;
            y = x;
            if (x.nwin <= 1024 && inring >= 2) {
                header(nslot, y, oy);
                two = staged(y.L, oy) && y.nwin <= 1024;
                if (!two) y = x;
            }
            if (two) inflight -= pieces_of(y.L);
            pending = inflight;
            cslot = two ? (nslot + 1u == NSLOT ? 0u : nslot + 1u) : nslot;
            inring -= two ? 2u : 1u;
            return true;
        }
    };
    if (UNI && blockIdx.x == 0 && n_ring < a.n_reads) // (the batch's last read, see n_ring)
        band_other<HAS_SEQ>(qual, seq, n_ring * a.stride, (int)a.read_len, n_ring, out, errword, a);
    bool have = fetch();
    while (have) {
        // ---- the bands of the two chains are built here, for every turn that follows with the same two window widths
        // (inside the loop of turns a conditional rebuild makes the compiler keep two copies of both sets: 60 registers)
        const int w0 = x.w, w1 = y.w;
        build_band(band0, w0);
        if (!UNI) build_band(band1, w1);
        do {
            wait_vmcnt(pending); // loads return in order: everything older than the later reads' pieces has landed
            // two reads: a chain each.  One read: both chains, 2048 windows per turn, until both windows are found
            const uint32_t step1 = two ? 0u : 1024u; // where chain 1 starts relative to chain 0
            for (int wbase = 0; wbase < x.nwin; wbase += 2048) {
                const bool on1 = two || wbase + 1024 < x.nwin;
                uint32_t lt, ge;
                const int mybase = wbase + (int)(half ? step1 : 0u) + 32 * l32;
                turn(true, w0, x.buf + (uint32_t)wbase, on1, w1, y.buf + (uint32_t)wbase + step1, mybase, half ? (on1 ? y.nwin : 0) : x.nwin, lt, ge);
                segment(0, wbase, lt, ge, mybase, x);
                if (on1) {
                    if (two) {
                        segment(1, 0, lt, ge, mybase, y);
                    } else { // the same read goes on in chain 1
                        segment(1, wbase + 1024, lt, ge, mybase, x);
                    }
                }
                if (two || x.phase == 2) break;
            }
            finish(x, two, y);
            have = fetch();
        } while (have && x.w == w0 && y.w == w1);
    }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_band(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                     const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                     const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream)
{
    // max_len = the longest read the caller expects (0 = unknown); longer reads still come out right, from
    // global memory (scan_read_global)
    if (a->n_reads == 0) return hipSuccess;
    if (max_len == 0 || max_len > 8192) max_len = 8192;
    if (max_len < 64) max_len = 64;
    sk_scan_args at = *a;
    at.team_maxlen = (uint32_t)max_len;
    const uint64_t wmax = max_len / 10 ? max_len / 10 : max_len;
    const bool has_seq = a->truncn != 0;
    const uint32_t regions = has_seq ? 2u : 1u;
    at.team_rbuf = (uint32_t)((max_len + 16 + 15) & ~(uint64_t)15);
    // reads in the ring: the one or two being scanned and about 8 KiB of reads in flight behind them
    static const int depth_env = [] { const char *e = getenv("SK_BAND_DEPTH"); return e ? atoi(e) : 0; }();
    uint32_t depth = depth_env > 0 ? (uint32_t)depth_env : (uint32_t)(8192 / max_len);
    if (depth < 2) depth = 2;
    if (depth > 8) depth = 8;
    at.stream_nb = depth + 2;
    // the ring, the overhang behind it (the KiB of windows a chain computes past a read's last, the band's reach,
    // slack), the slot headers
    const uint32_t ring = at.stream_nb * regions * at.team_rbuf;
    const uint32_t overhang = (uint32_t)((1024 + wmax + 64 + 128 + 15) & ~(uint64_t)15);
    at.stream_tbl = ring + overhang; // where the headers start
    const uint32_t lds_bytes = at.stream_tbl + 16u * at.stream_nb;
    if (lds_bytes > SK_LDS_PER_CU) return hipErrorInvalidValue;
    int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
    static const int wave_cap = [] { const char *e = getenv("SK_BAND_WAVES"); return e ? atoi(e) : 16; }();
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu > 4) per_cu &= ~3; // the same number of waves on every SIMD
    const uint64_t n_units = a->buf_bytes ? (a->n_reads + 7) / 8 : a->n_reads;
    uint64_t grid = (uint64_t)cu_count * per_cu;
    if (grid > n_units) grid = n_units;
    if (grid == 0) return hipSuccess;
    auto launch = [&](auto kern) {
        const kernel_facts facts = prepare_kernel(kern);
        if (facts.status != hipSuccess) return facts.status;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, qual, seq, offsets, lengths, out,
                           errword, at);
        return hipGetLastError();
    };
    if (a->buf_bytes) return has_seq ? launch(sk_scan_band_kernel<true, true, false>) : launch(sk_scan_band_kernel<false, true, false>);
    // equal lengths at a fixed stride, every read with something to scan and short enough for a slot: the lean form
    if (!offsets && !lengths && a->read_len > 0 && a->read_len >= (uint32_t)a->lthr && a->read_len <= at.team_maxlen)
        return has_seq ? launch(sk_scan_band_kernel<true, false, true>) : launch(sk_scan_band_kernel<false, false, true>);
    return has_seq ? launch(sk_scan_band_kernel<true, false, false>) : launch(sk_scan_band_kernel<false, false, false>);
}
