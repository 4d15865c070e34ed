// sk_band.hip -- the general kernel for MEDIUM reads (beyond a 64-read LDS tile, up to a few kilobases): a wave per
// read, the window sums of the whole read taken from the integer matrix pipe; see the block comment below.
#include "sk_kernel_common.h"

// ------------------------------------------------------------------------------------------
// sk_scan_band_kernel: reference src/trim.cpp:3-116 for one read per wave, the read resident in LDS.
//
// The lane-per-read tile kernels stop at rows of 512 bytes (64 rows must fit a wave's LDS buffer), the streaming
// kernel (sk_stream.hip) only pays for itself from ~4 kb on (its cost per READ is several hundred scalar
// instructions).  Between them round 2 ran teams of 16 lanes per read (sk_team.hip) on the vector ALU: 12 vector
// instructions per 4 windows per lane, 0.18-0.31 of the HBM peak.  Here the windows of ONE read are the columns of
// a matrix product:
//     S[32 n + m] - T = sum over blocks blk of  band_blk[m][k] * c[32 (n + blk) + k]  - T      (m, k in 0..31)
// i.e. D = sum_blk A_blk x B_blk + (-T) with B_blk = the read's bytes from 32 blk on, taken as 32 columns of 32
// consecutive bytes -- a lane's B operand is 16 CONSECUTIVE bytes of the read (one ds_read_b128, the wave's 64
// lanes together read one contiguous KiB), and one chain of v_mfma_i32_32x32x32_i8 gives 1024 windows.  The band
// of a window of width w covers (w + 30) / 32 + 1 blocks: the first and the last two are partial (per-lane
// constants, rebuilt when w changes, which in a batch of equal lengths is never), the ones in between are all
// ones.  Exact in int32 (a window sums at most a few hundred bytes).  Rows of A are permuted as in the tile kernel,
// so that a lane's 16 accumulators are 16 consecutive windows and one v_alignbit per window collects the signs;
// two chains (windows [2048 p, +1024) and the next 1024) fill the two halves of the wave, and after one
// v_permlane32_swap lane l holds the 32-window mask of windows 2048 p + 32 l ...: window order = lane order, so
// "the first window at/above the threshold" and "the first one below it after that" (trim.cpp:42, :61) are a
// v_ffbh and a wave minimum per 2048 windows.  Cost per KiB of read: (w + 30) / 32 + 1 MFMAs (5 at 1 kb, 15 at
// 4 kb; the matrix pipe is otherwise idle) and ~50 vector instructions, whatever the data -- averages hovering at
// the threshold cost nothing extra.
//
// A wave keeps a RING of reads in LDS (a.stream_nb slots; LDS-DMA, 16 bytes per lane, source address per lane: the
// image of a read starts aligned wherever it lies in the batch): the reads after the one being scanned are in flight,
// ~8 KiB per wave -- what the device needs outstanding to stream (the first version staged ONE read ahead: 1 KiB in
// flight per wave at 1 kb, 1.1 TB/s).  Loads return in order, so the wait for a read is a counted s_waitcnt vmcnt(n).
// With a.buf_bytes != 0 the kernel takes only the 64-read tiles sk_scan_tile_any_kernel left.
// A read longer than the buffers (a.team_maxlen) is scanned from global memory (scan_read_global: correctness path).
// ------------------------------------------------------------------------------------------
template <bool HAS_SEQ>
__global__ void __launch_bounds__(64)
sk_scan_band_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                    const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lengths,
                    sk_cut_dev *__restrict__ out, unsigned long long *errword, sk_scan_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x; // single-wave workgroups
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
    const int range = a.qmax - a.qmin;

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
    auto staged = [&](int L) { return L > 0 && L >= a.lthr && L <= (int)a.team_maxlen; };

    // ---- the read into LDS: chunk c (16 bytes) by lane c mod 64.  Returns the number of vector-memory instructions
    // issued, or -1 when the read's last chunk would leave the batch (those bytes are then copied one by one and the
    // caller waits for everything)
    auto stage = [&](const uint8_t *base, uint64_t o, int L, uint8_t *dst) -> int {
        const uint32_t nch = ((uint32_t)L + 15u) >> 4;
        const bool all_inside = o + 16ull * nch <= batch_end; // wave-uniform
        const uint8_t *src = base + o;
        int pieces = 0;
        for (uint32_t c0 = 0; c0 < nch; c0 += 64u, ++pieces) {
            const uint32_t so = 16u * (c0 + (uint32_t)lane);
            if (c0 + (uint32_t)lane < nch) {
                if (all_inside || o + so + 16u <= batch_end) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + so), (lptr_t)(dst + c0 * 16u), 16, 0, SK_DMA_AUX);
                } else { // the batch ends inside this chunk
                    for (uint32_t q = 0; q < 16u && o + so + q < batch_end; ++q) dst[so + q] = src[so + q];
                }
            }
        }
        return all_inside ? pieces : -1;
    };

    // ---- the band of the current window width (see the header): rebuilt when w changes
    int w_cur = -1, wq5 = 0, bmax = 0;
    sk_v4i A0 = {0, 0, 0, 0}, Aa = {0, 0, 0, 0}, Ab = {0, 0, 0, 0};
    sk_v16i negT;
    auto build_band = [&](int w) {
        w_cur = w;
        wq5 = w >> 5;
        bmax = (w + 30) >> 5; // the last block a window of the column's 32 reaches into (<= wq5 + 1)
        // This lane supplies row m' = lane & 31 of A; the hardware puts row m' into accumulator register r of
        // lane half hh with m' = (r & 3) + 8 (r >> 2) + 4 hh; that slot is to be window 16 hh + r of the column
        int mp = lane & 31;
        asm volatile("" : "+v"(mp)); // (keeps the compiler from hoisting 48 per-byte constants out of the read loop)
        const int hh = (mp >> 2) & 1, r = (mp & 3) | ((mp >> 3) << 2);
        const int win = 16 * hh + r;
        auto ones_below = [](int n) -> uint32_t { // 0x01 in the bytes j < n of a dword
            return n >= 4 ? 0x01010101u : (n <= 0 ? 0u : 0x01010101u & ((1u << (8 * n)) - 1u));
        };
        // bytes of positions p .. p + 3 (relative to the column's first byte): 1 where win <= position < win + w
        auto band4 = [&](int p) -> int { return (int)(ones_below(win + w - p) & ~ones_below(win - p)); };
        const int k0 = (lane >> 5) * 16; // the first position (within a block) this lane's bytes multiply
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A0[j] = band4(k0 + 4 * j);
            Aa[j] = band4(k0 + 4 * j + 32 * wq5);
            Ab[j] = band4(k0 + 4 * j + 32 * (wq5 + 1));
        }
        const int T = a.craw * w;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int seed = -T;
            asm volatile("" : "+v"(seed)); // one register per accumulator slot, not a scalar copied 16 times per chain
            negT[i] = seed;
        }
    };
    const sk_v4i ONES = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    auto frag = [](const uint8_t *p) -> sk_v4i { return *reinterpret_cast<const sk_v4i *>(p); };
    // S - T of the 1024 windows from `fb` on: accumulator i of lane (n, hh) = window 32 n + 16 hh + i
    auto chain = [&](const uint8_t *fb) -> sk_v16i {
        sk_v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, frag(fb), negT, 0, 0, 0);
        for (int blk = 1; blk < wq5; ++blk) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(ONES, frag(fb + 32 * blk), d, 0, 0, 0);
        if (wq5 >= 1) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(Aa, frag(fb + 32 * wq5), d, 0, 0, 0);
        if (bmax > wq5) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ab, frag(fb + 32 * (wq5 + 1)), d, 0, 0, 0);
        return d;
    };

    // ---- one staged read: bq (and bs) hold its quality (sequence) bytes from offset 0
    auto scan = [&](uint64_t r, int L, const uint8_t *bq, const uint8_t *bs) {
        const int w = window_of(L);
        const int nwin = L - w + 1;
        if (w != w_cur) build_band(w);
        const uint32_t *row32 = reinterpret_cast<const uint32_t *>(bq);
        const sk_v4u *row128 = reinterpret_cast<const sk_v4u *>(bq);

        // the first char at/above (below) the threshold from `from` on; one exists inside the window that starts
        // there (its average is on that side): trim.cpp:46-51, :65-70
        auto first_char = [&](int from, bool above) -> int {
            const int d0 = from >> 2, ndw = ((from & 3) + w + 3) >> 2;
            for (int it = 0; it < ndw; it += 64) {
                const int d = it + lane;
                uint32_t f = 0;
                if (d < ndw) {
                    f = ge_flags(row32[d0 + d], cthr4);
                    if (!above) f ^= H4;
                    if (d == 0) f &= ~0u << (8 * (from & 3));
                }
                const uint64_t m = __builtin_amdgcn_ballot_w64(f != 0);
                if (m) {
                    const int t = __builtin_ctzll(m);
                    const uint32_t ft = (uint32_t)__builtin_amdgcn_readlane((int)f, t);
                    return 4 * (d0 + it + t) + (__builtin_ctz(ft) >> 3);
                }
            }
            return INF;
        };

        // ---- all windows, 2048 per turn (trim.cpp:34-81 without the breaks; the turns stop once both windows are found)
        int phase = a.no5 ? 1 : 0; // 0: looking for the first S >= T, 1: for the first S < T after it, 2: both found
        int i0 = a.no5 ? -1 : INF, i1 = INF;
        const uint8_t *fb = bq + 32 * (lane & 31) + 16 * (lane >> 5);
        for (int wbase = 0; wbase < nwin && phase < 2; wbase += 2048) {
            const bool two = wbase + 1024 < nwin; // wave-uniform
            const sk_v16i d0 = chain(fb + wbase);
            uint32_t p0 = 0, p1 = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) p0 = __builtin_amdgcn_alignbit(p0, (uint32_t)d0[i], 31);
            if (two) {
                const sk_v16i d1 = chain(fb + wbase + 1024);
#pragma unroll
                for (int i = 0; i < 16; ++i) p1 = __builtin_amdgcn_alignbit(p1, (uint32_t)d1[i], 31);
            }
            // lanes 0..31 keep the first chain's columns, lanes 32..63 the second's: s[0] = windows 0..15 of the
            // lane's column, s[1] = windows 16..31; bit (31 - s) of M: window base + s is below the threshold
            const sk_v2u sw = __builtin_amdgcn_permlane32_swap(p0, p1, false, false);
            const uint32_t M = (sw[0] << 16) | sw[1];
            const int base = wbase + 32 * lane;
            const int nv = nwin - base;
            const uint32_t vmask = nv >= 32 ? ~0u : (nv <= 0 ? 0u : ~(~0u >> nv));
            const uint32_t lt = M & vmask, ge = ~M & vmask;
            if (phase == 0) { // trim.cpp:42
                i0 = wave_min(ge ? base + (int)ffbh_or_none(ge) : INF);
                if (i0 != INF) phase = 1;
            }
            if (phase == 1) { // trim.cpp:61: the first window below the threshold strictly after i0 (from 0 with -x)
                const int rel = base + 31 - i0; // how many of this lane's windows, counted from its last, lie after i0
                const uint32_t after = rel >= 32 ? ~0u : (rel <= 0 ? 0u : (1u << rel) - 1u);
                const uint32_t cand = lt & after;
                i1 = wave_min(cand ? base + (int)ffbh_or_none(cand) : INF);
                if (i1 != INF) phase = 2;
            }
        }
        const bool have5 = !a.no5 && i0 != INF;
        const bool found5 = a.no5 || i0 != INF;
        const bool done = found5 && i1 != INF;
        int five = 0, three = L;
        if (have5) {
            five = first_char(i0, true);
            if (five == INF) five = 0;
        }
        if (done) {
            three = first_char(i1, false);
            if (three == INF) three = L;
        }

        // ---- range check (trim.cpp:129): two v_sad_u8 per dword over the whole read; only a read with a char out
        // of range looks for where, and whether the reference would have read it
        {
            const int nch = (L + 15) >> 4;
            uint32_t sad = 0, visited = 0;
            for (int c = lane; c < nch; c += 64) {
                const sk_v4u x = row128[c];
                const int n = L - 16 * c;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t xq = first_bytes(x[u], n - 4 * u, min4); // fillers are legal chars
                    sad = __builtin_amdgcn_sad_u8(xq, min4, sad);
                    sad = __builtin_amdgcn_sad_u8(xq, max4, sad);
                }
                ++visited;
            }
            const bool bad = sad != visited * (uint32_t)(16 * range);
            if (__builtin_amdgcn_ballot_w64(bad)) {
                int p = INF;
                if (bad) {
                    for (int c = lane; c < nch && p == INF; c += 64) {
                        const sk_v4u x = row128[c];
#pragma unroll
                        for (int u = 3; u >= 0; --u) {
                            const uint32_t f = keep_first(bad_flags(x[u], min4, hi4), L - 16 * c - 4 * u);
                            if (f) p = 16 * c + 4 * u + (__builtin_ctz(f) >> 3);
                        }
                    }
                }
                const int pb = wave_min(p);
                const int touched = done ? i1 + w : L;
                if (pb < touched && lane == 0) report_error(errword, r, pb, (int)(int8_t)bq[pb]);
            }
        }

        // ---- the N rule: trim.cpp:86-98 (lowercase n: cut before it; only uppercase N: cut = -2)
        if (HAS_SEQ) {
            const sk_v4u *srow = reinterpret_cast<const sk_v4u *>(bs);
            uint32_t nlo = NONE, anyN = 0; // bit index of the first lowercase n; any uppercase N
            const int nch = (L + 15) >> 4;
            for (int c = lane; c < nch; c += 64) {
                const sk_v4u x = srow[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t xx = first_bytes(x[u], L - 16 * c - 4 * u, 0u);
                    const uint32_t y = (xx | 0x20202020u) ^ 0x6e6e6e6eu;
                    const uint32_t either = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & H4; // exact zero-byte flags
                    const uint32_t lower = either & (xx << 2); // bit 5 of the byte moved onto its flag
                    nlo = min(nlo, __builtin_elementwise_add_sat(ffbl_or_none(lower), (uint32_t)(8 * (16 * c + 4 * u))));
                    anyN |= either ^ lower;
                }
            }
            const int nl = wave_min(nlo == NONE ? INF : (int)(nlo >> 3));
            anyN = wave_or(anyN);
            if (nl != INF) three = nl - 1;
            else if (anyN) three = -2;
        }
        if (!found5 || (three - five < a.lthr)) { // trim.cpp:103-108
            five = -1;
            three = -1;
        }
        if (lane == 0) out[r] = sk_cut_dev{five, three};
        // every LDS read of this read is done before a later read's DMA may overwrite the buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // a read that does not go through LDS: nothing to scan (trim.cpp:21), or longer than the buffers
    auto other = [&](uint64_t r, uint64_t o, int L) {
        if (L > 0 && L >= a.lthr) {
            const sk_cut_dev cut = scan_read_global<HAS_SEQ>(qual + o, HAS_SEQ ? seq + o : nullptr, L, r, lane, a, errword);
            if (lane == 0) out[r] = cut;
        } else if (lane == 0) {
            out[r] = sk_cut_dev{-1, -1};
        }
    };

    // ---- which reads: every read of the batch, dealt one by one (read blockIdx.x, + gridDim.x, ...); or
    // (a.buf_bytes != 0) only the reads of the 64-read tiles sk_scan_tile_any_kernel left (the same test as there) --
    // if it left any: it has put this scan's number into the word after the error word for every tile it skipped.
    // Then runs of 8 consecutive reads are dealt to the waves, and a wave asks the question for the tile its run
    // lies in; the probe leaves read 64 tile + l's start and length in lane l.
    const bool leftovers = a.buf_bytes != 0;
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

    // ---- the ring: read number `issued` goes into slot issued mod NSLOT; up to NSLOT - 1 reads are in flight behind the
    // one being scanned.  `inflight` = vector-memory instructions of the reads issued and not yet scanned.
    auto pieces_of = [&](int L) -> int { return staged(L) ? (int)REGIONS * (int)(((((uint32_t)L + 15u) >> 4) + 63u) >> 6) : 0; };
    uint32_t issued = 0, consumed = 0;
    int inflight = 0;
    bool sync_all = false; // a read was copied carefully (the end of the batch): from then on wait for everything
    bool drained = false;
    auto issue = [&]() {
        uint64_t r = 0, o = 0;
        int L = 0;
        if (!advance(r, o, L)) {
            drained = true;
            return;
        }
        const uint32_t slot = issued % NSLOT;
        if (lane == 0) *reinterpret_cast<sk_v4u *>(headers + 4u * slot) = sk_v4u{(uint32_t)r, (uint32_t)L, (uint32_t)o, (uint32_t)(o >> 32)};
        if (staged(L)) {
            uint8_t *nb = lds + slot * slot_bytes;
            const int pq = stage(qual, o, L, nb);
            const int ps = HAS_SEQ ? stage(seq, o, L, nb + rb) : 0;
            if (pq < 0 || ps < 0) sync_all = true;
            inflight += pieces_of(L);
        }
        ++issued;
    };
    for (uint32_t d = 0; d + 1 < NSLOT && !drained; ++d) issue();
    while (consumed < issued) {
        if (!drained) issue(); // (into the one free slot)
        const uint32_t slot = consumed % NSLOT;
        const sk_v4u h = *reinterpret_cast<const sk_v4u *>(headers + 4u * slot);
        const uint64_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)h[0]);
        const int L = __builtin_amdgcn_readfirstlane((int)h[1]);
        const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)h[3]) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)h[2]);
        if (staged(L)) {
            inflight -= pieces_of(L);
            wait_vmcnt(sync_all ? 0 : inflight); // loads return in order: everything older than the later reads' pieces has landed
            const uint8_t *cb = lds + slot * slot_bytes;
            scan(r, L, cb, cb + rb);
        } else {
            other(r, o, L);
        }
        ++consumed;
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
    // reads in flight behind the one being scanned: about 8 KiB of them per wave, at least 2, at most 8
    static const int depth_env = [] { const char *e = getenv("SK_BAND_DEPTH"); return e ? atoi(e) : 0; }();
    uint32_t depth = depth_env > 0 ? (uint32_t)depth_env : (uint32_t)(8192 / max_len);
    if (depth < 2) depth = 2;
    if (depth > 8) depth = 8;
    at.stream_nb = depth + 1;
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
    return has_seq ? launch(sk_scan_band_kernel<true>) : launch(sk_scan_band_kernel<false>);
}
