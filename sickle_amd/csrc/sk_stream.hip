// sk_stream.hip -- the general kernel for long reads: a wave per read, the read STREAMED through a ring of
// 1 KiB blocks in LDS; see the block comment below.
#include "sk_kernel_common.h"

// ------------------------------------------------------------------------------------------
// sk_scan_stream_kernel: reads of any length (reference src/trim.cpp:3-116), one wave per read, no read
// ever resident as a whole.
//
// sk_scan_team_kernel stages a read whole, waits, scans: its LDS buffer is as long as the longest read
// (5 waves per CU at 30 kb), and every read pays its load latency in full.  Here a wave owns a SPAN of
// consecutive reads and a ring of NB blocks of 1 KiB; block b of a read = its bytes [1024 b, 1024 b + 1024),
// fetched by ONE LDS-DMA instruction (16 bytes per lane, the source address per lane: a read starts a block
// wherever it lies in the batch).  The blocks of the span -- read after read, with -n a read's sequence
// blocks after its quality blocks -- form one sequence; the DMA runs DEPTH (3) blocks ahead of the scan with
// counted waits (s_waitcnt vmcnt(n): loads return in order), across read boundaries, so the pipeline only
// drains at the end of a span.  8 KiB of ring + 2 KiB of table per wave at 30 kb: 16 waves per CU.  A wave's
// span: the batch cut at equal cost (bytes + a.stream_read_cost per read) by a search in `offsets`.
//
// The scan of a quality block (lane t holds chunk k = 64 b + t, the bytes [16 k, 16 k + 16)):
//   1. range check (two v_sad_u8 per dword) and chunk sum; a wave prefix scan turns the sums into
//      P16[k] = sum of the read's bytes before 16 k, kept in a table ring beside the blocks;
//   2. the window that ENDS in chunk k and starts on a multiple of 16: i = 16 a, a = k - (w >> 4):
//      S_i - T = P16[k] + (first w & 15 bytes of the chunk) - P16[a] - T =: v_a.  One per lane, exact.
//   3. the 15 windows between two aligned ones differ from them by at most 8 steps of at most
//      qmax - qmin each (chars in range; a char out of range among the bytes the reference reads is an
//      error whatever the cut): if v_{a-1} and v_a lie on one side of 0 by 8 (qmax - qmin) or more, so do
//      all windows between them.  Two ballots give the cells that may hold the first window at/above the
//      threshold and the first one below it; only those are evaluated window by window -- the flagged cell and
//      the three after it, a window per lane: byte differences, a wave scan, a ballot: once or twice per read.
//   4. the read's state (looking for the first S >= T; for the first S < T after it; done) is wave-uniform
//      and steps through the flagged cells in order; trim.cpp:46-51 and :65-70 (the first char at/above
//      resp. below the threshold inside the window) read the ring 64 bytes at a time.
// The trailing bytes a cell needs lie w + 16 behind the block being scanned: NB = depth + 2 + ceil((w + 16)
// / 1024) blocks, sized by the launcher from the caller's longest-read hint; a read whose window does not
// fit is scanned from global memory by the whole wave (scan_read_global: correctness path).
// With a.buf_bytes != 0 the kernel takes only the 64-read tiles sk_scan_tile_any_kernel left.
//
// What bounds it is instruction issue, the scalar unit first (one per CU for all its waves): hence the
// lockstep run in the block loop, the uniform constants kept in vector registers, and the loader that looks
// one read ahead.  DESIGN.md 4.3 has the counters of each step.
// ------------------------------------------------------------------------------------------
namespace {

// inclusive prefix sum over the wave, six DPP adds: within the rows, then lane 15 of row 0 / 2 into rows 1 / 3,
// then lane 31 into rows 2 and 3
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31
    return v;
}

// A wave-uniform value kept in a vector register: the kernel's uniform state does not fit the scalar file, and
// what only the vector ALU reads (masks, thresholds) need not compete for it (nor for the one scalar operand
// a VOP3 instruction may have).
__device__ __forceinline__ uint32_t in_vgpr(uint32_t s)
{
    uint32_t v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// s_waitcnt vmcnt(min(n, N)) with at most N + 1 cases (n is wave-uniform; waiting for fewer is always safe)
template <int N>
__device__ __forceinline__ void wait_vmcnt_upto(int n)
{
    if constexpr (N == 0) {
        wait_vmcnt_imm<0>();
    } else {
        if (n >= N) wait_vmcnt_imm<N>();
        else wait_vmcnt_upto<N - 1>(n);
    }
}

} // namespace

// DEPTH = blocks in flight ahead of the one being scanned
template <bool HAS_SEQ, int DEPTH>
__global__ void __launch_bounds__(64)
sk_scan_stream_kernel(const uint8_t *__restrict__ qual, const uint8_t *__restrict__ seq,
                      const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ lengths,
                      sk_cut_dev *__restrict__ out, unsigned long long *errword, sk_scan_args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x; // single-wave workgroups
    const int NB = (int)a.stream_nb;
    const uint32_t ring_bytes = (uint32_t)NB * 1024u;
    // P16 of the chunks behind the scan, indexed by chunk number modulo the table size (a power of two)
    uint32_t *table = reinterpret_cast<uint32_t *>(lds + ring_bytes);
    const uint32_t tmask4 = in_vgpr((a.stream_tbl - 1u) << 2);
    const uint64_t batch_end = a.n_reads ? rag_batch_end(offsets, lengths, a) : 0;
    const int maxw = (NB - DEPTH - 2) * 1024 - 16; // the widest window whose trailing bytes stay in the ring
    const uint32_t min4 = in_vgpr(splat((uint32_t)a.qmin)), max4 = in_vgpr(splat((uint32_t)a.qmax));
    const uint32_t hi4 = splat((uint32_t)(127 - a.qmax));
    const int range = a.qmax - a.qmin;
    const int B8 = (int)in_vgpr((uint32_t)(8 * range));
    const uint32_t clean = in_vgpr((uint32_t)(16 * range)); // the range-check sum of a chunk without a char out of range
    const uint32_t lane16 = 16u * (uint32_t)lane;

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
    auto streamed = [&](int L) { return L > 0 && L >= a.lthr && window_of(L) <= maxw; };

    // The spans.  All reads (a.buf_bytes == 0): ONE span per wave, the batch cut into gridDim.x runs of
    // consecutive reads of equal cost -- bytes plus a.stream_read_cost bytes per read, found by a search in `offsets`
    // (a deal of reads by number leaves the slowest of 3 584 waves 45 % over the mean on a 1-30 kb mix); equal
    // numbers of reads without offsets.  Or (a.buf_bytes != 0) the runs of 8 reads of the 64-read tiles
    // sk_scan_tile_any_kernel left -- if it left any: it has put this scan's number into the word after the
    // error word for every tile it skipped.  Runs rather than tiles so that the reads of one left-over tile
    // spread over the device; a wave asks the question for the tile its run lies in.
    // If the tile kernel left EVERY tile (word 6 of the error block: this scan's number << 32 | tiles left), the batch
    // is a long-read batch that came without a hint: spans of equal cost, as if it had been declared.
    // (Both words through the scalar cache -- written by the kernel before this one, and wave-uniform by construction:
    // a vector load would make everything that depends on them look divergent to the compiler.)
    bool leftovers = a.buf_bytes != 0;
    if (leftovers) {
        if (scalar_load(errword + 1) != a.scan_id) return;
        const uint64_t c = scalar_load(errword + 6);
        if ((uint32_t)(c >> 32) == (uint32_t)a.scan_id && (uint32_t)c == (uint32_t)((a.n_reads + 63) >> 6)) leftovers = false;
    }
    uint64_t span_lo = 0, span_hi = 0;
    if (!leftovers) {
        const uint64_t g = blockIdx.x, G = gridDim.x, n = a.n_reads;
        if (offsets) {
            // cost(r) = offsets[r] - offsets[0] + c r, r in [0, n]: ascending.  The span of wave g starts at the
            // first r with cost(r) >= g total / G and ends where the next one starts.  Both bounds at once: the
            // lower half of the wave searches one, the upper half the other, 32 probes per round.
            const uint64_t c = a.stream_read_cost, base = offsets[0];
            const uint64_t total = offsets[n] - base + c * n;
            const int half = lane >> 5, sub = lane & 31;
            const uint64_t tgt = (g + (uint64_t)half) * total / G; // total < 2^48, G < 2^16
            uint64_t l_ = 0, h_ = n; // the answer lies in [l_, h_] and cost(h_) >= tgt
            while (__builtin_amdgcn_ballot_w64(l_ < h_)) {
                const uint64_t step = (h_ - l_ + 31) / 32;
                const uint64_t p = min(l_ + (uint64_t)sub * max(step, (uint64_t)1), h_);
                const bool ge = offsets[p] - base + c * p >= tgt;
                const uint64_t m64 = __builtin_amdgcn_ballot_w64(ge);
                const uint32_t m = (uint32_t)(half ? m64 >> 32 : m64);
                if (l_ < h_) {
                    if (m == 0) {
                        l_ = min(l_ + 31 * step + 1, h_);
                    } else {
                        const uint64_t f = (uint64_t)__builtin_ctz(m);
                        if (f == 0) {
                            h_ = l_;
                        } else {
                            h_ = min(l_ + f * step, h_);
                            l_ = l_ + (f - 1) * step + 1;
                        }
                    }
                }
            }
            span_lo = readlane_u64(l_, 0);
            span_hi = g + 1 == G ? n : readlane_u64(l_, 32);
        } else {
            span_lo = n * g / G;
            span_hi = n * (g + 1) / G;
        }
    }
    const uint64_t n_units = leftovers ? (a.n_reads + 7) / 8 : (uint64_t)blockIdx.x + 1;

    uint32_t pbase = 0, cbase = 0; // ring byte offsets of the next block to load / the block being scanned
    for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        uint64_t lo = span_lo, hi = span_hi;
        if (leftovers) {
            lo = unit * 8;
            hi = min(a.n_reads, lo + 8);
            const sk_rag_tile pr_ = rag_probe(lo >> 6, lane, offsets, lengths, a);
            if (rag_tile_fits(pr_, a.buf_bytes)) continue;
        }
        // ---- the loader: a cursor over the blocks of the span, DEPTH ahead of the scan.  Per lane: where its
        // next chunk comes from and how many bytes of the read are left from there.
        // The read after the one being loaded is looked up one stream ahead (nstate: 0 not yet, 1 found, 2 none
        // left), so that the change of streams is a handful of moves wherever it happens.
        uint64_t pnext = lo, n_po = 0;
        int n_pL = 0, nstate = 0;
        int pact = 0;
        int pslow = 0; // the read's last chunk leaves the batch (careful loads), or its last block is empty
        int pk = 0, pleft = 0; // blocks left of the read's current stream (quality, then the sequence)
        uint64_t po = 0;
        int pL = 0;
        const uint8_t *psrc = nullptr;
        int prem = 0;
        int ahead = 0; // blocks issued and not yet scanned
        auto seek_next = [&]() { // the first read at or after pnext that goes through the ring
            nstate = 2;
            while (pnext < hi) {
                locate(pnext, n_po, n_pL);
                ++pnext;
                if (streamed(n_pL)) {
                    nstate = 1;
                    break;
                }
            }
        };
        auto next_stream = [&]() { // the current stream has been loaded to its end
            if (HAS_SEQ && pk == 0 && pact) {
                pk = 1;
                pleft = (pL + 1023) >> 10;
                psrc = seq + po + lane16;
                prem = pL - (int)lane16;
                return;
            }
            if (nstate == 0) seek_next();
            pact = 0;
            if (nstate == 1) {
                po = n_po;
                pL = n_pL;
                nstate = 0;
                pk = 0;
                pleft = (pL >> 10) + 1; // chunk index L >> 4 included: the window that ends with the read may end there
                // That block is empty if L is a multiple of 1024.  It still takes its turn with ONE DMA (the loads are
                // counted), so lane 0 is given one byte more to expect: it then fetches the 16 bytes behind the read,
                // which nobody looks at (every use of the block masks by L), and the read stays on the fast paths
                // like any other (round 2 sent such a read through the careful loader: 2x slower at 10 240 and
                // 30 720 bp).  Careful loads only where the last chunk -- or those 16 bytes -- would leave the batch.
                const int empty_last = (pL & 1023) == 0 ? 1 : 0;
                pslow = po + (((uint64_t)pL + 15u) & ~15ull) + (empty_last ? 16u : 0u) > batch_end ? 1 : 0;
                psrc = qual + po + lane16;
                prem = pL - (int)lane16 + ((empty_last && !pslow && lane == 0) ? 1 : 0);
                pact = 1;
            }
        };
        next_stream();
        auto issue = [&]() {
            if (nstate == 0) seek_next();
            if (!pact) return;
            uint8_t *dst = lds + pbase;
            if (!pslow) {
                if (prem > 0) __builtin_amdgcn_global_load_lds((gptr_t)psrc, (lptr_t)dst, 16, 0, SK_DMA_AUX);
            } else if (pk == 0 && pleft == 1 && (pL & 1023) == 0) { // nothing to load, but one DMA per block keeps the count
                if (lane == 0) __builtin_amdgcn_global_load_lds((gptr_t)(qual + po), (lptr_t)dst, 16, 0, SK_DMA_AUX);
            } else if (prem > 0) {
                if (prem >= 16 || (uint64_t)(psrc - (HAS_SEQ && pk ? seq : qual)) + 16u <= batch_end) {
                    __builtin_amdgcn_global_load_lds((gptr_t)psrc, (lptr_t)dst, 16, 0, SK_DMA_AUX);
                } else { // the batch ends inside this chunk: byte by byte
                    for (int q = 0; q < prem; ++q) dst[lane16 + (uint32_t)q] = psrc[q];
                }
            }
            ++ahead;
            pbase = pbase + 1024u == ring_bytes ? 0u : pbase + 1024u;
            psrc += 1024;
            prem -= 1024;
            if (--pleft == 0) next_stream();
        };
        // the scan's block has arrived when every load but those issued after it has returned
        auto arrive = [&]() {
            issue();
            wait_vmcnt_upto<DEPTH>(--ahead);
        };
        auto leave = [&]() {
            // every LDS read of this block is done before the loader comes round to its slot again
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cbase = cbase + 1024u == ring_bytes ? 0u : cbase + 1024u;
        };
#pragma unroll 1
        for (int d = 0; d < DEPTH; ++d) issue();

#pragma unroll 1
        for (uint64_t r = lo; r < hi; ++r) {
            uint64_t o;
            int L;
            locate(r, o, L);
            if (!(L > 0 && L >= a.lthr)) { // trim.cpp:21
                if (lane == 0) out[r] = sk_cut_dev{-1, -1};
                continue;
            }
            const int w = window_of(L);
            if (w > maxw) { // the window does not fit the ring: from global memory
                const sk_cut_dev cut = scan_read_global<HAS_SEQ>(qual + o, HAS_SEQ ? seq + o : nullptr, L, r, lane, a, errword);
                if (lane == 0) out[r] = cut;
                continue;
            }
            const int wq = w >> 4, wr = w & 15;
            const int nwin = L - w + 1;
            const int T = (int)in_vgpr((uint32_t)(a.craw * w));
            const uint32_t wq4 = in_vgpr(4u * (uint32_t)wq);
            const int nbq = (L >> 10) + 1;
            uint32_t pm[4]; // the first wr bytes of a chunk
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = wr - 4 * u;
                pm[u] = in_vgpr(n >= 4 ? ~0u : (n <= 0 ? 0u : (1u << (8 * n)) - 1u));
            }
            // blocks [jf0, jf1): every lane's aligned window exists, has one before it, and ends inside the read
            const int jf0 = (wq + 64) >> 6, jf1 = (L - 15) >> 10;
            const int jtail = (((nwin - 1) >> 4) + wq) >> 6; // the block the last aligned window ends in: not one of them
            const uint32_t nfast = (uint32_t)max(0, min(jf1, jtail) - jf0);
            int phase = a.no5 ? 1 : 0; // 0: looking for the first S >= T (trim.cpp:42), 1: for the first S < T after it (:61), 2: done
            int i0 = INF, i1 = INF, five = 0, three = L;
            uint32_t carry = 0; // sum of the read's bytes before the block
            int vprev = 0, vtail = 0;
            int pb = INF, pbch = 0; // first char out of range
            int j = 0;              // the block being scanned
            bool tail_seen = false; // the windows after the last aligned one were part of a cell already
            uint32_t k4 = 4u * (uint32_t)lane; // 4 * this lane's chunk number

            // byte at read position p, which lies in block j or up to NB - DEPTH - 1 blocks before it
            auto ringbyte = [&](int p) -> int {
                int at = (int)cbase - ((j - (p >> 10)) << 10);
                if (at < 0) at += (int)ring_bytes;
                return (int)lds[(uint32_t)at + (uint32_t)(p & 1023)];
            };
            // the first char at/above (below) the threshold from `from` on, inside the window that starts there
            // (it ends at or before the block being scanned)
            auto first_char = [&](int from, bool above) -> int {
                for (int g0 = 0; g0 < w; g0 += 64) {
                    const bool in = g0 + lane < w;
                    const int c = in ? ringbyte(from + g0 + lane) : 0;
                    const uint64_t m = __builtin_amdgcn_ballot_w64(in && ((c >= a.cthr_raw) == above));
                    if (m) return from + g0 + __builtin_ctzll(m);
                }
                return INF;
            };
            // One cell: the windows base + 1 .. base + cnt given Sb = S_base - T (cnt == 0: window 0 alone, Sb its
            // S - T), trim.cpp:34-81 over them in order
            auto cell = [&](int base, int Sb, int cnt) { // cnt <= 64: up to four cells in one go, a window per lane
                uint64_t ge, vm;
                if (cnt == 0) {
                    ge = Sb >= 0 ? 1u : 0u;
                    vm = 1u;
                } else {
                    int dl = 0;
                    if (lane < cnt) dl = ringbyte(base + w + lane) - ringbyte(base + lane); // trim.cpp:76-80
                    dl = (int)wave_scan_add((uint32_t)dl);
                    ge = __builtin_amdgcn_ballot_w64(Sb + dl >= 0); // bit u: S_{base + 1 + u} >= T
                    vm = cnt >= 64 ? ~0ull : (1ull << cnt) - 1ull;
                }
                uint64_t lt = ~ge & vm;
                if (phase == 0) {
                    lt = 0;
                    const uint64_t g = ge & vm;
                    if (g) {
                        const int u0 = __builtin_ctzll(g);
                        i0 = base + 1 + u0;
                        phase = 1;
                        lt = ~ge & vm & ~((2ull << u0) - 1ull);
                        five = first_char(i0, true); // trim.cpp:46-51
                        if (five == INF) five = 0;
                    }
                }
                if (phase == 1 && lt) {
                    i1 = base + 1 + __builtin_ctzll(lt);
                    phase = 2;
                    three = first_char(i1, false); // trim.cpp:65-70
                    if (three == INF) three = L;
                }
            };

            // The 16 windows that END in this lane's chunk kc of block jb (the block at cbase), i.e. 16 (at - 1) + 1 .. 16 at
            // with at = kc - wq >= 1, evaluated exactly: 4 windows per dword with the byte-parallel arithmetic of the tile
            // kernels.  The chars entering are the 16 bytes from offset wr of chunk kc - 1 on (five aligned dword reads,
            // v_alignbyte), the chars leaving are chunk at - 1, S - T starts from the aligned window before (vpl).
            // -> bit (16 - u): window 16 (at - 1) + u is below the threshold, u = 1..16.  ~60 vector instructions for the
            // block's 1024 windows, whatever the data: what a block costs when the averages HOVER at the threshold
            // (cell by cell -- `cell` above, 64 windows per call -- round 2 ran such reads at 0.95 TB/s).
            auto dense16 = [&](int jb, int kc, int vpl) -> uint32_t {
                auto ringaddr = [&](int pos) -> uint32_t { // of the dword that holds read position pos (in block jb or before)
                    int a_ = (int)cbase + (pos - (jb << 10));
                    if (a_ < 0) a_ += (int)ring_bytes;
                    return (uint32_t)a_ & ~3u;
                };
                const int pin = 16 * (kc - 1) + wr; // the first char entering: window 16 (at - 1) + 1 ends with it
                uint32_t e[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) e[u] = *reinterpret_cast<const uint32_t *>(lds + ringaddr((pin & ~3) + 4 * u));
                const sk_v4u yv = *reinterpret_cast<const sk_v4u *>(lds + ringaddr(16 * (kc - wq - 1)));
                int vv = vpl;
                uint32_t M = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t xin = __builtin_amdgcn_alignbyte(e[u + 1], e[u], (uint32_t)(wr & 3));
                    const int dd = (int)(((xin | H4) - yv[u]) ^ H4); // per byte: entering - leaving as int8 (both < 128)
                    const int t1 = __builtin_amdgcn_sdot4(dd, 0x00000001, vv, false);
                    const int t2 = __builtin_amdgcn_sdot4(dd, 0x00000101, vv, false);
                    const int t3 = __builtin_amdgcn_sdot4(dd, 0x00010101, vv, false);
                    const int t4 = __builtin_amdgcn_sdot4(dd, 0x01010101, vv, false);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t1, 31);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t2, 31);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t3, 31);
                    M = __builtin_amdgcn_alignbit(M, (uint32_t)t4, 31);
                    vv = t4;
                }
                return M & 0xffffu;
            };

            bool have_block = false; // the block to scan has arrived already (the lockstep run below stopped at it)
            // the block the first window ends in (lane r0 of it), if it is a whole block inside the read
            const int j0 = wq >> 6, r0 = wq & 63;
            const bool j0_plain = j0 < min(jf1, jtail);
#pragma unroll 1
            for (j = 0; j < nbq; ++j, k4 += 256u) {
                if (j == j0 && j0_plain && phase == (a.no5 ? 1 : 0) && pact && !pslow && ahead == DEPTH) {
                    // ---- the first window's block, when nothing happens in it but the usual: window 0 is at/above the
                    // threshold (trim.cpp:42: the 5' cut is found at once) and every later window of the block too.
                    // One turn of the lockstep run below with three kinds of lanes: before the first window's end
                    // (no window ends there), at it (window 0 alone), after it (a cell of 16 as everywhere).
                    if (prem > 0) __builtin_amdgcn_global_load_lds((gptr_t)psrc, (lptr_t)(lds + pbase), 16, 0, SK_DMA_AUX);
                    psrc += 1024;
                    prem -= 1024;
                    pbase = pbase + 1024u == ring_bytes ? 0u : pbase + 1024u;
                    if (--pleft == 0) next_stream();
                    wait_vmcnt_imm<DEPTH>();
                    const sk_v4u d = *reinterpret_cast<const sk_v4u *>(lds + cbase + lane16);
                    uint32_t sad = 0, sum = 0, part = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        sad = __builtin_amdgcn_sad_u8(d[u], min4, sad);
                        sad = __builtin_amdgcn_sad_u8(d[u], max4, sad);
                        sum = __builtin_amdgcn_sad_u8(d[u], 0u, sum);
                        part = __builtin_amdgcn_sad_u8(d[u] & pm[u], 0u, part);
                    }
                    const uint32_t incl = wave_scan_add(sum);
                    const uint32_t P = carry + incl - sum;
                    *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(table) + (k4 & tmask4)) = P;
                    const uint32_t Pa = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(table) + ((k4 - wq4) & tmask4));
                    const int v = (int)(P + part - Pa) - T; // (meaningless in the lanes before r0)
                    const int vp = __builtin_amdgcn_update_dpp(vprev, v, 0x138, 0xf, 0xf, false); // wave_shr:1
                    const int ai = lane - r0;
                    const bool quiet = sad == clean && (ai < 0 || (ai == 0 ? v >= 0 : min(v, vp) >= B8));
                    if (__builtin_amdgcn_ballot_w64(!quiet)) {
                        ahead = DEPTH + 1; // loaded, arrived, not scanned: the general turn below takes it
                        have_block = true;
                    } else {
                        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        vprev = __builtin_amdgcn_readlane(v, 63);
                        if (phase == 0) {
                            i0 = 0;
                            phase = 1;
                            five = first_char(0, true); // trim.cpp:46-51
                            if (five == INF) five = 0;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        cbase = cbase + 1024u == ring_bytes ? 0u : cbase + 1024u;
                        continue;
                    }
                }
                if (phase == 1 && pact && !pslow && ahead == DEPTH && (uint32_t)(j - jf0) < nfast) {
                    // ---- lockstep run: the scan is in the usual blocks of its read (see below) and the loader in the
                    // middle of a stream, DEPTH blocks ahead.  One load, one counted wait, one block scanned per
                    // turn, nothing else: what the general turn below decides per block is decided once here.
                    // The scalar unit is the bottleneck of this kernel (one per CU, shared by all its waves).
                    const int nrun = (int)(jf0 + nfast) - j;
                    int done = 0;
                    bool ev = false, stop = false;
                    while (done < nrun && !stop) {
                        if (prem > 0) __builtin_amdgcn_global_load_lds((gptr_t)psrc, (lptr_t)(lds + pbase), 16, 0, SK_DMA_AUX);
                        psrc += 1024;
                        prem -= 1024;
                        pbase = pbase + 1024u == ring_bytes ? 0u : pbase + 1024u;
                        if (--pleft == 0) { // the loader goes on with the next stream, the scan stays in its read
                            next_stream();
                            stop = !pact || pslow;
                        }
                        wait_vmcnt_imm<DEPTH>();
                        const sk_v4u d = *reinterpret_cast<const sk_v4u *>(lds + cbase + lane16);
                        uint32_t sad = 0, sum = 0, part = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sad = __builtin_amdgcn_sad_u8(d[u], min4, sad);
                            sad = __builtin_amdgcn_sad_u8(d[u], max4, sad);
                            sum = __builtin_amdgcn_sad_u8(d[u], 0u, sum);
                            part = __builtin_amdgcn_sad_u8(d[u] & pm[u], 0u, part);
                        }
                        const uint32_t incl = wave_scan_add(sum);
                        const uint32_t P = carry + incl - sum;
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(table) + (k4 & tmask4)) = P;
                        const uint32_t Pa = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(table) + ((k4 - wq4) & tmask4));
                        const int v = (int)(P + part - Pa) - T;
                        const int vp = __builtin_amdgcn_update_dpp(vprev, v, 0x138, 0xf, 0xf, false); // wave_shr:1
                        const bool quiet = min(v, vp) >= B8 && sad == clean;
                        if (__builtin_amdgcn_ballot_w64(!quiet)) {
                            // not bounded away from the threshold.  If every char is in range the block is evaluated exactly
                            // here (every lane of a block of the run has its cell), and the run goes on unless a window
                            // below the threshold shows: a read that hovers ABOVE the threshold stays in the run
                            bool stay = false;
                            if (__builtin_amdgcn_ballot_w64(sad != clean) == 0)
                                stay = __builtin_amdgcn_ballot_w64(dense16(j + done, (int)(k4 >> 2), vp) != 0) == 0;
                            if (!stay) {
                                ev = true;
                                break;
                            }
                        }
                        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        vprev = __builtin_amdgcn_readlane(v, 63);
                        cbase = cbase + 1024u == ring_bytes ? 0u : cbase + 1024u;
                        k4 += 256u;
                        ++done;
                    }
                    j += done;
                    if (ev) {
                        ahead = DEPTH + 1;
                        have_block = true;
                    }
                }
                if (!have_block && j == nbq - 1 && j == jtail && j >= jf0 && phase == 1 && pact && !pslow && ahead == DEPTH) {
                    // ---- the read's last block, when nothing happens in it either: every window that ends in it at/above
                    // the threshold, the windows after the last aligned one too, no char out of range -- the read
                    // keeps its 3' end (trim.cpp:13).  The lockstep turn with the read's end masked.
                    if (prem > 0) __builtin_amdgcn_global_load_lds((gptr_t)psrc, (lptr_t)(lds + pbase), 16, 0, SK_DMA_AUX);
                    psrc += 1024;
                    prem -= 1024;
                    pbase = pbase + 1024u == ring_bytes ? 0u : pbase + 1024u;
                    if (--pleft == 0) next_stream();
                    wait_vmcnt_imm<DEPTH>();
                    const sk_v4u d = *reinterpret_cast<const sk_v4u *>(lds + cbase + lane16);
                    const int x = 16 * (64 * j + lane);
                    uint32_t sad = 0, sum = 0, part = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int n = L - x - 4 * u;
                        const uint32_t xq = first_bytes(d[u], n, min4);
                        sad = __builtin_amdgcn_sad_u8(xq, min4, sad);
                        sad = __builtin_amdgcn_sad_u8(xq, max4, sad);
                        sum = __builtin_amdgcn_sad_u8(first_bytes(d[u], n, 0u), 0u, sum);
                        part = __builtin_amdgcn_sad_u8(d[u] & pm[u], 0u, part);
                    }
                    const uint32_t incl = wave_scan_add(sum);
                    const uint32_t P = carry + incl - sum;
                    *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(table) + (k4 & tmask4)) = P;
                    const uint32_t Pa = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(table) + ((k4 - wq4) & tmask4));
                    const int v = (int)(P + part - Pa) - T;
                    const int vp = __builtin_amdgcn_update_dpp(vprev, v, 0x138, 0xf, 0xf, false); // wave_shr:1
                    const bool val = x + wr <= L; // the aligned window that ends in this chunk exists
                    const bool quiet = sad == clean && (!val || min(v, vp) >= B8);
                    const int alast = (nwin - 1) >> 4, rem = nwin - 1 - 16 * alast;
                    const int vt = __builtin_amdgcn_readlane(v, (alast + wq) & 63); // the last aligned window
                    if (__builtin_amdgcn_ballot_w64(!quiet) || vt < rem * range) {
                        ahead = DEPTH + 1; // loaded, arrived, not scanned: the general turn below takes it
                        have_block = true;
                    } else {
                        tail_seen = true;
                        break; // three stays L
                    }
                }
                if (have_block) {
                    --ahead;
                    have_block = false;
                } else {
                    arrive();
                }
                if (phase < 2) {
                    const sk_v4u d = *reinterpret_cast<const sk_v4u *>(lds + cbase + lane16);
                    bool event = true;
                    const bool full = 1024 * (j + 1) <= L; // no lane's chunk reaches past the read
                    if (full && 64 * (j + 1) <= wq) {
                        // ---- a block before the end of the first window: range check, sums, prefix table
                        uint32_t sad = 0, sum = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sad = __builtin_amdgcn_sad_u8(d[u], min4, sad);
                            sad = __builtin_amdgcn_sad_u8(d[u], max4, sad);
                            sum = __builtin_amdgcn_sad_u8(d[u], 0u, sum);
                        }
                        if (__builtin_amdgcn_ballot_w64(sad != clean) == 0) {
                            const uint32_t incl = wave_scan_add(sum);
                            *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(table) + (k4 & tmask4)) = carry + incl - sum;
                            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            event = false;
                        }
                    } else if (phase == 1 && (uint32_t)(j - jf0) < nfast) {
                        // ---- the usual block: inside the read, past the first window, looking for the first S < T.
                        // Nothing happens in it if every aligned window and its predecessor are 8 (qmax - qmin)
                        // or more above the threshold and every char is in range.
                        uint32_t sad = 0, sum = 0, part = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sad = __builtin_amdgcn_sad_u8(d[u], min4, sad);
                            sad = __builtin_amdgcn_sad_u8(d[u], max4, sad);
                            sum = __builtin_amdgcn_sad_u8(d[u], 0u, sum);
                            part = __builtin_amdgcn_sad_u8(d[u] & pm[u], 0u, part);
                        }
                        const uint32_t incl = wave_scan_add(sum);
                        const uint32_t P = carry + incl - sum; // the bytes before this chunk
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(table) + (k4 & tmask4)) = P;
                        const uint32_t Pa = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(table) + ((k4 - wq4) & tmask4));
                        const int v = (int)(P + part - Pa) - T;
                        const int vp = __builtin_amdgcn_update_dpp(vprev, v, 0x138, 0xf, 0xf, false); // wave_shr:1
                        const bool quiet = min(v, vp) >= B8 && sad == clean;
                        if (__builtin_amdgcn_ballot_w64(!quiet) == 0) {
                            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            vprev = __builtin_amdgcn_readlane(v, 63);
                            event = false;
                        }
                    }
                    if (event) {
                        const int k = 64 * j + lane, x = 16 * k;
                        uint32_t sad = 0, sum = 0;
                        if (full) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                sad = __builtin_amdgcn_sad_u8(d[u], min4, sad);
                                sad = __builtin_amdgcn_sad_u8(d[u], max4, sad);
                                sum = __builtin_amdgcn_sad_u8(d[u], 0u, sum);
                            }
                        } else { // the read ends in this block
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int n = L - x - 4 * u;
                                const uint32_t xq = first_bytes(d[u], n, min4);
                                sad = __builtin_amdgcn_sad_u8(xq, min4, sad);
                                sad = __builtin_amdgcn_sad_u8(xq, max4, sad);
                                sum = __builtin_amdgcn_sad_u8(first_bytes(d[u], n, 0u), 0u, sum);
                            }
                        }
                        const bool bad = sad != clean; // fillers are legal chars
                        const uint32_t incl = wave_scan_add(sum);
                        const uint32_t P = carry + incl - sum; // the bytes before x
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(table) + (k4 & tmask4)) = P;
                        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);

                        // the aligned window that ends in this chunk
                        const int ai = k - wq;
                        const bool val = ai >= 0 && x + wr <= L;
                        uint32_t part = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) part = __builtin_amdgcn_sad_u8(d[u] & pm[u], 0u, part);
                        int v = 0;
                        if (val) v = (int)(P + part - *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(table) + ((k4 - wq4) & tmask4))) - T;
                        // v of the aligned window before: the lane to the left, lane 0 from the block before (wave_shr:1)
                        const int vp = __builtin_amdgcn_update_dpp(vprev, v, 0x138, 0xf, 0xf, false);
                        const bool first = ai == 0;
                        const bool cge = first ? v >= 0 : (v >= B8 && vp >= B8); // every window of the cell at/above the threshold
                        const bool clt = first ? v < 0 : (v < -B8 && vp < -B8); // every window of the cell below it
                        const uint64_t m_ge = __builtin_amdgcn_ballot_w64(val && !clt);
                        const uint64_t m_lt = __builtin_amdgcn_ballot_w64(val && !cge);
                        vprev = __builtin_amdgcn_readlane(v, 63);
                        if (j == jtail) vtail = __builtin_amdgcn_readlane(v, (((nwin - 1) >> 4) + wq) & 63);
                        if (__builtin_amdgcn_ballot_w64(bad) && pb == INF) { // trim.cpp:129: where, and which char
                            int p = INF;
                            if (bad) {
#pragma unroll
                                for (int u = 3; u >= 0; --u) {
                                    const uint32_t f = keep_first(bad_flags(d[u], min4, hi4), L - x - 4 * u);
                                    if (f) p = x + 4 * u + (__builtin_ctz(f) >> 3);
                                }
                            }
                            pb = wave_min(p);
                            pbch = (int)(int8_t)ringbyte(pb);
                        }
                        int cur = 0;
                        if (phase < 2 && __builtin_popcountll(phase == 0 ? m_ge : m_lt) > 2) {
                            // ---- many cells flagged: the averages HOVER at the threshold.  Every lane evaluates ITS cell exactly
                            // (dense16); window order = lane order, so the read's state steps by ballots.
                            const int at = k - wq; // this lane's aligned window: 16 at (val: it exists)
                            uint32_t lt16 = 0, ge16 = 0; // bit (31 - (u - 1)): window 16 (at - 1) + u is below / at-or-above T, u = 1..16
                            if (val && at >= 1) {
                                lt16 = dense16(j, k, vp) << 16;
                                ge16 = ~lt16 & 0xffff0000u;
                            } else if (val && at == 0) { // window 0 alone: as u = 16 of the cell "before the read"
                                lt16 = v < 0 ? 0x00010000u : 0u;
                                ge16 = v < 0 ? 0u : 0x00010000u;
                            }
                            int after_from = phase == 1 ? -1 : INF; // windows strictly after this one may be the first S < T
                            if (phase == 0) { // trim.cpp:42
                                const uint64_t m = __builtin_amdgcn_ballot_w64(ge16 != 0);
                                if (m) {
                                    const int t = __builtin_ctzll(m);
                                    i0 = 16 * (64 * j + t - wq - 1) + 1 + (int)__builtin_clz((uint32_t)__builtin_amdgcn_readlane((int)ge16, t));
                                    phase = 1;
                                    after_from = i0;
                                    five = first_char(i0, true); // trim.cpp:46-51
                                    if (five == INF) five = 0;
                                }
                            }
                            if (phase == 1) { // trim.cpp:61
                                const int rel = after_from - 16 * (at - 1); // this lane's windows u <= rel are not after it
                                const uint32_t aft = rel <= 0 ? ~0u : (rel >= 32 ? 0u : ~0u >> rel);
                                const uint32_t cand = lt16 & aft;
                                const uint64_t m = __builtin_amdgcn_ballot_w64(cand != 0);
                                if (m) {
                                    const int t = __builtin_ctzll(m);
                                    i1 = 16 * (64 * j + t - wq - 1) + 1 + (int)__builtin_clz((uint32_t)__builtin_amdgcn_readlane((int)cand, t));
                                    phase = 2;
                                    three = first_char(i1, false); // trim.cpp:65-70
                                    if (three == INF) three = L;
                                }
                            }
                            cur = 64;
                        }
                        while (phase < 2 && cur < 64) {
                            const uint64_t m = (phase == 0 ? m_ge : m_lt) & (~0ull << cur);
                            if (!m) break;
                            const int t = __builtin_ctzll(m);
                            const int at = 64 * j + t - wq;
                            if (at == 0) {
                                cell(-1, __builtin_amdgcn_readlane(v, t), 0);
                                cur = t + 1;
                            } else {
                                // this cell and the next three: a crossing takes a few cells, and the lanes are there.
                                // As far as the windows go, and their chars lie in this block.
                                const int base = 16 * (at - 1);
                                const int cnt = min(min(64, 16 * (64 - t)), nwin - 1 - base);
                                cell(base, __builtin_amdgcn_readlane(vp, t), cnt);
                                cur = t + ((cnt + 15) >> 4);
                                // (it ends on a multiple of 16, or with the read's last window: then the windows
                                // after the last aligned one have had their turn)
                                tail_seen = tail_seen || base + cnt == nwin - 1;
                            }
                        }
                    }
                }
                if (j + 1 < nbq) leave();
            }
            j = nbq - 1;
            if (phase < 2 && !tail_seen) { // the windows after the last aligned one
                const int alast = (nwin - 1) >> 4, rem = nwin - 1 - 16 * alast;
                if (rem > 0) {
                    const bool none = phase == 0 ? vtail < -rem * range : vtail >= rem * range;
                    if (!none) cell(16 * alast, vtail, rem);
                }
            }
            const int touched = phase == 2 ? i1 + w : L;
            if (pb < touched && lane == 0) report_error(errword, r, pb, pbch);
            leave();

            if (HAS_SEQ) { // trim.cpp:86-98
                uint32_t nlo = NONE, anyN = 0; // bit index of the first lowercase n; any uppercase N
                const int nbs = (L + 1023) >> 10;
#pragma unroll 1
                for (j = 0; j < nbs; ++j) {
                    arrive();
                    const sk_v4u d = *reinterpret_cast<const sk_v4u *>(lds + cbase + lane16);
                    const int x = 16 * (64 * j + lane);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t xx = first_bytes(d[u], L - x - 4 * u, 0u);
                        const uint32_t y = (xx | 0x20202020u) ^ 0x6e6e6e6eu;
                        const uint32_t either = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & H4;
                        const uint32_t lower = either & (xx << 2); // bit 5 of the byte moved onto its flag
                        nlo = min(nlo, __builtin_elementwise_add_sat(ffbl_or_none(lower), (uint32_t)(8 * (x + 4 * u))));
                        anyN |= either ^ lower;
                    }
                    leave();
                }
                const int nl = wave_min(nlo == NONE ? INF : (int)(nlo >> 3));
                anyN = wave_or(anyN);
                if (nl != INF) three = nl - 1;
                else if (anyN) three = -2;
            }
            const bool found5 = a.no5 || i0 != INF;
            if (!found5 || (three - five < a.lthr)) { // trim.cpp:103-108
                five = -1;
                three = -1;
            }
            if (lane == 0) out[r] = sk_cut_dev{five, three};
        }
    }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_stream(const uint8_t *qual, const uint8_t *seq, const uint64_t *offsets,
                                       const uint32_t *lengths, sk_cut_dev *out, unsigned long long *errword,
                                       const sk_scan_args *a, uint64_t max_len, int cu_count, hipStream_t stream)
{
    // max_len = the longest read the caller expects (0 = unknown): sizes the ring; longer reads still come out
    // right, from global memory (scan_read_global)
    if (a->n_reads == 0) return hipSuccess;
    if (max_len == 0) max_len = 32768;
    const uint64_t wmax = max_len >= 10 ? max_len / 10 : max_len;
    uint64_t back = 2 + (wmax + 16 + 1023) / 1024; // blocks behind the loader: the one being scanned and the window's trailing bytes
    if (back > 60) back = 60;
    static const int depth_env = [] { const char *e = getenv("SK_STREAM_DEPTH"); return e ? atoi(e) : 0; }();
    int depth = depth_env ? depth_env : 3;
    if (depth != 2 && depth != 3 && depth != 4 && depth != 8) depth = 3;
    sk_scan_args at = *a;
    at.stream_nb = (uint32_t)(back + depth);
    uint32_t tbl = 64;
    while (tbl < 64u * (uint32_t)(back + 1)) tbl <<= 1;
    at.stream_tbl = tbl;
    const uint32_t lds_bytes = at.stream_nb * 1024u + tbl * 4u;
    int per_cu = (int)(SK_LDS_PER_CU / lds_bytes);
    static const int wave_cap = [] { const char *e = getenv("SK_STREAM_WAVES"); return e ? atoi(e) : 24; }();
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu > 4) per_cu &= ~3; // the same number of waves on every SIMD (17 per CU measured slower than 16)
    if (per_cu < 1) return hipErrorInvalidValue;
    uint64_t grid = (uint64_t)cu_count * per_cu;
    // one span of consecutive reads per wave, cut at equal cost: a read counts as its bytes plus this many
    // (what a read costs beyond its blocks: the first window, the cut searches, the store)
    static const uint32_t read_cost = [] { const char *e = getenv("SK_STREAM_READ_COST"); return e ? (uint32_t)atoll(e) : 4096u; }();
    at.stream_read_cost = read_cost;
    const uint64_t n_units = a->buf_bytes ? (a->n_reads + 7) / 8 : a->n_reads;
    if (grid > n_units) grid = n_units;
    if (grid == 0) return hipSuccess;
    auto launch = [&](auto kern) {
        const kernel_facts facts = prepare_kernel(kern);
        if (facts.status != hipSuccess) return facts.status;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, qual, seq, offsets, lengths, out,
                           errword, at);
        return hipGetLastError();
    };
    auto pick = [&](auto seqtag) {
        constexpr bool S = decltype(seqtag)::value;
        switch (depth) {
        case 2: return launch(sk_scan_stream_kernel<S, 2>);
        case 8: return launch(sk_scan_stream_kernel<S, 8>);
        case 4: return launch(sk_scan_stream_kernel<S, 4>);
        default: return launch(sk_scan_stream_kernel<S, 3>);
        }
    };
    return a->truncn ? pick(std::true_type{}) : pick(std::false_type{});
}
