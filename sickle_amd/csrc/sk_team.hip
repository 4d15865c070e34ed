// sk_team.hip -- the general kernel (a team of lanes per read) and its launcher; see the block comment below.
#include "sk_kernel_common.h"

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
