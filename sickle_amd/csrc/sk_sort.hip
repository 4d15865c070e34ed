// sk_sort.hip -- ragged batches of mixed lengths: the reads of a batch regrouped on the device, window by window,
// into tiles of equal WINDOW WIDTH, so that the lane-per-read tile kernel can take them on its matrix path.
//
// A ragged batch (offsets) hands the tile kernel 64 CONSECUTIVE reads per tile.  When those differ in length the tile
// walks the vector-ALU path with every lane running to the longest read's window count -- 4.6x the vector
// instructions of the same reads grouped by length, 0.20 of the HBM peak on a 75-301 bp mix (profiles/r02).  What the
// matrix path needs is one window width w per tile (the band matrix), not one length: reads of 10 w .. 10 w + 9 bases
// share it.  So the batch is cut into windows of SK_SORT_WINDOW consecutive reads, and one workgroup per window
// counting-sorts the window's reads by w (52 classes: histogram, prefix, scatter -- in LDS; no quality byte moves):
//   tile lists: per class ceil(count / 64) tiles, 32 bytes each: {window, rows, w; where the window starts in the batch; its bytes;
//               the longest and the shortest read of the tile}
//   perm: 64 entries per tile, in list order: {offset of the read inside its window, its length, its number in the window}
// (a tile's entries lie where the scan can compute: it loads descriptor and entries side by side, a tile ahead)
// The scan (sk_scan_tile_body, SORT) then gathers a tile's rows by the re-striding loader -- a row start per lane.
// Why windows, and why the tile lists are kept per XCD: two neighbours in the batch share a 128-byte line, and with
// the whole batch sorted they would be fetched from HBM by tiles that run far apart in time and place -- 1.3-1.7x the
// bytes.  A window (8192 reads, ~1.5 MB) is what the waves of ONE XCD work on at a time: the second fetch of a line
// then comes from that XCD's L2.  Window k's tiles go to list k mod 8; the scan gives list x to the workgroups with
// blockIdx mod 8 == x (workgroups are dealt round-robin over the XCDs; if a runtime ever deals them otherwise only
// the locality is lost).
//
// Whether the sorted scan runs at all is decided on the device, and the scans read the verdict there (all of them are
// enqueued, the ones it goes against return at once): counts[8] != 0: some window's reads differ in length, from a
// look at the first eighth of every window (sk_sort_sample_kernel: a batch of one length must not pay for a regrouping it does not
// need -- 8 bytes per read written and read again; a mix that only shows in the other windows keeps the plain tile
// kernel, which is slower, not wrong); counts[9] = reads longer than the tile buffers take, counted exactly by the
// sort itself.  Uniform batches and batches with long reads stay with the kernels they had.
#include "sk_kernel_common.h"

// the first eighth of every window: do its reads differ in length?
__global__ void __launch_bounds__(256)
sk_sort_sample_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint32_t *__restrict__ counts, uint32_t *__restrict__ counts_of_next_scan)
{
    if (blockIdx.x == 0 && threadIdx.x < 16) counts_of_next_scan[threadIdx.x] = 0; // (this scan's were cleared by the scan before)
    const uint64_t r0 = (uint64_t)blockIdx.x * SK_SORT_WINDOW;
    const uint32_t m = (uint32_t)min((uint64_t)(SK_SORT_WINDOW / 8), n_reads - r0);
    const uint64_t first = offsets[r0 + 1] - offsets[r0];
    bool differs = false;
    for (uint32_t k = threadIdx.x; k < m; k += 256u) differs |= offsets[r0 + k + 1] - offsets[r0 + k] != first;
    // (a flag, not a count: one plain store per workgroup -- an atomic per wave, ~2 000 on one address, took 20 of this
    // kernel's 25 us)
    if (__syncthreads_or(differs) && threadIdx.x == 0) counts[8] = 1u;
}

// One workgroup per window.  Thread t takes the window's reads 8t .. 8t+7 (a wave: 512 consecutive reads), and the
// counting is done PER WAVE: 16 x 64 counters in LDS, so that the LDS atomics of a wave only meet the atomics of its own
// 64 lanes (one counter set per workgroup put all 8192 reads of a window on ~23 addresses, twice, one after the other:
// the kernel spent 15 of its 21 us there).  A class's reads are then ranked wave by wave -- a tile's 64 reads come from
// two or three neighbouring waves' stretches of the window, ~200 KB instead of the window's 1.5 MB.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
sk_sort_windows_kernel(const uint64_t *__restrict__ offsets, uint64_t n_reads, uint32_t max_len, uint64_t *__restrict__ perm,
                       unsigned long long *__restrict__ lists, uint32_t list_cap, uint32_t *__restrict__ counts /* [8] tiles per list, [8] flags[0], [9] flags[1] */)
{
    constexpr int W = SK_SORT_WINDOW, PER = W / THREADS, NC = 64, NW = THREADS / 64, NT = W / 64 + NC; // NT: more tiles than a window can have
    __shared__ uint32_t cnt_w[NW][NC]; // per wave and class: count, then the rank its first read of the class gets
    __shared__ uint32_t hist[NC], tbase[NC], gbase, tmax[NT], tmin[NT];
    __shared__ uint32_t rel[W + 1]; // the window's offsets relative to its first (loaded lane by lane, read 9 per thread)
    const int t = threadIdx.x, wave = t >> 6;
    const uint64_t widx = blockIdx.x, r0 = widx * W;
    const uint64_t wstart = offsets[r0]; // (issued with the verdict's load, not behind it)
    if (counts[8] == 0) return; // a batch of one length (as far as the sample saw): nothing to regroup
    const uint32_t m = (uint32_t)min((uint64_t)W, n_reads - r0);
    for (int i = t; i < NW * NC; i += THREADS) (&cnt_w[0][0])[i] = 0;
    if (t < NT) tmax[t] = 0, tmin[t] = 0xffffu;
    __syncthreads();
    uint32_t nlong = 0;
#pragma unroll
    for (int i = 0; i <= PER; ++i) {
        const uint32_t k = (uint32_t)t + (uint32_t)i * THREADS;
        if (k <= m && (i < PER || t == 0)) {
            const uint64_t o = offsets[r0 + k];
            if (o < wstart || o - wstart > 0xffffffffull) ++nlong; // a window beyond 4 GiB, or offsets that do not ascend: not for the tiles
            rel[k] = (uint32_t)(o - wstart);
        }
    }
    __syncthreads();
    uint32_t ro[PER];
    uint16_t ln[PER];
    uint8_t cl[PER];
    const uint32_t k0 = (uint32_t)t * PER;
    uint32_t o = k0 < m ? rel[k0] : 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t k = k0 + (uint32_t)i;
        cl[i] = 255;
        if (k < m) {
            const uint32_t e = rel[k + 1];
            const uint32_t L64 = e - o; // (not ascending: wraps to a length no tile takes)
            const uint32_t L = min(L64, 0xffffu);
            uint32_t c;
            if (L64 > max_len) {
                c = 63;
                ++nlong;
            } else {
                const uint32_t w = L / 10 ? L / 10 : L; // reference src/trim.cpp:8, :30
                c = w; // 0 (an empty read) .. 50
            }
            ro[i] = o;
            ln[i] = (uint16_t)L;
            cl[i] = (uint8_t)c;
            atomicAdd(&cnt_w[wave][c], 1u);
            o = e;
        }
    }
    __syncthreads();
    // per class: the waves' counts -> where each wave's reads of the class begin; then exclusive prefixes over the 64
    // classes: tiles (one wave)
    uint32_t g_mine = 0;
    if (t < NC) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t c = cnt_w[w][t];
            cnt_w[w][t] = run;
            run += c;
        }
        hist[t] = run;
        const uint32_t nt = (run + 63u) >> 6;
        uint32_t q = nt;
#pragma unroll
        for (int d = 1; d < NC; d <<= 1) {
            const uint32_t uq = __shfl_up(q, d, 64);
            if (t >= d) q += uq;
        }
        tbase[t] = q - nt;
        // the window's tiles get their places in the list of XCD widx mod 8 -- the atomic's round trip runs under the ranking below
        if (t == NC - 1) g_mine = atomicAdd(&counts[widx & 7u], q);
    }
    if (nlong) atomicAdd(&counts[9], nlong);
    __syncthreads();
    uint32_t rank[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        rank[i] = 0;
        if (cl[i] != 255) {
            rank[i] = atomicAdd(&cnt_w[wave][cl[i]], 1u); // the read's place in its class: tile rank / 64, lane rank % 64
            const uint32_t ltile = tbase[cl[i]] + (rank[i] >> 6);
            atomicMax(&tmax[ltile], (uint32_t)ln[i]); // the longest and the shortest read of the tile: the scan sizes its image
            atomicMin(&tmin[ltile], (uint32_t)ln[i]); // and its unmasked loops by them, without a reduction over the lanes
        }
    }
    if (t == NC - 1) gbase = g_mine;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t k = k0 + (uint32_t)i;
        if (cl[i] != 255) {
            const uint32_t tile = gbase + tbase[cl[i]] + (rank[i] >> 6);
            if (tile < list_cap)
                perm[((size_t)(widx & 7u) * list_cap + tile) * 64u + (rank[i] & 63u)] = (uint64_t)ro[i] | ((uint64_t)ln[i] << 32) | ((uint64_t)k << 48);
        }
    }
    if (t < NC) {
        const uint32_t cnt = hist[t], nt = (cnt + 63u) >> 6;
        unsigned long long *dst = lists + ((size_t)(widx & 7u) * list_cap + gbase + tbase[t]) * 4u;
        const unsigned long long span = rel[m]; // (a window beyond 4 GiB is flagged above: the batch then goes to the other kernels)
        for (uint32_t j = 0; j < nt; ++j) {
            const uint32_t rows = min(64u, cnt - 64u * j);
            if (gbase + tbase[t] + j < list_cap) {
                dst[4 * j] = (unsigned long long)widx | ((unsigned long long)rows << 48) | ((unsigned long long)t << 56);
                dst[4 * j + 1] = wstart;
                dst[4 * j + 2] = span;
                dst[4 * j + 3] = (unsigned long long)tmax[tbase[t] + j] | ((unsigned long long)tmin[tbase[t] + j] << 16);
            }
        }
    }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_sort(const uint64_t *offsets, uint64_t n_reads, uint32_t max_len, uint64_t *perm,
                                     unsigned long long *lists, uint32_t list_cap, uint32_t *counts, uint32_t *counts_of_next_scan,
                                     hipStream_t stream)
{
    const uint64_t windows = (n_reads + SK_SORT_WINDOW - 1) / SK_SORT_WINDOW;
    hipLaunchKernelGGL(sk_sort_sample_kernel, dim3((unsigned)windows), dim3(256), 0, stream, offsets, n_reads, counts, counts_of_next_scan);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sk_sort_windows_kernel<1024>, dim3((unsigned)windows), dim3(1024), 0, stream, offsets, n_reads, max_len, perm, lists,
                       list_cap, counts);
    return hipGetLastError();
}
