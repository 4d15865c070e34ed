// sk_kernel_common.h -- what the kernel translation units share: byte-parallel helpers, DPP reductions, the
// error word, the LDS-DMA tile copy, counted waits, the description of a ragged tile, and the once-per-
// (device, kernel) launch preparation.  Everything is in an anonymous namespace: each unit gets its own copy.
#ifndef SK_KERNEL_COMMON_H
#define SK_KERNEL_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>
#include <algorithm>

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

typedef int sk_v4i __attribute__((ext_vector_type(4)));
typedef int sk_v16i __attribute__((ext_vector_type(16)));
typedef unsigned sk_v2u __attribute__((ext_vector_type(2)));
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
// KEEP: default cache policy instead of nt -- for the gathered rows of a regrouped batch, whose first and last 128-byte
// lines are shared with the neighbouring reads and are to be found in the XCD's L2 by the tile that takes those
template <bool WIDE, bool KEEP = false>
__device__ __forceinline__ void dma_piece(const uint8_t *g, uint8_t *l)
{
    if (WIDE) __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, KEEP ? 0 : SK_DMA_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, KEEP ? 0 : SK_DMA_AUX);
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

// ---- shared by the general kernels (sk_team.hip, sk_stream.hip): reductions and scans over a team of 16 lanes
// (a DPP row) or the whole wave, and the byte-by-byte scan of one read from global memory
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

// offsets[r], offsets[r + 1] / lengths[r] for a wave-uniform r through the scalar cache.  The compiler takes
// vector loads for them (it cannot prove the arrays unwritten in a kernel that stores), and those would sit in
// the vector-memory counter between the blocks in flight: every read boundary would drain the loader.
template <typename T>
__device__ __forceinline__ const T *in_sgprs(const T *p) // the address is the same in every lane: say so
{
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return reinterpret_cast<const T *>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void scalar_load_pair(const uint64_t *p, uint64_t &x, uint64_t &y)
{
    sk_v4u v;
    p = in_sgprs(p);
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    x = ((uint64_t)v[1] << 32) | v[0];
    y = ((uint64_t)v[3] << 32) | v[2];
}
__device__ __forceinline__ uint64_t scalar_load(const unsigned long long *p)
{
    sk_v2u v;
    p = in_sgprs(p);
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return ((uint64_t)v[1] << 32) | v[0];
}
__device__ __forceinline__ uint32_t scalar_load(const uint32_t *p)
{
    uint32_t v;
    p = in_sgprs(p);
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
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

namespace {

// The dynamic-LDS ceiling of a kernel is a property of (device, function) in the HIP runtime; it is
// raised ONCE per pair to the CU's 160 KiB and every launch then passes its own size.  (Setting it per
// launch to that launch's size raced between host threads scanning batches of different strides on one
// device: A sets 80 KiB, B sets 10 KiB, A's launch fails.)  Also caches the kernel's register count.
struct kernel_facts {
    hipError_t status = hipSuccess;
    int regs = 0;
};
// single-wave workgroups a CU's registers hold of a kernel (512 per lane and SIMD, allocated in eights, four SIMDs).  The
// tile kernels' grids are persistent: a workgroup beyond that would run as a second round, a wave to a CU.
inline int reg_fit(const kernel_facts &f)
{
    const int regs = f.regs > 0 ? f.regs : 256;
    const int fits = 4 * (512 / ((regs + 7) & ~7));
    return fits < 4 ? 4 : fits;
}
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

} // namespace

#endif
