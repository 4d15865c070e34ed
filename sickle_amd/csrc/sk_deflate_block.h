// sk_deflate_block.h -- one BGZF block's worth of FASTQ text -> a deflate stream, written as
// PHASES that 64 lanes of one wavefront run in step (a barrier between phases).
//
// The encoding is the one of host/FqDeflate.cpp (copies from the same column of the line four lines
// up, runs, everything else literals, one dynamic Huffman block).  The split over lanes: a LINE is
// the unit of work (line l belongs to lane l mod 64) for tokenising, counting, sizing and emitting;
// the Huffman codes and the block header are built by lane 0 between two barriers.  Bits of
// neighbouring lines meet inside one 32-bit word, so the first and the last word of a line go out
// with atomic OR into a zeroed buffer, the words in between with plain stores.
//
// The same source compiles for the device (sk_deflate.hip) and for the host, where a test harness
// runs the lanes one after the other, phase by phase (tests/cpu_shim/gpu_deflate_sim.cpp).
#ifndef SK_DEFLATE_BLOCK_H
#define SK_DEFLATE_BLOCK_H

#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define SKD_FN __device__ __forceinline__
#define SKD_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define SKD_ATOMIC_OR(p, v) atomicOr((p), (v))
#else
#define SKD_FN inline
#define SKD_ATOMIC_ADD(p, v) (*(p) += (v))
#define SKD_ATOMIC_OR(p, v) (*(p) |= (v))
#endif

#define SKD_LANES 64
#define SKD_MAX_LINES 2048      /* lines beyond this are taken as one long last line */
#define SKD_BLOCK_MAX 65280     /* input bytes per block (BGZF) */
#define SKD_OUT_WORDS 16384     /* 64 KiB of output words per block slot */
#define SKD_MIN_ALIGNED 6
#define SKD_MIN_RUN 5

// per-block working state: LDS on the device
struct skd_shared {
    uint32_t lfreq[288], dfreq[32];
    uint16_t lcode[288], dcode[32];
    uint8_t llen[288], dlen[32];
    uint32_t line_start[SKD_MAX_LINES + 1];
    uint32_t line_tokens[SKD_MAX_LINES]; // tokens of the line (they sit at tok[line_start[l]...])
    uint32_t line_bit[SKD_MAX_LINES];    // bits of the line, then (after the scan) its first bit position
    uint32_t seg_count[SKD_LANES + 1];
    uint32_t n_lines, header_bits, total_bits;
};

// tokens: literal = the byte; match = 1<<31 | (length-3)<<15 | (distance-1)
SKD_FN uint32_t skd_match(uint32_t len, uint32_t dist) { return (1u << 31) | ((len - 3) << 15) | (dist - 1); }

SKD_FN void skd_len_code(uint32_t len, uint32_t *sym, uint32_t *extra_bits, uint32_t *extra)
{
    // RFC 1951 3.2.5
    if (len == 258) {
        *sym = 28; *extra_bits = 0; *extra = 0;
        return;
    }
    if (len < 11) {
        *sym = len - 3; *extra_bits = 0; *extra = 0;
        return;
    }
    uint32_t l = len - 3, eb = 0;
    // groups of 4 codes share an extra-bit count: codes 8..11 -> 1, 12..15 -> 2, ...
    uint32_t hb = 31 - (uint32_t)__builtin_clz(l); // l >= 8
    eb = hb - 2;
    *sym = 4 * eb + 4 + ((l >> eb) & 3);
    *extra_bits = eb;
    *extra = l & ((1u << eb) - 1);
}

SKD_FN void skd_dist_code(uint32_t dist, uint32_t *sym, uint32_t *extra_bits, uint32_t *extra)
{
    uint32_t d = dist - 1;
    if (d < 4) {
        *sym = d; *extra_bits = 0; *extra = 0;
        return;
    }
    uint32_t hb = 31 - (uint32_t)__builtin_clz(d); // d >= 4
    uint32_t eb = hb - 1;
    *sym = 2 * eb + 2 + ((d >> eb) & 1);
    *extra_bits = eb;
    *extra = d & ((1u << eb) - 1);
}

// ---- phase 0: clear, count newlines per segment
SKD_FN void skd_phase_clear(skd_shared *sh, uint32_t *out_words, int lane)
{
    for (int i = lane; i < 288; i += SKD_LANES) sh->lfreq[i] = 0;
    if (lane < 32) sh->dfreq[lane] = 0;
    for (int i = lane; i < SKD_OUT_WORDS; i += SKD_LANES) out_words[i] = 0;
}

SKD_FN void skd_phase_count_newlines(skd_shared *sh, const uint8_t *p, uint32_t n, int lane)
{
    const uint32_t seg = (n + SKD_LANES - 1) / SKD_LANES;
    const uint32_t lo = (uint32_t)lane * seg, hi = lo + seg < n ? lo + seg : n;
    uint32_t c = 0;
    for (uint32_t i = lo; i < hi; ++i) c += p[i] == '\n';
    sh->seg_count[lane] = c;
}

// lane 0: exclusive scan of the segment counts (line k+1 starts after the k-th newline)
SKD_FN void skd_phase_scan_segments(skd_shared *sh, uint32_t n)
{
    uint32_t at = 1; // line 0 starts at byte 0
    for (int k = 0; k < SKD_LANES; ++k) {
        const uint32_t c = sh->seg_count[k];
        sh->seg_count[k] = at;
        at += c;
    }
    sh->seg_count[SKD_LANES] = at;
    sh->line_start[0] = 0;
    (void)n;
}

SKD_FN void skd_phase_line_starts(skd_shared *sh, const uint8_t *p, uint32_t n, int lane)
{
    const uint32_t seg = (n + SKD_LANES - 1) / SKD_LANES;
    const uint32_t lo = (uint32_t)lane * seg, hi = lo + seg < n ? lo + seg : n;
    uint32_t k = sh->seg_count[lane];
    for (uint32_t i = lo; i < hi; ++i)
        if (p[i] == '\n') {
            if (k < SKD_MAX_LINES && i + 1 < n) sh->line_start[k] = i + 1;
            ++k;
        }
}

// lane 0: the number of lines (a newline that ends the block starts no line)
SKD_FN void skd_phase_close_lines(skd_shared *sh, const uint8_t *p, uint32_t n)
{
    uint32_t lines = sh->seg_count[SKD_LANES]; // 1 + newlines
    if (n && p[n - 1] == '\n') --lines;
    if (n == 0) lines = 0;
    if (lines > SKD_MAX_LINES) lines = SKD_MAX_LINES;
    sh->n_lines = lines;
    sh->line_start[lines] = n;
}

// ---- phase 1: tokens of the lines of this lane, and their counts
SKD_FN void skd_phase_tokenize(skd_shared *sh, const uint8_t *p, uint32_t *tok, int lane)
{
    const uint32_t L = sh->n_lines;
    for (uint32_t l = (uint32_t)lane; l < L; l += SKD_LANES) {
        const uint32_t i = sh->line_start[l], end = sh->line_start[l + 1];
        uint32_t *t = tok + i;
        uint32_t nt = 0;
        const bool has_ref = l >= 4;
        const uint32_t ref_line = has_ref ? sh->line_start[l - 4] : 0, ref_end = has_ref ? sh->line_start[l - 3] : 0;
        bool use_ref = false;
        if (has_ref && i - ref_line <= 32768) {
            uint32_t head = SKD_MIN_ALIGNED;
            if (end - i < head) head = end - i;
            if (ref_end - ref_line < head) head = ref_end - ref_line;
            use_ref = head > 0;
            for (uint32_t k = 0; k < head && use_ref; ++k) use_ref = p[i + k] == p[ref_line + k];
        }
        int shift = 0;
        uint32_t j = i;
        while (j < end) {
            uint32_t best = 0, best_ref = 0;
            if (use_ref) {
                const int tries[5] = {0, 1, -1, 2, -2};
                for (int q = 0; q < 5; ++q) {
                    const int col = (int)(j - i) + shift + tries[q];
                    if (col < 0) continue;
                    const uint32_t r = ref_line + (uint32_t)col;
                    if (r >= ref_end || j - r > 32768) continue;
                    uint32_t lim = end - j;
                    if (ref_end - r < lim) lim = ref_end - r;
                    if (lim > 258) lim = 258;
                    uint32_t m = 0;
                    while (m < lim && p[r + m] == p[j + m]) ++m;
                    if (m > best) {
                        best = m;
                        best_ref = r;
                    }
                    if (best >= SKD_MIN_ALIGNED) break;
                }
            }
            uint32_t run = 0;
            if (j > 0 && p[j] == p[j - 1]) {
                uint32_t lim = end - j;
                if (lim > 258) lim = 258;
                run = 1;
                while (run < lim && p[j + run] == p[j]) ++run;
            }
            uint32_t token;
            if (best >= SKD_MIN_ALIGNED && best >= run) {
                token = skd_match(best, j - best_ref);
                shift = (int)best_ref - (int)ref_line - (int)(j - i);
                j += best;
            } else if (run >= SKD_MIN_RUN) {
                token = skd_match(run, 1);
                j += run;
            } else {
                token = p[j];
                ++j;
            }
            t[nt++] = token;
            if (token >> 31) {
                uint32_t s, eb, ex;
                skd_len_code(((token >> 15) & 0xff) + 3, &s, &eb, &ex);
                SKD_ATOMIC_ADD(&sh->lfreq[257 + s], 1u);
                skd_dist_code((token & 0x7fff) + 1, &s, &eb, &ex);
                SKD_ATOMIC_ADD(&sh->dfreq[s], 1u);
            } else {
                SKD_ATOMIC_ADD(&sh->lfreq[token], 1u);
            }
        }
        sh->line_tokens[l] = nt;
    }
}

// ---- phase 2 (lane 0): the two Huffman codes and the block header
// length-limited canonical code for n <= 288 symbols; work arrays on the caller's stack
SKD_FN void skd_huffman(const uint32_t *freq, int n, int max_len, uint8_t *lens, uint16_t *codes)
{
    uint16_t order[288]; // used symbols, ascending by frequency
    int used = 0;
    for (int s = 0; s < n; ++s) {
        lens[s] = 0;
        codes[s] = 0;
        if (freq[s]) order[used++] = (uint16_t)s;
    }
    if (used == 0) return;
    if (used == 1) {
        lens[order[0]] = 1;
        lens[order[0] == 0 ? 1 : 0] = 1;
    } else {
        for (int i = 1; i < used; ++i) { // insertion sort, ascending (ties: lower symbol first)
            const uint16_t s = order[i];
            int j = i - 1;
            while (j >= 0 && (freq[order[j]] > freq[s] || (freq[order[j]] == freq[s] && order[j] > s))) {
                order[j + 1] = order[j];
                --j;
            }
            order[j + 1] = s;
        }
        // two queues: leaves (sorted) and internal nodes (created in non-decreasing weight)
        uint32_t weight[288]; // of internal node k
        int16_t parent[576];  // nodes 0..used-1 leaves (in `order`), used.. internal
        int leaf = 0, inode = 0, made = 0;
        for (int k = 0; k < used - 1; ++k) {
            uint32_t w = 0;
            for (int pick = 0; pick < 2; ++pick) {
                const bool take_leaf = leaf < used && (inode >= made || freq[order[leaf]] <= weight[inode]);
                if (take_leaf) {
                    w += freq[order[leaf]];
                    parent[leaf++] = (int16_t)(used + made);
                } else {
                    w += weight[inode];
                    parent[used + inode++] = (int16_t)(used + made);
                }
            }
            weight[made++] = w;
        }
        const int root = used + made - 1;
        int count[40];
        for (int l = 0; l < 40; ++l) count[l] = 0;
        uint8_t depth_in[288]; // depth of internal node k
        depth_in[made - 1] = 0;
        for (int k = made - 2; k >= 0; --k) depth_in[k] = (uint8_t)(depth_in[parent[used + k] - used] + 1);
        for (int i = 0; i < used; ++i) {
            int d = depth_in[parent[i] - used] + 1;
            if (d > 39) d = 39;
            count[d]++;
        }
        (void)root;
        for (int l = max_len + 1; l < 40; ++l) {
            count[max_len] += count[l];
            count[l] = 0;
        }
        uint32_t total = 0;
        for (int l = max_len; l > 0; --l) total += (uint32_t)count[l] << (max_len - l);
        while (total != (1u << max_len)) {
            count[max_len]--;
            for (int l = max_len - 1; l > 0; --l)
                if (count[l]) {
                    count[l]--;
                    count[l + 1] += 2;
                    break;
                }
            total--;
        }
        // shortest codes to the most frequent symbols: walk `order` from its end
        int at = used - 1;
        for (int l = 1; l <= max_len; ++l)
            for (int k = 0; k < count[l]; ++k) lens[order[at--]] = (uint8_t)l;
    }
    uint32_t next_code[17];
    int bl_count[17];
    for (int l = 0; l < 17; ++l) bl_count[l] = 0;
    for (int s = 0; s < n; ++s) bl_count[lens[s]]++;
    bl_count[0] = 0;
    uint32_t code = 0;
    next_code[0] = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code + (uint32_t)bl_count[l - 1]) << 1;
        next_code[l] = code;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = next_code[l]++;
        uint32_t r = 0;
        for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
        codes[s] = (uint16_t)r;
    }
}

struct skd_bits { // serial bit writer into the (zeroed) output words
    uint32_t *w;
    uint32_t pos; // in bits
};
SKD_FN void skd_put(skd_bits *b, uint32_t bits, uint32_t count)
{
    if (!count) return;
    const uint32_t word = b->pos >> 5, off = b->pos & 31;
    b->w[word] |= bits << off;
    if (off + count > 32) b->w[word + 1] |= bits >> (32 - off);
    b->pos += count;
}

SKD_FN void skd_phase_codes_and_header(skd_shared *sh, uint32_t *out_words)
{
    sh->lfreq[256] = 1;
    int dused = 0;
    for (int s = 0; s < 30; ++s) dused += sh->dfreq[s] != 0;
    for (int s = 0; s < 30 && dused < 2; ++s)
        if (!sh->dfreq[s]) {
            sh->dfreq[s] = 1;
            ++dused;
        }
    skd_huffman(sh->lfreq, 286, 15, sh->llen, sh->lcode);
    skd_huffman(sh->dfreq, 30, 15, sh->dlen, sh->dcode);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && sh->llen[hlit - 1] == 0) --hlit;
    while (hdist > 1 && sh->dlen[hdist - 1] == 0) --hdist;
    uint8_t all[316];
    for (int i = 0; i < hlit; ++i) all[i] = sh->llen[i];
    for (int i = 0; i < hdist; ++i) all[hlit + i] = sh->dlen[i];
    const int total = hlit + hdist;
    uint8_t cl_sym[316], cl_eb[316], cl_ex[316];
    int ncl = 0;
    uint32_t clfreq[19];
    for (int i = 0; i < 19; ++i) clfreq[i] = 0;
    for (int i = 0; i < total;) {
        int run = 1;
        while (i + run < total && all[i + run] == all[i]) ++run;
        const uint8_t v = all[i];
        int left = run;
        if (v == 0) {
            while (left >= 11) {
                const int r = left < 138 ? left : 138;
                cl_sym[ncl] = 18; cl_eb[ncl] = 7; cl_ex[ncl++] = (uint8_t)(r - 11);
                left -= r;
            }
            if (left >= 3) {
                cl_sym[ncl] = 17; cl_eb[ncl] = 3; cl_ex[ncl++] = (uint8_t)(left - 3);
                left = 0;
            }
        } else {
            cl_sym[ncl] = v; cl_eb[ncl] = 0; cl_ex[ncl++] = 0;
            --left;
            while (left >= 3) {
                const int r = left < 6 ? left : 6;
                cl_sym[ncl] = 16; cl_eb[ncl] = 2; cl_ex[ncl++] = (uint8_t)(r - 3);
                left -= r;
            }
        }
        while (left-- > 0) {
            cl_sym[ncl] = v; cl_eb[ncl] = 0; cl_ex[ncl++] = 0;
        }
        i += run;
    }
    for (int i = 0; i < ncl; ++i) clfreq[cl_sym[i]]++;
    uint8_t cllen[19];
    uint16_t clcode[19];
    skd_huffman(clfreq, 19, 7, cllen, clcode);
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cllen[order[hclen - 1]] == 0) --hclen;
    skd_bits b = {out_words, 0};
    skd_put(&b, 1, 1);
    skd_put(&b, 2, 2);
    skd_put(&b, (uint32_t)(hlit - 257), 5);
    skd_put(&b, (uint32_t)(hdist - 1), 5);
    skd_put(&b, (uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) skd_put(&b, cllen[order[i]], 3);
    for (int i = 0; i < ncl; ++i) {
        skd_put(&b, clcode[cl_sym[i]], cllen[cl_sym[i]]);
        skd_put(&b, cl_ex[i], cl_eb[i]);
    }
    sh->header_bits = b.pos;
}

// ---- phase 3: bits of each line
SKD_FN void skd_phase_size_lines(skd_shared *sh, const uint32_t *tok, int lane)
{
    const uint32_t L = sh->n_lines;
    for (uint32_t l = (uint32_t)lane; l < L; l += SKD_LANES) {
        const uint32_t *t = tok + sh->line_start[l];
        const uint32_t nt = sh->line_tokens[l];
        uint32_t bits = 0;
        for (uint32_t k = 0; k < nt; ++k) {
            const uint32_t token = t[k];
            if (token >> 31) {
                uint32_t s, eb, ex;
                skd_len_code(((token >> 15) & 0xff) + 3, &s, &eb, &ex);
                bits += sh->llen[257 + s] + eb;
                skd_dist_code((token & 0x7fff) + 1, &s, &eb, &ex);
                bits += sh->dlen[s] + eb;
            } else {
                bits += sh->llen[token];
            }
        }
        sh->line_bit[l] = bits;
    }
}

// ---- phase 4 (lane 0): where each line's bits start; the end-of-block code after the last
SKD_FN void skd_phase_place_lines(skd_shared *sh, uint32_t *out_words)
{
    uint32_t at = sh->header_bits;
    for (uint32_t l = 0; l < sh->n_lines; ++l) {
        const uint32_t bits = sh->line_bit[l];
        sh->line_bit[l] = at;
        at += bits;
    }
    sh->total_bits = at + sh->llen[256];
    if (sh->total_bits <= (SKD_OUT_WORDS - 2) * 32u) {
        skd_bits b = {out_words, at};
        skd_put(&b, sh->lcode[256], sh->llen[256]);
    }
}

// ---- phase 5: the lines' bits.  Words wholly inside a line are stored, its first and last OR-ed.
SKD_FN void skd_phase_emit(skd_shared *sh, const uint32_t *tok, uint32_t *out_words, int lane)
{
    if (sh->total_bits > (SKD_OUT_WORDS - 2) * 32u) return; // does not fit the slot: the caller stores the text instead
    const uint32_t L = sh->n_lines;
    for (uint32_t l = (uint32_t)lane; l < L; l += SKD_LANES) {
        const uint32_t *t = tok + sh->line_start[l];
        const uint32_t nt = sh->line_tokens[l];
        uint32_t pos = sh->line_bit[l];
        uint32_t word = pos >> 5;
        uint64_t acc = 0;
        uint32_t have = pos & 31; // bits of `acc` in use (the low `pos & 31` are another line's: kept zero)
        bool first = true;
        for (uint32_t k = 0; k < nt; ++k) {
            const uint32_t token = t[k];
            uint64_t bits;
            uint32_t count;
            if (token >> 31) {
                uint32_t s, eb, ex, s2, eb2, ex2;
                skd_len_code(((token >> 15) & 0xff) + 3, &s, &eb, &ex);
                skd_dist_code((token & 0x7fff) + 1, &s2, &eb2, &ex2);
                bits = sh->lcode[257 + s];
                count = sh->llen[257 + s];
                bits |= (uint64_t)ex << count;
                count += eb;
                bits |= (uint64_t)sh->dcode[s2] << count;
                count += sh->dlen[s2];
                bits |= (uint64_t)ex2 << count;
                count += eb2;
            } else {
                bits = sh->lcode[token];
                count = sh->llen[token];
            }
            // at most 31 + 48 bits: a 48-bit token may need two flushes
            uint32_t take = count > 32 ? 32 : count;
            acc |= (bits & ((1ull << take) - 1)) << have;
            have += take;
            if (have >= 32) {
                if (first) SKD_ATOMIC_OR(&out_words[word], (uint32_t)acc);
                else out_words[word] = (uint32_t)acc;
                first = false;
                ++word;
                acc >>= 32;
                have -= 32;
            }
            if (count > 32) {
                acc |= (bits >> 32) << have;
                have += count - 32;
                if (have >= 32) {
                    if (first) SKD_ATOMIC_OR(&out_words[word], (uint32_t)acc);
                    else out_words[word] = (uint32_t)acc;
                    first = false;
                    ++word;
                    acc >>= 32;
                    have -= 32;
                }
            }
        }
        if (have) SKD_ATOMIC_OR(&out_words[word], (uint32_t)acc);
    }
}

#endif
