// TEST INFRASTRUCTURE ONLY: sk_bgzf_deflate for the host-only build of the CLI (sickle_hostcheck):
// the phases of sickle_amd/csrc/sk_deflate_block.h, the code the GPU kernel runs, executed here lane
// after lane.  Never linked into the product.
#include "sk_deflate_block.h"
#include "sickle_amd.h"
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
const char *sk_bgzf_last_error(void) { return ""; }
void *sk_bgzf_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void sk_bgzf_host_free(void *p) { free(p); }
int sk_bgzf_deflate(int, const uint8_t *text, const uint32_t *sizes, uint32_t n_blocks, uint8_t *out, uint32_t *out_sizes)
{
    static thread_local skd_shared sh;
    std::vector<uint32_t> tok(SKD_BLOCK_MAX + 8);
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint8_t *p = text + (size_t)b * SKD_BLOCK_MAX;
        const uint32_t n = sizes[b];
        uint32_t *w = (uint32_t *)(out + (size_t)b * 65536);
        if (n == 0) {
            out_sizes[b] = 0;
            continue;
        }
#define ALL_LANES(call) for (int lane = 0; lane < SKD_LANES; ++lane) { call; }
        ALL_LANES(skd_phase_clear(&sh, w, lane));
        ALL_LANES(skd_phase_count_newlines(&sh, p, n, lane));
        skd_phase_scan_segments(&sh, n);
        ALL_LANES(skd_phase_line_starts(&sh, p, n, lane));
        skd_phase_close_lines(&sh, p, n);
        ALL_LANES(skd_phase_tokenize(&sh, p, tok.data(), lane));
        skd_phase_codes_and_header(&sh, w);
        ALL_LANES(skd_phase_size_lines(&sh, tok.data(), lane));
        skd_phase_place_lines(&sh, w);
        ALL_LANES(skd_phase_emit(&sh, tok.data(), w, lane));
        out_sizes[b] = sh.total_bits > (SKD_OUT_WORDS - 2) * 32u ? 0u : (sh.total_bits + 7) / 8;
    }
    return SK_OK;
}
}
