// sk_deflate.hip -- BGZF block deflate on the GPU for the -g writer (SICKLE_GZ_LEVEL=gpu).
//
// One wavefront (a 64-thread workgroup) per block of up to 65 280 bytes of output text; the
// encoding and its split into phases are in sk_deflate_block.h (shared with the host harness that
// checks it).  Blocks are independent, so a batch of a few thousand of them fills the device; the
// serial parts of a block (Huffman construction, header) run on one lane while 5 workgroups per CU
// keep the other SIMDs busy.  The caller frames the deflate streams as BGZF members (header,
// CRC-32, ISIZE) on the host.
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/sickle_amd.h"
#include "sk_deflate_block.h"

namespace {

__global__ void __launch_bounds__(SKD_LANES)
sk_bgzf_deflate_kernel(const uint8_t *__restrict__ text, const uint32_t *__restrict__ sizes, uint32_t n_blocks,
                       uint32_t *__restrict__ out_words, uint32_t *__restrict__ tok_scratch, uint32_t *__restrict__ out_sizes)
{
    __shared__ skd_shared sh;
    const int lane = (int)threadIdx.x;
    uint32_t *tok = tok_scratch + (size_t)blockIdx.x * (SKD_BLOCK_MAX + 8);
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const uint8_t *p = text + (size_t)b * SKD_BLOCK_MAX;
        const uint32_t n = sizes[b];
        uint32_t *out = out_words + (size_t)b * SKD_OUT_WORDS;
        if (n == 0) {
            if (lane == 0) out_sizes[b] = 0;
            continue;
        }
        skd_phase_clear(&sh, out, lane);
        skd_phase_count_newlines(&sh, p, n, lane);
        __syncthreads();
        if (lane == 0) skd_phase_scan_segments(&sh, n);
        __syncthreads();
        skd_phase_line_starts(&sh, p, n, lane);
        __syncthreads();
        if (lane == 0) skd_phase_close_lines(&sh, p, n);
        __syncthreads();
        skd_phase_tokenize(&sh, p, tok, lane);
        __syncthreads();
        if (lane == 0) skd_phase_codes_and_header(&sh, out);
        __syncthreads();
        skd_phase_size_lines(&sh, tok, lane);
        __syncthreads();
        if (lane == 0) skd_phase_place_lines(&sh, out);
        __syncthreads();
        skd_phase_emit(&sh, tok, out, lane);
        __syncthreads();
        if (lane == 0) out_sizes[b] = sh.total_bits > (SKD_OUT_WORDS - 2) * 32u ? 0u : (sh.total_bits + 7) / 8;
        __syncthreads();
    }
}

struct DeflateState {
    hipStream_t stream = nullptr;
    uint8_t *d_text = nullptr;
    uint32_t *d_sizes = nullptr, *d_out = nullptr, *d_tok = nullptr, *d_out_sizes = nullptr;
    size_t cap_blocks = 0, cap_grid = 0;
    int cu_count = 256;
    bool ready = false;
};
std::mutex g_lock;
DeflateState g_state[16];
thread_local char g_error[256];

#define SKD_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            snprintf(g_error, sizeof g_error, "%s failed: %s", #call, hipGetErrorString(e_));      \
            return SK_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

} // namespace

extern "C" {

const char *sk_bgzf_last_error(void) { return g_error; }

void *sk_bgzf_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void sk_bgzf_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int sk_bgzf_deflate(int device, const uint8_t *text, const uint32_t *sizes, uint32_t n_blocks, uint8_t *out, uint32_t *out_sizes)
{
    if (device < 0 || device >= 16 || (n_blocks && (!text || !sizes || !out || !out_sizes))) return SK_EINVAL;
    if (n_blocks == 0) return SK_OK;
    std::lock_guard<std::mutex> lk(g_lock); // callers are the writer threads of the output files: one batch at a time
    for (uint32_t b = 0; b < n_blocks; ++b)
        if (sizes[b] > SKD_BLOCK_MAX) { // the kernel's token scratch and LDS tables are sized for one BGZF block
            snprintf(g_error, sizeof g_error, "sizes[%u] = %u exceeds the block size %u", b, sizes[b], (unsigned)SKD_BLOCK_MAX);
            return SK_EINVAL;
        }
    DeflateState &s = g_state[device];
    SKD_HIP(hipSetDevice(device));
    if (!s.ready) {
        hipDeviceProp_t prop;
        SKD_HIP(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            snprintf(g_error, sizeof g_error, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
            return SK_ENODEV;
        }
        s.cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        SKD_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        s.ready = true;
    }
    const size_t grid = std::min<size_t>(n_blocks, (size_t)s.cu_count * 5); // 5 workgroups per CU by LDS
    if (n_blocks > s.cap_blocks) {
        if (s.d_text) (void)hipFree(s.d_text);
        if (s.d_sizes) (void)hipFree(s.d_sizes);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_out_sizes) (void)hipFree(s.d_out_sizes);
        s.cap_blocks = 0;
        const size_t cap = n_blocks + n_blocks / 4 + 16;
        SKD_HIP(hipMalloc(&s.d_text, cap * SKD_BLOCK_MAX));
        SKD_HIP(hipMalloc(&s.d_sizes, cap * sizeof(uint32_t)));
        SKD_HIP(hipMalloc(&s.d_out, cap * SKD_OUT_WORDS * sizeof(uint32_t)));
        SKD_HIP(hipMalloc(&s.d_out_sizes, cap * sizeof(uint32_t)));
        s.cap_blocks = cap;
    }
    if (grid > s.cap_grid) {
        if (s.d_tok) (void)hipFree(s.d_tok);
        s.cap_grid = 0;
        const size_t cap = (size_t)s.cu_count * 5;
        SKD_HIP(hipMalloc(&s.d_tok, cap * (SKD_BLOCK_MAX + 8) * sizeof(uint32_t)));
        s.cap_grid = cap;
    }
    // the text ends with its last block: nothing is read past sizes[n_blocks-1] bytes of it
    const size_t text_bytes = (size_t)(n_blocks - 1) * SKD_BLOCK_MAX + sizes[n_blocks - 1];
    SKD_HIP(hipMemcpyAsync(s.d_text, text, text_bytes, hipMemcpyHostToDevice, s.stream));
    SKD_HIP(hipMemcpyAsync(s.d_sizes, sizes, (size_t)n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
    hipLaunchKernelGGL(sk_bgzf_deflate_kernel, dim3((unsigned)grid), dim3(SKD_LANES), 0, s.stream, s.d_text, s.d_sizes, n_blocks,
                       s.d_out, s.d_tok, s.d_out_sizes);
    SKD_HIP(hipGetLastError());
    SKD_HIP(hipMemcpyAsync(out, s.d_out, (size_t)n_blocks * SKD_OUT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
    SKD_HIP(hipMemcpyAsync(out_sizes, s.d_out_sizes, (size_t)n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
    SKD_HIP(hipStreamSynchronize(s.stream));
    return SK_OK;
}

} // extern "C"
