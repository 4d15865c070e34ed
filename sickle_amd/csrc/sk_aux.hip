// sk_aux.hip -- small kernels beside the scan: the read-bandwidth probe (bench.py's second roofline denominator)
// and the pair classification of reference src/trim_paired.cpp:543-567 over the cuts of a scan.
#include "sk_kernel_common.h"

// ------------------------------------------------------------------------------------------
// measurement aid: what a read-only stream of this buffer gets on this device (sk_probe_read_bandwidth)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sk_read_probe_kernel(const sk_v4u *__restrict__ src, size_t n16, uint32_t *sink)
{
    constexpr int UNROLL = 4;
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256 * UNROLL;
    sk_v4u acc = {0, 0, 0, 0};
    for (; i + 256 * (UNROLL - 1) < n16; i += step) {
        sk_v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    // keeps the loads alive; quality bytes never fold to this value
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && n16 == 1) sink[threadIdx.x & 1] = 1;
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_read_probe(const void *buf, size_t bytes, uint32_t *sink, int cu_count,
                                           hipStream_t stream)
{
    hipLaunchKernelGGL(sk_read_probe_kernel, dim3((unsigned)cu_count * 32u), dim3(256), 0, stream,
                       reinterpret_cast<const sk_v4u *>(buf), bytes / 16, sink);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// pair classification: reference src/trim_paired.cpp:543-567 over the cuts of a scan (mates at 2k, 2k+1).
// One 16-byte load per pair and lane; the four class counts of a wave come from ballots, one lane adds them.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sk_pair_count_kernel(const sk_v4i *__restrict__ cuts, uint64_t n_pairs, uint8_t *__restrict__ classes,
                                                            unsigned long long *counters)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long c_both = 0, c_first = 0, c_second = 0, c_none = 0; // per wave, kept by every lane
    for (uint64_t k0 = (uint64_t)blockIdx.x * 256; k0 < n_pairs; k0 += stride) {
        const uint64_t k = k0 + threadIdx.x;
        int cls = -1;
        if (k < n_pairs) {
            const sk_v4i c = __builtin_nontemporal_load(cuts + k); // {five1, three1, five2, three2}
            const bool r1 = c[1] >= 0, r2 = c[3] >= 0;             // src/trim_paired.cpp:500,502
            cls = r1 ? (r2 ? 0 : 1) : (r2 ? 2 : 3);
            if (classes) classes[k] = (uint8_t)cls;
        }
        c_both += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 0));
        c_first += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 1));
        c_second += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 2));
        c_none += __builtin_popcountll(__builtin_amdgcn_ballot_w64(cls == 3));
    }
    if ((threadIdx.x & 63) == 0) {
        if (c_both) atomicAdd(counters + 0, c_both);
        if (c_first) atomicAdd(counters + 1, c_first);
        if (c_second) atomicAdd(counters + 2, c_second);
        if (c_none) atomicAdd(counters + 3, c_none);
    }
}

extern "C" __attribute__((visibility("hidden"))) hipError_t sk_launch_pair_count(const sk_cut_dev *cuts, uint64_t n_pairs, uint8_t *classes,
                                           unsigned long long *counters, int cu_count, hipStream_t stream)
{
    if (n_pairs == 0) return hipSuccess;
    uint64_t grid = (n_pairs + 255) / 256;
    if (grid > (uint64_t)cu_count * 16) grid = (uint64_t)cu_count * 16;
    hipLaunchKernelGGL(sk_pair_count_kernel, dim3((unsigned)grid), dim3(256), 0, stream, reinterpret_cast<const sk_v4i *>(cuts), n_pairs,
                       classes, counters);
    return hipGetLastError();
}
