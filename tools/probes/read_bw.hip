// probe: what read bandwidth does a streaming kernel get on this device, by load mechanism?
// (a) global_load_dwordx4 into VGPRs, XOR-folded; (b) the same with the nt cache policy;
// (c) LDS-DMA (global_load_lds_dwordx4) without touching the LDS contents.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void read_vgpr(const v4u *__restrict__ src, size_t n16, uint32_t *out)
{
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    v4u acc = {0, 0, 0, 0};
    for (; i + 256 * (UNROLL - 1) < n16; i += stride) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + 256 * u) : src[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = 1;
}

// one wave per block, each trip DMAs 4 x 1 KiB into LDS
template <int AUX>
__global__ __launch_bounds__(64) void read_lds_dma(const unsigned char *__restrict__ src, size_t n_tiles, size_t tile_bytes, uint32_t *out)
{
    extern __shared__ unsigned char lds[];
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const unsigned char *g = src + t * tile_bytes + threadIdx.x * 16;
        for (size_t off = 0; off < tile_bytes; off += 1024) {
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(g + off), (void __attribute__((address_space(3))) *)(lds + off), 16, 0, AUX);
        }
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lds[threadIdx.x] == 0xee && n_tiles == 1) out[0] = 1;
}

// (d) the kernel's own traffic shape without its arithmetic: one wave per workgroup, ten 1 KiB
// loads per 9728-byte tile (the last one half), WRITE = 8 bytes per lane per tile (the cut pairs)
template <int WRITE, int NT>
__global__ __launch_bounds__(64) void read_tiles(const unsigned char *__restrict__ src, size_t n_tiles, uint2 *out)
{
    v4u acc = {0, 0, 0, 0};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const unsigned char *g = src + t * 9728;
        v4u v[10];
#pragma unroll
        for (int p = 0; p < 10; ++p) {
            unsigned off = p * 1024 + threadIdx.x * 16;
            if (p == 9 && off > 9728 - 16) off = 9728 - 16;
            v[p] = NT ? __builtin_nontemporal_load((const v4u *)(g + off)) : *(const v4u *)(g + off);
        }
#pragma unroll
        for (int p = 0; p < 10; ++p) { acc.x ^= v[p].x; acc.y ^= v[p].y; acc.z ^= v[p].z; acc.w ^= v[p].w; }
        if (WRITE == 1) out[t * 64 + threadIdx.x] = make_uint2(acc.x, acc.y);
        if (WRITE == 2) { typedef unsigned v2u __attribute__((ext_vector_type(2))); v2u c = {acc.x, acc.y}; __builtin_nontemporal_store(c, (v2u *)out + t * 64 + threadIdx.x); }
        if (WRITE == 3 && threadIdx.x < 32) { v4u c = {acc.x, acc.y, acc.z, acc.w}; ((v4u *)out)[t * 32 + threadIdx.x] = c; }
    }
    if (!WRITE && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = make_uint2(1, 1);
}

// (e) as (d), but a wave takes GROUP consecutive tiles per round and writes their cut pairs as one
// contiguous burst (GROUP x 512 bytes) afterwards
template <int GROUP>
__global__ __launch_bounds__(64) void read_tiles_grouped(const unsigned char *__restrict__ src, size_t n_tiles, uint2 *out)
{
    const size_t n_groups = n_tiles / GROUP;
    for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        uint2 cuts[GROUP];
#pragma unroll
        for (int k = 0; k < GROUP; ++k) {
            const unsigned char *gp = src + (g * GROUP + k) * 9728;
            v4u v[10];
            v4u acc = {0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < 10; ++p) {
                unsigned off = p * 1024 + threadIdx.x * 16;
                if (p == 9 && off > 9728 - 16) off = 9728 - 16;
                v[p] = __builtin_nontemporal_load((const v4u *)(gp + off));
            }
#pragma unroll
            for (int p = 0; p < 10; ++p) { acc.x ^= v[p].x; acc.y ^= v[p].y; acc.z ^= v[p].z; acc.w ^= v[p].w; }
            cuts[k] = make_uint2(acc.x ^ acc.z, acc.y ^ acc.w);
        }
#pragma unroll
        for (int k = 0; k < GROUP; ++k) out[(g * GROUP + k) * 64 + threadIdx.x] = cuts[k];
    }
}

int main()
{
    const size_t bytes = 1520000000ull / 9728 * 9728;
    unsigned char *d; uint32_t *o;
    CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&o, 4096));
    CK(hipMemset(d, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0;
        for (int r = 0; r < 20; ++r) {
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-44s avg %.4f ms  %.0f GB/s   best %.4f ms %.0f GB/s\n", name, sum / 20, bytes / (sum / 20) / 1e6, best, bytes / best / 1e6);
    };
    const size_t n16 = bytes / 16;
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        char nm[96];
        snprintf(nm, sizeof nm, "vgpr x4 unroll4 blocks=%d", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL((read_vgpr<0, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u *)d, n16, o); });
        snprintf(nm, sizeof nm, "vgpr x4 unroll4 nt blocks=%d", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL((read_vgpr<1, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u *)d, n16, o); });
    }
    timeit("vgpr x4 unroll8 nt blocks=8192", [&] { hipLaunchKernelGGL((read_vgpr<1, 8>), dim3(8192), dim3(256), 0, 0, (const v4u *)d, n16, o); });
    const size_t tile = 9728, n_tiles = bytes / tile;
    for (int blocks : {4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "lds dma aux0 blocks=%d (9.5 KiB/wave)", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL((read_lds_dma<0>), dim3(blocks), dim3(64), 9728 + 1024, 0, d, n_tiles, tile, o); });
        snprintf(nm, sizeof nm, "lds dma aux2(nt) blocks=%d", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL((read_lds_dma<2>), dim3(blocks), dim3(64), 9728 + 1024, 0, d, n_tiles, tile, o); });
    }
    uint2 *cuts; CK(hipMalloc(&cuts, n_tiles * 64 * 8));
    for (int per_cu : {12, 16}) {
        char nm[96];
        snprintf(nm, sizeof nm, "tiles vgpr nt, no writes, %d waves/CU", per_cu);
        timeit(nm, [&] { hipLaunchKernelGGL((read_tiles<0, 1>), dim3(256 * per_cu), dim3(64), 0, 0, d, n_tiles, cuts); });
        snprintf(nm, sizeof nm, "tiles vgpr nt, 8 B/lane written, %d waves/CU", per_cu);
        timeit(nm, [&] { hipLaunchKernelGGL((read_tiles<1, 1>), dim3(256 * per_cu), dim3(64), 0, 0, d, n_tiles, cuts); });
        snprintf(nm, sizeof nm, "tiles vgpr nt, 8 B/lane nt store, %d waves/CU", per_cu);
        timeit(nm, [&] { hipLaunchKernelGGL((read_tiles<2, 1>), dim3(256 * per_cu), dim3(64), 0, 0, d, n_tiles, cuts); });
        snprintf(nm, sizeof nm, "tiles vgpr nt, 16 B x 32 lanes, %d waves/CU", per_cu);
        timeit(nm, [&] { hipLaunchKernelGGL((read_tiles<3, 1>), dim3(256 * per_cu), dim3(64), 0, 0, d, n_tiles, cuts); });
    }
    timeit("tiles grouped x4, 2 KiB write bursts, 12 waves/CU", [&] { hipLaunchKernelGGL((read_tiles_grouped<4>), dim3(256 * 12), dim3(64), 0, 0, d, n_tiles, cuts); });
    timeit("tiles grouped x8, 4 KiB write bursts, 12 waves/CU", [&] { hipLaunchKernelGGL((read_tiles_grouped<8>), dim3(256 * 12), dim3(64), 0, 0, d, n_tiles, cuts); });
    timeit("tiles grouped x16, 8 KiB write bursts, 12 waves/CU", [&] { hipLaunchKernelGGL((read_tiles_grouped<16>), dim3(256 * 12), dim3(64), 0, 0, d, n_tiles, cuts); });
    timeit("tiles vgpr nt, 8 B/lane written, 12 waves/CU (again)", [&] { hipLaunchKernelGGL((read_tiles<1, 1>), dim3(256 * 12), dim3(64), 0, 0, d, n_tiles, cuts); });
    timeit("tiles vgpr nt, no writes, 12 waves/CU (again)", [&] { hipLaunchKernelGGL((read_tiles<0, 1>), dim3(256 * 12), dim3(64), 0, 0, d, n_tiles, cuts); });
    return 0;
}
