// probe: how to get PACKED rows (150-byte reads back to back, rows at odd addresses) into an LDS tile.
//  (a) LDS-DMA 16 B per lane from the aligned span (what sk_scan_tile_any_kernel does): image = global
//      image, rows unaligned in LDS  -> then: what do unaligned ds_read_b64 / ds_read_b32 cost?  (kernel C)
//  (b) LDS-DMA 4 B per lane with a per-lane UNALIGNED global address: lane i of piece p fetches dword
//      (row, k) -> the LDS image has rows at a pitch of 152 bytes, 8-byte aligned           (kernel A)
//  (c) LDS-DMA 16 B per lane with a per-lane unaligned global address: rows at a pitch of 160 (kernel B)
// Each loader is first checked for correctness on a small buffer, then timed over 1.5 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;
constexpr int L = 150;

// MODE 0: aligned-span 16-byte DMA (10 pieces, image = global image)
// MODE 1: 4-byte DMA, per-lane address, rows at pitch 152 (38 pieces)
// MODE 2: 16-byte DMA, per-lane unaligned address, rows at pitch 160 (10 pieces)
template <int MODE, int AUX>
__global__ __launch_bounds__(64) void load_tiles(const unsigned char *__restrict__ src, size_t n_tiles, unsigned char *dump, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const unsigned char *g = src + t * (64 * L);
        if (MODE == 0) {
            const size_t mis = (size_t)g & 15;
            const unsigned char *ga = g - mis;
#pragma unroll
            for (int p = 0; p < 10; ++p)
                __builtin_amdgcn_global_load_lds((gptr_t)(ga + p * 1024 + lane * 16), (lptr_t)(lds + p * 1024), 16, 0, AUX);
        } else if (MODE == 1) {
            int r = lane / 38, k = lane % 38; // dword i = 64 p + lane of the 64 x 38 dword image
#pragma unroll
            for (int p = 0; p < 38; ++p) {
                __builtin_amdgcn_global_load_lds((gptr_t)(g + r * L + 4 * k), (lptr_t)(lds + p * 256), 4, 0, AUX);
                k += 26; r += 1; // 64 = 38 + 26
                if (k >= 38) { k -= 38; r += 1; }
            }
        } else {
            int r = lane / 10, c = lane % 10; // 16-byte chunk i = 64 p + lane of the 64 x 10 chunk image
#pragma unroll
            for (int p = 0; p < 10; ++p) {
                __builtin_amdgcn_global_load_lds((gptr_t)(g + r * L + 16 * c), (lptr_t)(lds + p * 1024), 16, 0, AUX);
                c += 4; r += 6; // 64 = 6 * 10 + 4
                if (c >= 10) { c -= 10; r += 1; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dump && t == 0) {
            for (int i = lane; i < 10240; i += 64) dump[i] = lds[i];
        }
        acc ^= *(const uint32_t *)(lds + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u) out[0] = 1;
}

// kernel C: LDS read cost.  64 rows at `pitch`, starting at byte `base`; every lane walks its row with
// 19 ds_read_b64 (or 38 ds_read_b32), REPS times; no global traffic.
typedef uint64_t __attribute__((aligned(1))) u64u;
typedef uint32_t __attribute__((aligned(1))) u32u;
template <int WIDE>
__global__ __launch_bounds__(64) void lds_walk(int pitch, int base, int reps, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 12288 / 4; i += 64) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned char *row = lds + base + threadIdx.x * pitch;
    uint64_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        if (WIDE) {
#pragma unroll
            for (int k = 0; k < 19; ++k) acc ^= *(const u64u *)(row + 8 * k);
        } else {
#pragma unroll
            for (int k = 0; k < 38; ++k) acc ^= *(const u32u *)(row + 4 * k);
        }
        asm volatile("" : "+v"(acc));
    }
    if (acc == 0x1234567887654321ull) out[0] = 1;
}

int main()
{
    const size_t tile = 64 * L;
    const size_t n_tiles = 1520000000ull / tile;
    const size_t bytes = n_tiles * tile;
    unsigned char *d, *dump; uint32_t *o;
    CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&o, 4096)); CK(hipMalloc(&dump, 10240));
    std::vector<unsigned char> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 131u + (i >> 8) * 7u) & 0xff);
    CK(hipMemset(d, 1, bytes + 4096));
    CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<unsigned char> img(10240);
    // ---- correctness of the two re-striding loaders on tile 0
    for (int mode = 1; mode <= 2; ++mode) {
        CK(hipMemset(dump, 0, 10240));
        if (mode == 1) hipLaunchKernelGGL((load_tiles<1, 0>), dim3(1), dim3(64), 10240 + 1024, 0, d, (size_t)1, dump, o);
        else hipLaunchKernelGGL((load_tiles<2, 0>), dim3(1), dim3(64), 10240 + 1024, 0, d, (size_t)1, dump, o);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(img.data(), dump, 10240, hipMemcpyDeviceToHost));
        const int pitch = mode == 1 ? 152 : 160;
        int bad = 0;
        for (int r = 0; r < 64; ++r)
            for (int x = 0; x < L; ++x)
                if (img[r * pitch + x] != h[r * L + x]) { if (bad < 3) printf("  mode %d: row %d byte %d: %02x != %02x\n", mode, r, x, img[r * pitch + x], h[r * L + x]); ++bad; }
        printf("re-striding loader mode %d (pitch %d): %s (%d bad bytes)\n", mode, pitch, bad ? "WRONG" : "correct", bad);
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, double gb, auto launch) {
        for (int w = 0; w < 20; ++w) launch();
        CK(hipDeviceSynchronize());
        const int R = 20;
        CK(hipEventRecord(a));
        for (int r = 0; r < R; ++r) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
        if (gb > 0) printf("%-52s %.4f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3);
        else printf("%-52s %.4f ms\n", name, ms);
    };
    const double gb = bytes / 1e9;
    for (int per_cu : {12, 16}) {
        char nm[96];
        snprintf(nm, sizeof nm, "aligned-span 16 B DMA nt, %d waves/CU", per_cu);
        timeit(nm, gb, [&] { hipLaunchKernelGGL((load_tiles<0, 2>), dim3(256 * per_cu), dim3(64), 10240, 0, d, n_tiles, (unsigned char *)nullptr, o); });
        snprintf(nm, sizeof nm, "4 B DMA per-lane address nt (pitch 152), %d waves/CU", per_cu);
        timeit(nm, gb, [&] { hipLaunchKernelGGL((load_tiles<1, 2>), dim3(256 * per_cu), dim3(64), 10240, 0, d, n_tiles, (unsigned char *)nullptr, o); });
        snprintf(nm, sizeof nm, "4 B DMA per-lane address default policy, %d waves/CU", per_cu);
        timeit(nm, gb, [&] { hipLaunchKernelGGL((load_tiles<1, 0>), dim3(256 * per_cu), dim3(64), 10240, 0, d, n_tiles, (unsigned char *)nullptr, o); });
        snprintf(nm, sizeof nm, "16 B DMA unaligned per-lane address nt (pitch 160), %d", per_cu);
        timeit(nm, gb, [&] { hipLaunchKernelGGL((load_tiles<2, 2>), dim3(256 * per_cu), dim3(64), 10240, 0, d, n_tiles, (unsigned char *)nullptr, o); });
        snprintf(nm, sizeof nm, "16 B DMA unaligned per-lane address default (pitch 160), %d", per_cu);
        timeit(nm, gb, [&] { hipLaunchKernelGGL((load_tiles<2, 0>), dim3(256 * per_cu), dim3(64), 10240, 0, d, n_tiles, (unsigned char *)nullptr, o); });
    }
    // ---- LDS walks: 16 waves per CU, 2000 row walks each
    for (int wide = 1; wide >= 0; --wide)
        for (int pitch : {152, 150, 160, 151}) {
            for (int base : {0, 2, 4, 1}) {
                char nm[96];
                snprintf(nm, sizeof nm, "lds walk %s pitch %d base %d", wide ? "b64" : "b32", pitch, base);
                if (wide) timeit(nm, 0, [&] { hipLaunchKernelGGL((lds_walk<1>), dim3(256 * 16), dim3(64), 12288 + 512, 0, pitch, base, 2000, o); });
                else timeit(nm, 0, [&] { hipLaunchKernelGGL((lds_walk<0>), dim3(256 * 16), dim3(64), 12288 + 512, 0, pitch, base, 2000, o); });
            }
        }
    return 0;
}
