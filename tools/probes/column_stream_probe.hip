// probe: can a wave stream 64 LONG rows side by side?  The lane-per-read mapping for reads beyond a 64-row LDS tile
// needs the rows to come in as COLUMN slices: one LDS-DMA instruction = the same 16 bytes of 64 different rows (lane =
// row; the slice is 1 KiB contiguous in LDS, so a lane's ds_read_b128 of its own 16 bytes is conflict-free), a ring of
// slices per wave, DEPTH slices in flight with counted waits.  Each instruction touches 64 different cache lines, and
// the 8 instructions that cover a 128-byte line come ~DEPTH turns apart: does that stream from HBM?
// Rows of L bytes at stride L (reads back to back), tiles of 64 consecutive rows; every byte is read once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void waitvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// WIDE: slices per DMA group issued back to back for one row segment (1: 16 B per row per turn; 4: 64 B; 8: a whole line)
template <int DEPTH, int AUX, int WIDE>
__global__ __launch_bounds__(64) void stream_cols(const unsigned char *__restrict__ src, size_t n_tiles, int L, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NS = DEPTH * WIDE + 2 * WIDE; // ring slots
    const int lane = threadIdx.x;
    const int nsl = (L + 15) >> 4; // slices per row
    const int ngr = (nsl + WIDE - 1) / WIDE;
    uint32_t acc = 0;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const unsigned char *row = src + (t * 64 + lane) * (size_t)L;
        auto issue = [&](int g) {
#pragma unroll
            for (int u = 0; u < WIDE; ++u) {
                const int s = g * WIDE + u;
                __builtin_amdgcn_global_load_lds((gptr_t)(row + 16 * min(s, nsl - 1)), (lptr_t)(lds + ((g % (DEPTH + 2)) * WIDE + u) * 1024), 16, 0, AUX);
            }
        };
        for (int g = 0; g < DEPTH && g < ngr; ++g) issue(g);
        for (int g = 0; g < ngr; ++g) {
            if (g + DEPTH < ngr) {
                issue(g + DEPTH);
                waitvm<DEPTH * WIDE>();
            } else {
                waitvm<0>();
            }
#pragma unroll
            for (int u = 0; u < WIDE; ++u) {
                const v4u d = *reinterpret_cast<const v4u *>(lds + ((g % (DEPTH + 2)) * WIDE + u) * 1024 + lane * 16);
                acc += __builtin_amdgcn_sad_u8(d[0], d[1], 0) + __builtin_amdgcn_sad_u8(d[2], d[3], 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t total = 1000000000ull;
    unsigned char *d; uint32_t *o;
    CK(hipMalloc(&d, total + 65536)); CK(hipMalloc(&o, 4096));
    CK(hipMemset(d, 60, total + 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, double gb, auto launch) {
        for (int w = 0; w < 5; ++w) launch();
        CK(hipDeviceSynchronize());
        const int R = 10;
        CK(hipEventRecord(a));
        for (int r = 0; r < R; ++r) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
        printf("%-64s %.4f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3); fflush(stdout);
    };
    for (int L : {600, 1000, 2000, 4000}) {
        const size_t n_tiles = total / ((size_t)64 * L);
        const double gb = n_tiles * 64.0 * L / 1e9;
        for (int per_cu : {8, 16}) {
            char nm[128];
#define RUN(DEPTH, AUX, WIDE) \
            snprintf(nm, sizeof nm, "L %4d  %2d waves/CU  depth %d x %d slices  %s", L, per_cu, DEPTH, WIDE, AUX ? "nt" : "default"); \
            timeit(nm, gb, [&] { hipLaunchKernelGGL((stream_cols<DEPTH, AUX, WIDE>), dim3(256 * per_cu), dim3(64), (DEPTH * WIDE + 2 * WIDE) * 1024, 0, d, n_tiles, L, o); });
            RUN(8, 0, 1) RUN(8, 2, 1) RUN(4, 0, 4) RUN(2, 0, 8) RUN(2, 2, 8)
        }
    }
    return 0;
}
