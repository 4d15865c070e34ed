// Probe (diagnostic, not product): which k-order do the A/B fragments of
// v_mfma_i32_32x32x32_i8 use on gfx950, and is the C/D map the documented one?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t* A, const int8_t* B, int* D, int layout) {
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v4i v; int8_t b[16]; } a, b;
    for (int j = 0; j < 16; ++j) {
        int kk = layout == 0 ? 16 * h + j : (j < 8 ? 8 * h + j : 16 + 8 * h + (j - 8));
        a.b[j] = A[r * 32 + kk];   // A[m=r][k]
        b.b[j] = B[kk * 32 + r];   // B[k][n=r]
    }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r] = c[reg];
    }
}
int main() {
    std::vector<int8_t> A(1024), B(1024); std::vector<int> ref(1024), D(1024);
    srand(1); for (auto& x : A) x = rand() % 7 - 3; for (auto& x : B) x = rand() % 7 - 3;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int q = 0; q < 32; ++q) s += A[m*32+q] * B[q*32+n]; ref[m*32+n] = s; }
    int8_t *dA, *dB; int* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    for (int layout = 0; layout < 2; ++layout) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, layout);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != ref[i];
        printf("layout %d (%s): %d mismatches\n", layout, layout == 0 ? "k = 16h + j" : "k = 8h + j | 16 + 8h + j-8", bad);
    }
    return 0;
}
