// Checks the operand / result lane layouts assumed for v_mfma_f32_32x32x16_f16 on gfx950:
//   A: lane l holds A[i = l&31][k = 8*(l>>5) + j], j = 0..7      B: lane l holds B[k = 8*(l>>5) + j][n = l&31]
//   D: lane l, reg r holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_layout.hip && ./mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* D) {  // A[32][16], B[16][32], D[32][32] row-major
    const int l = threadIdx.x;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = (_Float16)B[(8 * (l >> 5) + j) * 32 + (l & 31)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
    srand(1);
    for (auto& x : hA) x = (rand() % 17 - 8) / 8.0f;   // exactly representable in f16, asymmetric
    for (auto& x : hB) x = (rand() % 13 - 6) / 4.0f;
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + n]; ref[i * 32 + n] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
    printf("mfma_f32_32x32x16_f16 layout check: max |D - ref| = %g -> %s\n", err, err < 1e-3 ? "LAYOUT OK" : "LAYOUT MISMATCH");
    return err < 1e-3 ? 0 : 1;
}
