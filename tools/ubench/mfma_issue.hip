// Single-wave-per-SIMD issue rate of v_mfma_f32_32x32x16_f16 vs the number of independent accumulator chains, with
// the accumulators in VGPRs (unified file) -- input for the policy kernel's schedule.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(0.5f - j * 0.01f); }
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = c + r;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    float s = 0;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 500;
    unsigned long long h[256];
#define RUN(CH) { hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost); double a = 0; for (int i = 0; i < 256; ++i) a += h[i]; \
    printf("%d independent chains: %.1f ticks per MFMA (1 wave per SIMD)\n", CH, a / 256 / (iters * 8.0 * CH)); }
    RUN(1) RUN(2) RUN(4) RUN(8)
    return 0;
}
