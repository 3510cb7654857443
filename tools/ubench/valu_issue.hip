// Microbenchmark (MI355X): single-wave VALU issue rate, with and without interleaved LDS broadcast reads.
// hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters, const float* wsrc) {
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(wsrc[threadIdx.x & 63], 1.0f, 0.5f, 0.25f);
    __syncthreads();
    float h[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) h[j] = threadIdx.x * 0.001f + j;
    float x = out[threadIdx.x];
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 64 independent FMAs, VGPR operands
#pragma unroll
            for (int j = 0; j < 64; ++j) h[j] = fmaf(h[j], x, 1.0f);
        } else if (MODE == 1) {  // 16 broadcast ds_read_b128 + 64 FMAs using them
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 w = lds[(j + it) & 255];
                h[4 * j + 0] = fmaf(w.x, x, h[4 * j + 0]);
                h[4 * j + 1] = fmaf(w.y, x, h[4 * j + 1]);
                h[4 * j + 2] = fmaf(w.z, x, h[4 * j + 2]);
                h[4 * j + 3] = fmaf(w.w, x, h[4 * j + 3]);
            }
        } else if (MODE == 2) {  // dependent chain
#pragma unroll
            for (int j = 0; j < 64; ++j) h[0] = fmaf(h[0], x, 1.0f);
        } else if (MODE == 3) {  // 64 independent v_mul+v_add pairs (no fma)
#pragma unroll
            for (int j = 0; j < 64; ++j) h[j] = h[j] * x;
        }
    }
    unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) s += h[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 200;
    float *out, *w; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4 * 4); hipMalloc(&w, 64 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    hipMemset(out, 0, blocks * 256 * 4); hipMemset(w, 0, 256);
    const char* names[] = {"64 indep v_fma (VGPR)", "16 ds_read_b128 bcast + 64 v_fma", "64 dependent v_fma", "64 indep v_mul"};
    for (int wpb = 1; wpb <= 4; wpb *= 2) {  // waves per SIMD: grid 256*wpb blocks of 256 threads
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<unsigned long long> h(blocks * wpb);
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks * wpb), dim3(256), 0, 0, out, cyc, iters, w);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks * wpb), dim3(256), 0, 0, out, cyc, iters, w);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks * wpb), dim3(256), 0, 0, out, cyc, iters, w);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks * wpb), dim3(256), 0, 0, out, cyc, iters, w);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
            printf("blocks/CU=%d  %-36s : %.2f clock64 ticks per VALU instr per wave\n", wpb, names[mode], avg / (iters * 64.0));
        }
    }
    // wall-clock calibration of clock64
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, 20000, w); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %llu ticks in %.3f ms -> clock64 runs at %.1f MHz\n", c, ms, c / (ms * 1e3));
    return 0;
}
