// Microbenchmark (MI355X): what bounds a read-streaming kernel of phase B's shape (quadrace_ppo.hip)?
//
//   hipcc --offload-arch=gfx950 -O3 -o bin/stream_read stream_read.hip && bin/stream_read
//
// One-wave workgroups (like phase B) read a buffer that was written by the PREVIOUS kernel (so nothing is L2-resident at
// kernel start: L2 is invalidated at kernel boundaries) in 1 KB wave-loads (64 lanes x 16 B), DEPTH loads in flight per wave.
// REUSE waves of the same XCD (workgroup ids congruent mod 8) read the SAME region: the first touch misses to HBM / the
// memory-side cache, the others hit the XCD's L2 -- unique bytes fixed at 54.6 MB (phase B's scratch), L2->CU bytes = REUSE x that.
// Prints time and bandwidths for a sweep of (waves per CU, DEPTH, REUSE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) fill_kernel(float4* p, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// region r = [r * per_region, (r + 1) * per_region) float4 rows of 64
template <int DEPTH>
__global__ void __launch_bounds__(64) read_kernel(const float4* __restrict__ src, float* __restrict__ sink, int reuse, int rows_per_region) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int region = xcd + 8 * (slot / reuse);
    const float4* p = src + (size_t)region * rows_per_region * 64 + threadIdx.x;
    float acc = 0.0f;
    for (int r0 = 0; r0 < rows_per_region; r0 += DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) v[j] = p[(size_t)(r0 + j < rows_per_region ? r0 + j : rows_per_region - 1) * 64];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) acc += v[j].x + v[j].w;
    }
    if (acc == 12345.0f) sink[id] = acc;
}

template <int DEPTH>
double run(const float4* buf, size_t n4, float* sink, int waves, int reuse, hipStream_t st) {
    const int regions = waves / reuse;
    const int rows = (int)(n4 / 64 / regions);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, st, const_cast<float4*>(buf), n4);   // producer kernel before every read
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_kernel<DEPTH>, dim3(waves), dim3(64), 0, st, buf, sink, reuse, rows);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms * 1e3 < best) best = ms * 1e3;
    }
    return best;
}

int main() {
    const size_t bytes = 54600000 / 65536 * 65536;
    const size_t n4 = bytes / 16;
    float4* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4 * 65536));
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("{\"unique_MB\": %.1f, \"runs\": [", bytes / 1e6);
    bool first = true;
    const int wpc[] = {1, 2, 3, 4, 6, 8};
    for (int reuse : {1, 2, 4})
        for (int w : wpc)
            for (int depth : {8, 16, 32}) {
                const int waves = 256 * w;
                if (waves % (8 * reuse)) continue;
                double us = depth == 8 ? run<8>(buf, n4, sink, waves, reuse, st) : depth == 16 ? run<16>(buf, n4, sink, waves, reuse, st)
                                                                                                 : run<32>(buf, n4, sink, waves, reuse, st);
                printf("%s{\"waves_per_cu\": %d, \"KB_in_flight_per_wave\": %d, \"reuse\": %d, \"us\": %.2f, \"unique_TBps\": %.2f, \"l2_to_cu_TBps\": %.2f}",
                       first ? "" : ", ", w, depth, reuse, us, bytes / us / 1e6, bytes * (double)reuse / us / 1e6);
                first = false;
            }
    printf("]}\n");
    return 0;
}
