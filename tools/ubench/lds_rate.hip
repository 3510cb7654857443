// Microbenchmark (MI355X): LDS read throughput per CU for the three read forms the PPO gradient kernel uses, with its own
// address patterns:  b128 = ds_read_b128, lane-linear (forward operand images);  b64 = ds_read_b64, lane-linear;
// tr = ds_read_b64_tr_b16 with the exchange area's lane constant (ex_lane_const) / the swizzled image's (tr_lane_hidden).
//   hipcc --offload-arch=gfx950 -O3 -o bin/lds_rate lds_rate.hip && bin/lds_rate
// One workgroup per CU (LDS 64 KB+), W waves, every wave issues kIters x 12 reads (12 in flight, then one wait), different offsets.
// Prints bytes per cycle per CU for W = 1, 2, 4, 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned ex_lane_const(int lane, int second) {
    const int a = (lane >> 4) & 1, b = lane & 1, hh = lane >> 5, k = (lane >> 2) & 3, mhi = (lane >> 1) & 1;
    return 16u * (unsigned)(64 * a + 32 * b + 4 * (hh ^ b) + 8 * (second ^ a) + k) + 8u * (unsigned)mhi;
}
__device__ __forceinline__ unsigned tr_lane_plain(int lane) {   // un-swizzled: 4-way bank conflicts expected
    const int a = (lane >> 4) & 1, b = lane & 1, hh = lane >> 5, k = (lane >> 2) & 3, mhi = (lane >> 1) & 1;
    return 16u * (unsigned)(64 * a + 32 * b + 4 * hh + k) + 8u * (unsigned)mhi;
}

constexpr int kIters = 2000;

// MODE 0 b128 linear, 1 b64 linear, 2 tr (exchange pattern, swizzled), 3 tr (plain pattern)
template <int MODE>
__global__ void __launch_bounds__(512) rate_kernel(unsigned long long* __restrict__ out, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned base = (unsigned)(size_t)smem + 2048u * (unsigned)(wave & 3);
    if (MODE == 0) base += 16u * lane;
    if (MODE == 1) base += 8u * lane;
    if (MODE == 2) base += ex_lane_const(lane, wave >> 2);
    if (MODE == 3) base += tr_lane_plain(lane);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
        if (MODE == 0) {
            u32x4 v[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(base), "n"(q * 4096 + 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 12; ++q) acc ^= v[q];
        } else {
            u32x2 v[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                if (MODE == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[q]) : "v"(base), "n"(q * 4096 + 512));
                else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v[q]) : "v"(base), "n"(q * 4096 + (q & 1) * 256));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 12; ++q) { acc.x ^= v[q].x; acc.y ^= v[q].y; }
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
}

template <int MODE>
void run(const char* name, int bytes_per_lane) {
    unsigned long long* d_out; unsigned* d_sink;
    CK(hipMalloc(&d_out, 256 * 8 * 8)); CK(hipMalloc(&d_sink, 4));
    CK(hipFuncSetAttribute((const void*)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    printf("%-28s", name);
    for (int W : {1, 2, 4, 8}) {
        unsigned long long h[256 * 8];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(64 * W), 96 * 1024, 0, d_out, d_sink);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        double worst = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < W; ++w) if ((double)h[b * 8 + w] > worst) worst = (double)h[b * 8 + w];
        const double bytes = (double)W * kIters * 12 * 64 * bytes_per_lane;
        printf("  W=%d: %6.1f B/clk (%5.1f clk/read/wave)", W, bytes / worst, worst / (kIters * 12.0));
    }
    printf("\n");
}

int main() {
    run<0>("ds_read_b128 linear", 16);
    run<1>("ds_read_b64 linear", 8);
    run<2>("ds_read_b64_tr_b16 exchange", 8);
    run<3>("ds_read_b64_tr_b16 plain", 8);
    return 0;
}
