// Microbenchmark (MI355X): cost of reducing per-workgroup 32x32 f32 tiles into a shared gradient buffer with f32 atomics.
//
//   hipcc --offload-arch=gfx950 -O3 -o bin/l2_atomics l2_atomics.hip && bin/l2_atomics
//
// Shape = the fused PPO gradient kernel: 256 workgroups x 4 waves, every wave adds 10 tiles x 16 registers x 64 lanes
// (= 40 960 floats per workgroup, the same 160 KB of addresses from every workgroup).
//   agent      global_atomic_add_f32 ... sc1   one buffer, agent scope (what atomicAdd() compiles to): executed memory-side
//   xcd        global_atomic_add_f32 (no scope bits) into the copy of THIS XCD (s_getreg XCC_ID): executed in the XCD's L2,
//              8 copies, made visible by the end-of-kernel write-back; the consumer sums 8 copies
//   store      plain per-workgroup partial stores (256 x 160 KB = 41 MB) for comparison
// Checks that the xcd variant adds up exactly (integers in f32) and prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kTilesPerWave = 10, kWaves = 4, kElems = kWaves * kTilesPerWave * 1024;  // 40 960

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

// MODE 0 agent-scope atomics, 1 XCD-local atomics, 2 plain stores of per-workgroup partials, 3 nothing (launch floor)
template <int MODE>
__global__ void __launch_bounds__(256) reduce_kernel(float* __restrict__ buf, int* __restrict__ xcd_seen) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = buf;
    if (MODE == 1 || MODE == 4) {
        const int x = xcc_id();
        base = buf + (size_t)x * kElems;
        if (threadIdx.x == 0) xcd_seen[blockIdx.x] = x;
    }
    if (MODE == 2) base = buf + (size_t)blockIdx.x * kElems;
#pragma unroll
    for (int t = 0; t < kTilesPerWave; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // MODE 4: every workgroup starts at a different tile (rotation by workgroup id), so that the 32 workgroups of an XCD
            // do not all hit the same cache lines at the same moment
            const int tt = MODE == 4 ? (wave * kTilesPerWave + t + (int)(blockIdx.x >> 3) * 5) % (kWaves * kTilesPerWave) : wave * kTilesPerWave + t;
            float* p = base + (tt * 16 + r) * 64 + lane;
            const float v = 1.0f;
            if (MODE == 0) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            if (MODE == 1 || MODE == 4) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            if (MODE == 2) __builtin_nontemporal_store(v, p);
        }
}

template <int MODE>
double time_us(float* buf, int* seen, hipStream_t st, int K) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(reduce_kernel<MODE>, dim3(256), dim3(256), 0, st, buf, seen);
    CK(hipStreamSynchronize(st));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(reduce_kernel<MODE>, dim3(256), dim3(256), 0, st, buf, seen);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e3 / K < best) best = ms * 1e3 / K;
    }
    return best;
}

int main() {
    float* buf; int* seen;
    CK(hipMalloc(&buf, sizeof(float) * (size_t)kElems * 256));
    CK(hipMalloc(&seen, sizeof(int) * 256));
    hipStream_t st; CK(hipStreamCreate(&st));
    // correctness of the XCD-local form: one launch, then sum the 8 copies on the host
    CK(hipMemset(buf, 0, sizeof(float) * (size_t)kElems * 8));
    hipLaunchKernelGGL(reduce_kernel<1>, dim3(256), dim3(256), 0, st, buf, seen);
    CK(hipStreamSynchronize(st));
    std::vector<float> h((size_t)kElems * 8);
    std::vector<int> hs(256);
    CK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), seen, 256 * 4, hipMemcpyDeviceToHost));
    int per_xcd[8] = {0}, bad = 0, rr_match = 0;
    for (int b = 0; b < 256; ++b) { per_xcd[hs[b]]++; rr_match += hs[b] == b % 8; }
    for (int i = 0; i < kElems; ++i) {
        float s = 0;
        for (int x = 0; x < 8; ++x) s += h[(size_t)x * kElems + i];
        bad += s != 256.0f;
    }
    int copy_ok = 1;
    for (int x = 0; x < 8; ++x) copy_ok &= h[(size_t)x * kElems] == (float)per_xcd[x];
    const double t_floor = time_us<3>(buf, seen, st, 500);
    const double t_agent = time_us<0>(buf, seen, st, 200);
    const double t_xcd = time_us<1>(buf, seen, st, 500);
    const double t_store = time_us<2>(buf, seen, st, 500);
    const double t_rot = time_us<4>(buf, seen, st, 500);
    printf("{\"elems_per_wg\": %d, \"wgs\": 256, \"xcd_sum_errors\": %d, \"copies_match_wg_counts\": %d, \"wgs_per_xcd\": [%d,%d,%d,%d,%d,%d,%d,%d], "
           "\"wg_id_mod8_is_xcd\": %d, \"us\": {\"empty\": %.2f, \"agent_atomics\": %.2f, \"xcd_local_atomics\": %.2f, \"partial_stores_41MB\": %.2f, \"xcd_local_atomics_rotated\": %.2f}}\n",
           kElems, bad, copy_ok, per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7], rr_match,
           t_floor, t_agent, t_xcd, t_store, t_rot);
    return 0;
}
