// Microbenchmark (MI355X): what does a kernel of the step kernel's SHAPE cost when it does no arithmetic?
//
//   hipcc --offload-arch=gfx950 -O3 -o bin/launch_floor launch_floor.hip && bin/launch_floor [n_envs]
//
// K back-to-back launches on one stream (the calling pattern of qr_step_launches), timed with hipEvents:
//   empty      256-thread workgroups, no memory traffic               -> launch + dispatch + completion floor
//   copy       per lane: 7 x 16 B loads (112 B) + 11 x 16 B stores (176 B) = the E2E step kernel's 285 B/env
//              algorithmic traffic with the same planar float4 layout, fresh output rows per launch
//   copy_nt    same, stores marked non-temporal (streaming)
//   copy_wt    same, stores written through (sc0 sc1): no dirty L2 lines left for the end-of-kernel write-back
//   *_b128/_b64  128- / 64-thread workgroups (2 / 4 workgroups per CU at N = 65 536)
//   *_graph    the K launches captured into one hipGraph and replayed;  empty_1wg: a single workgroup
//   chain_X    copy + a dependent chain of X fmaf between the loads and the stores (one wave per SIMD cannot hide it)
// Prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kLoads = 7, kStores = 11;

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) empty_kernel(int n, float* sink) {
    if (n < 0) sink[threadIdx.x] = 1.0f;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 as_vec(float4 v) { f32x4 r = {v.x, v.y, v.z, v.w}; return r; }
__device__ __forceinline__ void store_wt(float4* p, float4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(as_vec(v)) : "memory");
}
__device__ __forceinline__ void store_nt(float4* p, float4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(as_vec(v)) : "memory");
}
__device__ __forceinline__ void store_nt_wt(float4* p, float4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(as_vec(v)) : "memory");
}

// MODE 0 plain, 1 nt, 2 write-through, 3 nt + write-through
template <int BLOCK, int MODE, int CHAIN>
__global__ void __launch_bounds__(BLOCK) copy_kernel(int n, int stride, const float4* __restrict__ in,
                                                     float4* __restrict__ state_out, float4* __restrict__ out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 v[kLoads];
#pragma unroll
    for (int p = 0; p < kLoads; ++p) v[p] = in[(size_t)p * stride + i];
    float acc = v[0].x;
#pragma unroll 16
    for (int c = 0; c < CHAIN; ++c) acc = fmaf(acc, v[1].y, v[2].z);
    v[0].x = acc;
#pragma unroll
    for (int p = 0; p < kStores; ++p) {
        // first 4 planes: the state written back in place (same lines every launch); the rest: fresh rollout rows
        float4* dst = (p < 4) ? state_out + (size_t)p * stride + i : out + (size_t)(p - 4) * stride + i;
        const float4 val = v[p % kLoads];
        if (MODE == 0) *dst = val;
        else if (MODE == 1) store_nt(dst, val);
        else if (MODE == 2) store_wt(dst, val);
        else store_nt_wt(dst, val);
    }
}

struct Ctx {
    int n, stride, K;
    float4 *in, *state, *out;
    hipStream_t st;
    hipEvent_t e0, e1;
};

template <typename F>
double time_us(Ctx& c, F launch) {
    for (int k = 0; k < 20; ++k) launch(k);
    CK(hipStreamSynchronize(c.st));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(c.e0, c.st));
        for (int k = 0; k < c.K; ++k) launch(k);
        CK(hipEventRecord(c.e1, c.st));
        CK(hipEventSynchronize(c.e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, c.e0, c.e1));
        if (ms * 1e3 / c.K < best) best = ms * 1e3 / c.K;
    }
    return best;
}

// the same back-to-back launches captured once into a hipGraph and replayed (does the graph path shorten the gap?)
template <typename F>
double time_graph_us(Ctx& c, F launch) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(c.st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < c.K; ++k) launch(k);
    CK(hipStreamEndCapture(c.st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, c.st));
    CK(hipStreamSynchronize(c.st));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(c.e0, c.st));
        CK(hipGraphLaunch(ge, c.st));
        CK(hipEventRecord(c.e1, c.st));
        CK(hipEventSynchronize(c.e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, c.e0, c.e1));
        if (ms * 1e3 / c.K < best) best = ms * 1e3 / c.K;
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return best;
}

template <int BLOCK, int MODE, int CHAIN>
double run_copy(Ctx& c) {
    const int grid = (c.n + BLOCK - 1) / BLOCK;
    return time_us(c, [&](int k) {
        hipLaunchKernelGGL((copy_kernel<BLOCK, MODE, CHAIN>), dim3(grid), dim3(BLOCK), 0, c.st, c.n, c.stride, c.in, c.state,
                           c.out + (size_t)(k % 64) * (kStores - 4) * c.stride);
    });
}

template <int BLOCK>
double run_empty(Ctx& c) {
    const int grid = (c.n + BLOCK - 1) / BLOCK;
    return time_us(c, [&](int) { hipLaunchKernelGGL((empty_kernel<BLOCK>), dim3(grid), dim3(BLOCK), 0, c.st, c.n, (float*)c.out); });
}

int main(int argc, char** argv) {
    Ctx c;
    c.n = argc > 1 ? atoi(argv[1]) : 65536;
    c.stride = (c.n + 255) / 256 * 256;
    c.K = 1000;
    CK(hipMalloc(&c.in, sizeof(float4) * (size_t)c.stride * kLoads));
    CK(hipMalloc(&c.state, sizeof(float4) * (size_t)c.stride * 4));
    CK(hipMalloc(&c.out, sizeof(float4) * (size_t)c.stride * (kStores - 4) * 64));
    CK(hipMemset(c.in, 0, sizeof(float4) * (size_t)c.stride * kLoads));
    CK(hipStreamCreate(&c.st));
    CK(hipEventCreate(&c.e0));
    CK(hipEventCreate(&c.e1));
    const double bytes = (double)c.n * 16.0 * (kLoads + kStores);
    std::vector<std::pair<std::string, double>> r;
    r.push_back({"empty_b256", run_empty<256>(c)});
    r.push_back({"empty_b128", run_empty<128>(c)});
    r.push_back({"empty_b64", run_empty<64>(c)});
    r.push_back({"copy_b256", run_copy<256, 0, 0>(c)});
    r.push_back({"copy_b128", run_copy<128, 0, 0>(c)});
    r.push_back({"copy_b64", run_copy<64, 0, 0>(c)});
    r.push_back({"copy_nt_b256", run_copy<256, 1, 0>(c)});
    r.push_back({"copy_wt_b256", run_copy<256, 2, 0>(c)});
    r.push_back({"copy_ntwt_b256", run_copy<256, 3, 0>(c)});
    r.push_back({"copy_nt_b128", run_copy<128, 1, 0>(c)});
    r.push_back({"copy_wt_b128", run_copy<128, 2, 0>(c)});
    {
        const int grid = (c.n + 255) / 256;
        r.push_back({"empty_b256_graph", time_graph_us(c, [&](int) {
            hipLaunchKernelGGL((empty_kernel<256>), dim3(grid), dim3(256), 0, c.st, c.n, (float*)c.out); })});
        r.push_back({"copy_nt_b256_graph", time_graph_us(c, [&](int k) {
            hipLaunchKernelGGL((copy_kernel<256, 1, 0>), dim3(grid), dim3(256), 0, c.st, c.n, c.stride, c.in, c.state,
                               c.out + (size_t)(k % 64) * (kStores - 4) * c.stride); })});
        // one workgroup only: the pure inter-kernel dependency cost without the dispatch of 256 workgroups
        r.push_back({"empty_1wg", time_us(c, [&](int) {
            hipLaunchKernelGGL((empty_kernel<256>), dim3(1), dim3(256), 0, c.st, c.n, (float*)c.out); })});
    }
    r.push_back({"chain512_b256", run_copy<256, 0, 512>(c)});
    r.push_back({"chain2048_b256", run_copy<256, 0, 2048>(c)});
    r.push_back({"chain2048_wt_b256", run_copy<256, 2, 2048>(c)});
    r.push_back({"chain2048_nt_b256", run_copy<256, 1, 2048>(c)});
    printf("{\"n_envs\": %d, \"launches\": %d, \"bytes_per_launch\": %.0f, \"us_per_launch\": {", c.n, c.K, bytes);
    for (size_t i = 0; i < r.size(); ++i) printf("%s\"%s\": %.3f", i ? ", " : "", r[i].first.c_str(), r[i].second);
    printf("}, \"GBps\": {");
    bool first = true;
    for (auto& kv : r)
        if (kv.first.rfind("empty", 0) != 0) {
            printf("%s\"%s\": %.1f", first ? "" : ", ", kv.first.c_str(), bytes / kv.second / 1e3);
            first = false;
        }
    printf("}}\n");
    return 0;
}
