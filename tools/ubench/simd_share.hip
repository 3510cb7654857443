// Micro-benchmark (MI355X): does a wave's VALU work come out right while ANOTHER wave on the same SIMD runs matrix instructions?
// 512-thread workgroups: waves 0-3 (one per SIMD) are VICTIMS -- a fixed, data-dependent recurrence of packed / scalar f32 VALU
// instructions whose final values depend on every intermediate bit -- waves 4-7 (the second wave of each SIMD) are AGGRESSORS:
// idle (s_sleep), f16 MFMA 32x32x16 back to back, f32 MFMA, or VALU.  The victims' results are compared bit for bit with the same
// launch with idle aggressors.
//   hipcc --offload-arch=gfx950 -O3 -o bin/simd_share simd_share.hip && bin/simd_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int VICT, int AGGR>
__global__ void __launch_bounds__(512) k(float* out, float* sink, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        const int gid = (blockIdx.x * 4 + wave) * 64 + lane;
        float seed = 1.0f + 1e-3f * (float)(gid % 977);
        if (VICT == 0) {   // packed f32 fma / mul-clamp chains (what the residual block's output layer is made of)
            f32x2 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = f32x2{seed + 0.01f * j, seed - 0.02f * j};
            const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_elementwise_fma(a[j], m, c);
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(a[j]) : "v"(a[j]), "v"(m));
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = a[j] + f32x2{0.25f, 0.5f};
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += a[j].x * 3.0f + a[j].y;
            out[gid] = s;
        } else if (VICT == 1) {   // scalar f32 fma chains
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = seed + 0.01f * j;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], 1.0000001f, 1e-7f * (j + 1));
#pragma unroll
                for (int j = 0; j < 16; ++j) a[j] = fminf(a[j], 8.0f) - 0.5f * (a[j] > 4.0f ? 1.0f : 0.0f);
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += a[j] * (j + 1);
            out[gid] = s;
        } else if (VICT == 3 || VICT == 4) {   // packed fma with a CROSSED source (op_sel:[0,1,0] op_sel_hi:[1,0,1]) -- 4: the same, uncrossed
            f32x2 a[8], x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = f32x2{seed + 0.01f * j, seed - 0.02f * j}; x[j] = f32x2{0.5f + 0.001f * j, 0.25f - 0.001f * j}; }
            const f32x2 m = {1.0000001f, 0.9999999f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (VICT == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[j]) : "v"(x[j]), "v"(m));
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(x[j]), "v"(m));
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = x[j] * f32x2{0.999f, 1.001f};
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += a[j].x * 3.0f + a[j].y;
            out[gid] = s;
        } else {   // v_permlane32_swap + adds (the tile exchange)
            unsigned a = __float_as_uint(seed), b = a ^ 0x1234567u;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
                    a = r[0] * 1664525u + 1013904223u;
                    b = r[1] ^ (a >> 7);
                }
            }
            out[gid] = __uint_as_float((a ^ b) & 0x3fffffffu);
        }
    } else {
        if (AGGR == 0) {
            for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
        } else if (AGGR == 1 || AGGR == 3) {   // f16 matrix instructions, four accumulators, back to back (3: with s_nop gaps like a mixed kernel)
            f32x16 acc[4] = {};
            f16x8 av, bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) { av[j] = (_Float16)(0.001f * (lane + j)); bv[j] = (_Float16)(0.5f - 0.001f * j); }
            for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[j], 0, 0, 0);
                    if (AGGR == 3) asm volatile("s_nop 7");
                }
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
            if (s == 12345.678f) sink[0] = s;
        } else if (AGGR == 2) {   // f32 matrix instruction (runs on the f32 vector ALUs)
            f32x16 acc[4] = {};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.001f * lane, 0.5f, acc[j], 0, 0, 0);
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
            if (s == 12345.678f) sink[0] = s;
        } else {   // VALU aggressor: packed fma
            f32x2 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = f32x2{1.0f + j, 2.0f - j};
            for (int it = 0; it < iters * 3; ++it) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_elementwise_fma(a[j], f32x2{1.0000001f, 0.9999999f}, f32x2{1e-7f, -1e-7f});
            }
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += a[j].x + a[j].y;
            if (s == 12345.678f) sink[0] = s;
        }
    }
}

template <int VICT, int AGGR>
static void run(const char* vn, const char* an, float* out, float* sink, std::vector<float>& ref, bool is_ref, int iters, int reps) {
    const int blocks = 256, n = blocks * 256;
    std::vector<float> h(n);
    unsigned long long bad = 0, q[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((k<VICT, AGGR>), dim3(blocks), dim3(512), 0, 0, out, sink, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        if (is_ref && r == 0) { ref = h; continue; }
        for (int i = 0; i < n; ++i)
            if (memcmp(&h[i], &ref[i], 4)) { ++bad; ++q[(i & 63) >> 4]; }
    }
    printf("victim %-22s aggressor %-28s: %llu differing lane-results of %llu  (by lane quarter: %llu %llu %llu %llu)\n", vn, an, bad,
           (unsigned long long)n * (reps - (is_ref ? 1 : 0)), q[0], q[1], q[2], q[3]);
}

int main() {
    float *out, *sink;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&sink, 4);
    const int iters = 3000, reps = 20;
    std::vector<float> ref;
#define ROW(V, VN)                                                                                   \
    run<V, 0>(VN, "idle (reference, repeated)", out, sink, ref, true, iters, reps);                  \
    run<V, 1>(VN, "f16 MFMA 32x32x16 back to back", out, sink, ref, false, iters, reps);             \
    run<V, 3>(VN, "f16 MFMA with gaps", out, sink, ref, false, iters, reps);                         \
    run<V, 2>(VN, "f32 MFMA 32x32x2", out, sink, ref, false, iters, reps);                           \
    run<V, 4>(VN, "packed f32 fma", out, sink, ref, false, iters, reps);
    ROW(0, "pk_fma/pk_mul clamp/pk_add")
    ROW(1, "scalar fma/min/cmp")
    ROW(2, "permlane32_swap + int")
    ROW(3, "pk_fma CROSSED op_sel")
    ROW(4, "pk_fma plain (asm)")
    return 0;
}
