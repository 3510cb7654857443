// Microbenchmark (MI355X): VALU instruction throughput per SIMD versus resident waves -- is a plain f32 wave64 VALU op a
// 2-cycle or a 4-cycle instruction, does v_pk_fma_f32 cost the same as v_fma_f32, and what does an f32 MFMA in ANOTHER wave of
// the same SIMD do to it?   hipcc --offload-arch=gfx950 -O3 -o bin/valu_rate valu_rate.hip && bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    float h[16];
    f32x2 p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { h[j] = threadIdx.x * 0.001f + j; p[j] = f32x2{h[j], h[j] + 1.0f}; }
    float x = 1.0f + out[threadIdx.x] * 1e-9f;
    unsigned long long msk = __builtin_amdgcn_read_exec() >> 7;
    f32x2 xx = {x, x};
    f32x16 acc = {0};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: role branches below are uniform
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 a8, b8;
    for (int q = 0; q < 8; ++q) { a8[q] = (__bf16)(0.01f * q + threadIdx.x); b8[q] = (__bf16)(0.02f * q + x); }
    asm volatile("" : "+v"(a8), "+v"(b8));
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(h[j]) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 1) {
#define S(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[j]) : "v"(xx));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 2) {
#define S(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(h[j]) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 3) {
#define S(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(h[j]) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 4) {
#define S(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(h[j]) : "v"(x) : );
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 5) {   // f32 MFMA only, dependent chain on one accumulator (4 per 64 "instructions" so that iters match)
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, h[0], acc, 0, 0, 0);
        } else if (MODE == 6) {   // waves 0,1 of the workgroup: f32 MFMA stream; waves 2,3: v_fma stream (use with 8-wave... see host)
            if (wave & 1) {
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, h[0], acc, 0, 0, 0);
            } else {
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(h[j]) : "v"(x));
                REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
            }
        } else if (MODE == 7) {   // same wave: one f32 MFMA followed by 15 independent v_fma (does VALU hide under an f32 MFMA?)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, h[0], acc, 0, 0, 0);
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(h[j]) : "v"(x));
            S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#undef S
        } else if (MODE == 8) {   // same wave: one bf16 MFMA (32x32x16, 8 passes) followed by 7 independent v_fma
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(h[j]) : "v"(x));
            S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#undef S
        } else if (MODE == 11) {  // bf16 MFMA only, dependent chain (per MFMA)
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        } else if (MODE == 12) {  // odd waves bf16 MFMA stream / even waves v_fma
            if (wave & 1) {
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            } else {
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(h[j]) : "v"(x));
                REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
            }
        } else if (MODE == 13) {  // v_cndmask e64 with an SGPR-pair mask
#define S(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(h[j]) : "v"(x), "s"(msk));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 14) {  // v_cmp + v_cndmask pairs (vcc written, then read)
#define S(j) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(h[j]) : "v"(x) : "vcc");
            REP16(S) REP16(S)
#undef S
        } else if (MODE == 15) {  // 1 f32 MFMA then 15 v_fma, INDEPENDENT accumulators over 2 MFMAs (no dep stall)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, h[0], acc, 0, 0, 0);
        } else if (MODE == 16) {  // integer multiplies (Philox): low half
#define S(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(h[j]) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 17) {  // high half
#define S(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(h[j]) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 18) {  // both halves in one instruction
#define S(j) asm volatile("v_mad_u64_u32 %0, %2, %1, %3, 0" : "+v"(p[j]), "+v"(h[j]), "=s"(msk) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 9) {   // dependent v_fma chain
#define S(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(h[0]) : "v"(x));
            REP16(S) REP16(S) REP16(S) REP16(S)
#undef S
        } else if (MODE == 10) {  // permlane32_swap pairs
#define S(j) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(h[j]), "+v"(h[(j + 1) & 15]));
            S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14) S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14)
            S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14) S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14)
            S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14) S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14)
            S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14) S(0) S(2) S(4) S(6) S(8) S(10) S(12) S(14)
#undef S
        }
    }
    unsigned long long t1 = clock64();
    float s = acc[0] + acc[5];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += h[j] + p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc, double per_iter) {
    const int iters = 400;
    printf("%-58s", name);
    for (int wpb = 1; wpb <= 4; wpb *= 2) {
        const int blocks = 256 * wpb;
        std::vector<unsigned long long> h(blocks * 4);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double even = 0, odd = 0;
        for (int i = 0; i < blocks * 4; ++i) ((i & 1) ? odd : even) += h[i];
        even /= blocks * 2; odd /= blocks * 2;
        if (MODE == 6 || MODE == 12) printf(" | w/SIMD=%d: fma-waves %.2f cyc/instr, mfma-waves %.1f cyc/mfma", wpb, even / (iters * per_iter), odd / (iters * 4.0));
        else printf(" | w/SIMD=%d: %.2f", wpb, (even + odd) / 2 / (iters * per_iter));
    }
    printf("\n");
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 4 * 8);
    hipMemset(out, 0, 1024 * 256 * 4);
    printf("clock64 ticks per instruction per WAVE (x waves/SIMD = SIMD cycles per instruction at saturation)\n");
    run<0>("64 indep v_fma_f32", out, cyc, 64);
    run<1>("64 indep v_pk_fma_f32", out, cyc, 64);
    run<2>("64 indep v_mul_f32", out, cyc, 64);
    run<3>("64 indep v_add_u32", out, cyc, 64);
    run<4>("64 indep v_cndmask_b32", out, cyc, 64);
    run<16>("64 indep v_mul_lo_u32", out, cyc, 64);
    run<17>("64 indep v_mul_hi_u32", out, cyc, 64);
    run<18>("64 indep v_mad_u64_u32", out, cyc, 64);
    run<9>("64 DEPENDENT v_fma_f32", out, cyc, 64);
    run<10>("64 v_permlane32_swap", out, cyc, 64);
    run<5>("4 dependent v_mfma_f32_32x32x2_f32 (per MFMA)", out, cyc, 4);
    run<6>("odd waves f32 MFMA / even waves v_fma, same SIMDs?", out, cyc, 64);
    run<11>("4 dependent v_mfma_f32_32x32x16_bf16 (per MFMA)", out, cyc, 4);
    run<13>("64 indep v_cndmask_b32_e64 (SGPR mask)", out, cyc, 64);
    run<14>("32 x (v_cmp vcc ; v_cndmask vcc) per pair", out, cyc, 32);
    run<12>("odd waves bf16 MFMA / even waves v_fma", out, cyc, 64);
    run<7>("1 f32 MFMA + 15 v_fma in ONE wave (per group of 16)", out, cyc, 1);
    run<8>("1 bf16 MFMA 32x32x16 + 7 v_fma in ONE wave (per group of 8)", out, cyc, 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %llu ticks in %.3f ms -> clock64 runs at %.1f MHz\n", c, ms, c / (ms * 1e3));
    return 0;
}
