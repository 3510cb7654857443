// Microbenchmark (MI355X), round 6: what does a SECOND instruction stream on a SIMD buy at 64 envs per SIMD?
//   (a) half-EXEC: does a wave whose EXEC holds only the low 32 lanes issue faster (v_fma_f32, v_pk_fma_f32, f16 MFMA 32x32x16,
//       ds_write_b128)?  If not, "32 envs per wave, two waves per SIMD" moves the same 64 envs per ~5 ticks as one full wave.
//   (b) wave-specialised pair: one wave per SIMD running a step-like stream (NA VALU + 10 f16 MFMA + NB VALU per iteration) against
//       512-thread workgroups in which waves 0-3 run the NA part and waves 4-7 (the second wave of each SIMD) the MFMA + NB part,
//       handing values over through LDS with two s_barrier per iteration; NA is split into a part that DEPENDS on the partner's
//       result and a part that does not (runs in the partner's shadow).
//   (c) where the waves of a 512-thread workgroup land: HW_ID of every wave (SIMD id, CU id) -- waves w and w + 4 on one SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o bin/pair_issue pair_issue.hip && bin/pair_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static void ck(const char* w);
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP16(S) REP8(S) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

// ---------------------------------------------------------------- (a) half EXEC
template <int MODE, bool HALF>
__global__ void __launch_bounds__(256) half_k(float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[256 * 4];
    float h[16];
    f32x2 p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { h[j] = threadIdx.x * 0.001f + j; p[j] = f32x2{h[j], h[j] + 1.0f}; }
    float x = 1.0f + out[threadIdx.x] * 1e-9f;
    f32x2 xx = {x, x};
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f16x8 a8, b8;
    for (int q = 0; q < 8; ++q) { a8[q] = (_Float16)(0.01f * q + (threadIdx.x & 7)); b8[q] = (_Float16)(0.02f * q); }
    asm volatile("" : "+v"(a8), "+v"(b8));
    f32x4* slot = lds + threadIdx.x;
    f32x4 v4 = {x, x, x, x};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long saved = __builtin_amdgcn_read_exec();
    __syncthreads();
    if (HALF) asm volatile("s_mov_b64 exec, 0xffffffff" ::: "exec");
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
        } else if (MODE == 2) {   // 4 independent accumulators: MFMA issue rate
            for (int q = 0; q < 4; ++q) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a8), "v"(b8));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a8), "v"(b8));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a8), "v"(b8));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a8), "v"(b8));
            }
        } else if (MODE == 3) {   // 16 ds_write_b128 (lane-private 16-byte slots, conflict-free)
#define S(j) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"((unsigned)(size_t)slot), "v"(v4), "n"(0) : "memory");
            REP16(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long t1 = clock64();
    if (HALF) asm volatile("s_mov_b64 exec, %0" :: "s"(saved) : "exec");
    float s = acc0[0] + acc1[5] + acc2[3] + acc3[7] + (*slot)[0];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += h[j] + p[j].x + p[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, bool HALF>
void run_half(const char* name, float* out, unsigned long long* cyc, double per_iter) {
    const int iters = 400;
    printf("%-44s %s", name, HALF ? "EXEC=low32 " : "EXEC=all   ");
    for (int wpb = 1; wpb <= 4; wpb *= 2) {
        const int blocks = 256 * wpb;
        std::vector<unsigned long long> h(blocks * 4);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((half_k<MODE, HALF>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        printf(" | w/SIMD=%d: %6.2f", wpb, s / h.size() / (iters * per_iter));
    }
    printf("\n");
    ck(name);
}

// ---------------------------------------------------------------- (b) wave-specialised pair
// a "VALU instruction" of the model streams: 8 independent v_fma chains, round robin (ILP like the step's)
// (round 6, second model) the real step is not issue-bound but a MIX: half of its instructions sit in dependent chains (8.4 ticks each for a
// lone wave), half have an independent neighbour (5.0): 6.7 ticks per instruction on average, as measured (3 820 cycles / 572).  V8 = four
// dependent + four independent v_fma; -DPAIR_ILP8 restores the first model (eight independent chains, 5.2 ticks per instruction).
#ifdef PAIR_ILP8
#define V8(h, x) asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n" \
                              "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" \
                              : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]) : "v"(x))
#else
#define V8(h, x) asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %0, %0, %8, %0\n" \
                              "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" \
                              : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]) : "v"(x))
#endif
#define PERM8(h) asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n" \
                              "v_permlane32_swap_b32 %0, %2\n v_permlane32_swap_b32 %1, %3\n v_permlane32_swap_b32 %4, %6\n v_permlane32_swap_b32 %5, %7" \
                              : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]))
template <int N8>
__device__ __forceinline__ void valu_block(float* h, float x) {
#pragma unroll
    for (int i = 0; i < N8; ++i) V8(h, x);
}
__device__ __forceinline__ void mlp_block(f32x16& hT0, f32x16& hM0, f32x16& hT1, f32x16& hM1, f16x8 a8, f16x8 b8) {
    const f32x16 z = {0};
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, x, y, z_) ({ f32x16 d_; asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=v"(d_) : "v"(A), "v"(B), "v"(C)); d_; })
    hT0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, z, 0, 0, 0); hM0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, z, 0, 0, 0);
    hT1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, z, 0, 0, 0); hM1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, z, 0, 0, 0);
    hT0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, hT0, 0, 0, 0); hM0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, hM0, 0, 0, 0);
    hT1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, hT1, 0, 0, 0); hM1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, hM1, 0, 0, 0);
    hM0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, hM0, 0, 0, 0); hM1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, hM1, 0, 0, 0);
#undef __builtin_amdgcn_mfma_f32_32x32x16_f16
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");   // inline-asm MFMA results: the compiler inserts no wait states before their readers
}
// layer-2-like: 32 x (v_pk_mul clamp + v_pk_fma) on the accumulators
__device__ __forceinline__ float layer2_block(const f32x16& a, const f32x16& b, const f32x16& c, const f32x16& d, float x) {
    f32x2 s0 = {0, 0}, s1 = {0, 0};
    const f32x2 w = {x, x}, dn = {0x1p-40f, 0x1p-40f};
#define L2(A, r) { f32x2 t = {A[r], A[r + 1]}; f32x2 u; asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(u) : "v"(t), "v"(dn)); \
                   if ((r) & 2) s1 = __builtin_elementwise_fma(w, u, s1); else s0 = __builtin_elementwise_fma(w, u, s0); }
#define L2A(A) L2(A, 0) L2(A, 2) L2(A, 4) L2(A, 6) L2(A, 8) L2(A, 10) L2(A, 12) L2(A, 14)
    L2A(a) L2A(b) L2A(c) L2A(c) L2A(c) L2A(d) L2A(d) L2A(d)
#undef L2A
#undef L2
    return (s0.x + s1.x) + (s0.y + s1.y);
}

// SOLO: 256 threads, one wave does everything.  kDep8 / kInd8 / kB8: VALU counts (in units of 8 instructions) of the part of A
// that depends on B's result, the independent part of A, and B's own VALU work around its 10 MFMA + 64 layer-2 instructions.
template <int kDep8, int kInd8, int kB8>
__global__ void __launch_bounds__(256) solo_k(float* out, unsigned long long* cyc, int iters) {
    float h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = threadIdx.x * 0.001f + j;
    float x = 1.0f + out[threadIdx.x] * 1e-9f;
    f16x8 a8, b8;
    for (int q = 0; q < 8; ++q) { a8[q] = (_Float16)(0.01f * q + (threadIdx.x & 7)); b8[q] = (_Float16)(0.02f * q); }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        valu_block<kB8 / 2>(h, x);                        // operand split
        PERM8(h);
        b8[0] = (_Float16)h[0];
        f32x16 hT0, hM0, hT1, hM1;
        mlp_block(hT0, hM0, hT1, hM1, a8, b8);
        const float r = layer2_block(hT0, hT1, hM0, hM1, x);
        h[0] += r;
        valu_block<kB8 / 2>(h, x);
        valu_block<kDep8>(h, x);
        valu_block<kInd8>(h, x);
    }
    unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += h[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// PAIR: 512 threads.  waves 0-3 = A (env waves), waves 4-7 = B (helper waves).  Mailboxes in LDS, two s_barrier per iteration.
template <int kDep8, int kInd8, int kB8, int mode>
__global__ void __launch_bounds__(512) pair_k(float* out, unsigned long long* cyc, unsigned* hwid, int iters) {
    __shared__ f32x4 xin[4][3][64];   // A -> B: 12 floats per env
    __shared__ f32x4 yout[4][64];     // B -> A: 4 floats per env
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pair = wave & 3;
    float h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = threadIdx.x * 0.001f + j;
    float x = 1.0f + out[threadIdx.x] * 1e-9f;
    if (lane == 0) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        hwid[blockIdx.x * 8 + wave] = hw;
    }
    yout[pair][lane] = f32x4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = clock64();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
            const f32x4 y = yout[pair][lane];             // B's result of the previous hand-over
            h[0] += y[0]; h[1] += y[1]; h[2] += y[2]; h[3] += y[3];
            if (mode != 3) valu_block<kDep8>(h, x);
            xin[pair][0][lane] = f32x4{h[0], h[1], h[2], h[3]};
            xin[pair][1][lane] = f32x4{h[4], h[5], h[6], h[7]};
            xin[pair][2][lane] = f32x4{h[0], h[2], h[4], h[6]};
            __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0)
            if (mode != 2 && mode != 5) __builtin_amdgcn_s_barrier();  // #1: inputs published
            __builtin_amdgcn_sched_barrier(0);
            if (mode != 3) valu_block<kInd8>(h, x);
            __builtin_amdgcn_sched_barrier(0);
            if (mode != 2 && mode != 5) __builtin_amdgcn_s_barrier();  // #2: results published
        }
    } else {
        f16x8 a8, b8;
        for (int q = 0; q < 8; ++q) { a8[q] = (_Float16)(0.01f * q + (threadIdx.x & 7)); b8[q] = (_Float16)(0.02f * q); }
        if (mode >= 4) __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it < iters; ++it) {
            if (mode != 2 && mode != 5) __builtin_amdgcn_s_barrier();  // #1
            const f32x4 i0 = xin[pair][0][lane], i1 = xin[pair][1][lane], i2 = xin[pair][2][lane];
            h[0] = i0[0]; h[1] = i0[1]; h[2] = i0[2]; h[3] = i0[3]; h[4] = i1[0]; h[5] = i1[1]; h[6] = i2[2]; h[7] = i2[3];
            if (mode != 1) {
                valu_block<kB8 / 2>(h, x);
                PERM8(h);
                b8[0] = (_Float16)h[0];
                f32x16 hT0, hM0, hT1, hM1;
                mlp_block(hT0, hM0, hT1, hM1, a8, b8);
                const float r = layer2_block(hT0, hT1, hM0, hM1, x);
                h[0] += r;
                valu_block<kB8 / 2>(h, x);
            }
            yout[pair][lane] = f32x4{h[0], h[1], h[2], h[3]};
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (mode != 2 && mode != 5) __builtin_amdgcn_s_barrier();  // #2
        }
    }
    unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += h[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int kDep8, int kInd8, int kB8>
void run_pair(float* out, unsigned long long* cyc, unsigned* hwid) {
    const int iters = 300;
    std::vector<unsigned long long> h(256 * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((solo_k<kDep8, kInd8, kB8>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
    hipMemcpy(h.data(), cyc, 256 * 4 * 8, hipMemcpyDeviceToHost);
    double solo = 0;
    for (int i = 0; i < 1024; ++i) solo += h[i];
    solo /= 1024.0 * iters;
    double pa = 0, pb = 0, res[6][2];
#define RUN_MODE(mode) { \
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((pair_k<kDep8, kInd8, kB8, mode>), dim3(256), dim3(512), 0, 0, out, cyc, hwid, iters); hipDeviceSynchronize(); } \
        hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost); \
        pa = pb = 0; \
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? pa : pb) += h[b * 8 + w]; \
        pa /= 1024.0 * iters; pb /= 1024.0 * iters; \
        res[mode][0] = pa; res[mode][1] = pb; }
    RUN_MODE(5) RUN_MODE(4) RUN_MODE(3) RUN_MODE(2) RUN_MODE(1) RUN_MODE(0)
#undef RUN_MODE
    const int na = 8 * (kDep8 + kInd8), nb = 8 * kB8 + 8 + 10 + 128 + 4;
    printf("A: %3d dependent + %3d independent VALU | B: %3d VALU + 8 swaps + 10 MFMA + 128 pk  (%d instr / step in all) | solo %6.0f cyc/step (%.2f/instr) | pair %6.0f cyc/step  = %.2fx | B idle: %6.0f | A idle: %6.0f | free-running: A %6.0f  B %6.0f | B at s_setprio 3: pair %6.0f, free-running A %6.0f  B %6.0f\n",
           8 * kDep8, 8 * kInd8, 8 * kB8, na + nb, solo, solo / (na + nb), pa, solo / pa, res[1][0], res[3][1], res[2][0], res[2][1], res[4][0], res[5][0], res[5][1]);
}

static void ck(const char* w) { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("ERROR after %s: %s\n", w, hipGetErrorString(e)); fflush(stdout); exit(1); } }
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; unsigned long long* cyc; unsigned* hwid;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 4096 * 8 * 8); hipMalloc(&hwid, 256 * 8 * 4);
    hipMemset(out, 0, 1024 * 512 * 4);
    printf("## (a) ticks per instruction per wave, full EXEC vs EXEC = low 32 lanes (x waves/SIMD = SIMD cycles per instruction)\n");
    run_half<0, false>("64 indep v_fma_f32", out, cyc, 64);           run_half<0, true>("64 indep v_fma_f32", out, cyc, 64);
    run_half<1, false>("64 indep v_pk_fma_f32", out, cyc, 64);        run_half<1, true>("64 indep v_pk_fma_f32", out, cyc, 64);
    run_half<2, false>("16 v_mfma_f32_32x32x16_f16, 4 accumulators", out, cyc, 16);  printf("(f16 MFMA under EXEC = low 32 lanes: the launch faults -- not a usable form)\n");
    run_half<3, false>("16 ds_write_b128 + wait", out, cyc, 16);
    printf("## (b) one wave per SIMD doing the whole step vs a wave-specialised pair per SIMD (two s_barrier + LDS hand-over per step)\n");
    run_pair<12, 50, 8>(out, cyc, hwid);     //  the fused E2E step: A = 96 dependent + 400 independent, B = the residual MLPs (64 + 8 + 10 + 128)
    run_pair<6, 56, 8>(out, cyc, hwid);      //  shorter dependent part
    run_pair<12, 42, 16>(out, cyc, hwid);    //  more of the work on B (e.g. the observation)
    run_pair<12, 34, 24>(out, cyc, hwid);
    run_pair<24, 38, 8>(out, cyc, hwid);     //  long dependent part
    printf("## (c) HW_ID of the eight waves of workgroups 0..3 of the pair kernel: wave -> (se, cu, simd)\n");
    std::vector<unsigned> hw(256 * 8);
    hipMemcpy(hw.data(), hwid, 256 * 8 * 4, hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) {
        const unsigned a = hw[b * 8 + w], c = hw[b * 8 + w + 4];
        same += (((a >> 4) & 3) == ((c >> 4) & 3)) && (((a >> 8) & 15) == ((c >> 8) & 15));
    }
    for (int b = 0; b < 4; ++b) {
        printf("wg %d:", b);
        for (int w = 0; w < 8; ++w) { const unsigned v = hw[b * 8 + w]; printf("  w%d=(se%u cu%u simd%u)", w, (v >> 13) & 7, (v >> 8) & 15, (v >> 4) & 3); }
        printf("\n");
    }
    printf("waves w and w+4 on the same SIMD of the same CU: %d of 1024 pairs\n", same);
    return 0;
}
