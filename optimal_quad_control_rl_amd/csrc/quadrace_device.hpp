// quadrace_device.hpp -- gfx950 device code of the vectorised quadrotor race environment.
//
// One lane simulates one environment.  Everything here is float32 elementwise ODE work (no MFMA):
// the cost is HBM traffic for the state/obs plus ~2 k VALU instructions per env-step, so the layout
// rules that matter are coalesced 16-byte-per-lane accesses and LDS-resident constant tables.
//
// Behavioural contract = the reference's Quadcopter3DGates (R: "3D quad race.ipynb",
// I: "3D quad race INDI inner loop.ipynb"; raw .ipynb line numbers):
//   equations of motion     R:57-152 / I:43-110
//   residual MLPs           R:227-262 (weights NNDroneModel/*.pt)
//   step / reward / dones   R:498-595 / I:301-385
//   reset distributions     R:452-493 / I:267-299   (stream: this build's Philox4x32-10 spec)
//   gate-frame observation  R:365-450 / I:218-265
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qr {

constexpr int kE2E = 0;
constexpr int kINDI = 1;
constexpr int kMaxGates = 32;
constexpr int kMaxGatesAhead = 4;
constexpr int kGateStride = 12;   // floats per gate row in the LDS table (3 x float4)
constexpr int kMlpFloats = 740;  // weights + biases of the two residual networks
constexpr int kBlock = 256;       // 4 wave64 per workgroup

// flags
constexpr int kFlagResidual = 1;
constexpr int kFlagPause = 2;
constexpr int kFlagPauseIfCollision = 4;

// Gate table row (LDS, 48 B): [x y z yaw | cos(yaw) sin(yaw) 0 0 | rel_x rel_y rel_z rel_yaw]
//
// MLP table (LDS, read with wave-uniform addresses = hardware broadcast).  Both hidden layers are fused into one
// 64-wide layer (units 0..31 thrust net, 32..63 moment net) and the table is stored in CONSUMPTION order as
// twelve 64-float chunks, so the kernel can stream it through a double-buffered register window:
//   chunk 0      b1[64]
//   chunk 1..7   W1t[i][64]   i = 0..6: inputs (w1..w4, vbx, vby, vbz) feed both nets
//   chunk 8      W1m[7][32] | W1m[8][32]      inputs (p, q) feed only the moment net (units 32..63)
//   chunk 9      W1m[9][32] | W2[0][32]       input r ; thrust output row (units 0..31)
//   chunk 10     W2[1][32]  | W2[2][32]       moment output rows (units 32..63)
//   chunk 11     W2[3][32]  | b2[4] | pad
// Why LDS and a software pipeline (measured on MI355X, N = 65 536 = one wave per SIMD): a wave that waits for each
// ds_read (or s_load -- the scalar path was tried: 24 exposed scalar-cache round trips) spends 8.3-8.5 k of its
// 18 k cycles in the MLP; streaming the next chunk while the current one feeds 64 FMAs hides that latency.
constexpr int kMlpChunks = 12;
constexpr int kMlpTableFloats = kMlpChunks * 64;  // 768 (740 used)

struct Params {
    // planar state in HBM (structure of float4 arrays, plane stride = n_stride elements)
    float4* ws;         // E2E: 4 planes (x y z vx | vy vz phi theta | psi p q r | w1 w2 w3 w4); INDI: 3 planes
    float* tn;          // INDI: T_norm
    float4* dA;         // E2E: (M_ext_x, M_ext_y, M_ext_z, F_ext_z)
    float2* dB;         // E2E: (F_ext_x, F_ext_y)
    int2* ts;           // (target_gate, step_count)
    uint32_t* episode;  // per-env reset counter = RNG stream position
    const float* tables;  // device copy of [gate table | mlp table]
    int n, n_stride, num_gates, gates_ahead, max_steps, flags;
    float dt;
    uint32_t seed_lo, seed_hi, gid_lo, gid_hi;  // Philox key, global id of env 0
    float start[3];
    float dist_lo[6], dist_hi[6];
    float dist_scale;
    float obs_lo[4], obs_inv[4];  // observation scaling of (Mx,My,Mz,Fz): lo and 1/(hi-lo) after the R:419-441 fix-up
#ifdef QR_PHASE_TIMING
    unsigned long long* ticks;  // [n_waves][16] shader-clock stamps (profiling build only, tools/phase_timing.py)
#endif
};

#ifdef QR_PHASE_TIMING
#define QR_TICK(P, slot)                                                                      \
    do {                                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                           \
        if ((P).ticks && (threadIdx.x & 63) == 0)                                             \
            (P).ticks[((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 16 + (slot)] = clock64(); \
    } while (0)
#else
#define QR_TICK(P, slot) do { } while (0)
#endif

// -------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11).  counter = (global env id lo, hi, episode, block), key = seed.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const uint32_t lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }
// Separately rounded operations (bit-identical to the CPU oracle).  NB: HIP's __fadd_rn/__fmul_rn are plain
// operators that hipcc may still contract into an FMA, so contraction is switched off with the pragma.
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
// lo + (hi-lo)*u
__device__ __forceinline__ float uni(float lo, float hi, float u) {
#pragma clang fp contract(off)
    const float span = hi - lo;
    const float prod = span * u;
    return lo + prod;
}

template <int V>
struct Env {
    static constexpr int S = (V == kE2E) ? 16 : 13;
    float s[S];   // world state
    float d[6];   // constant external disturbances (E2E only)
    int target, steps;
};

// reset_ for one env: distributions of R:455-489 / I:270-296
template <int V>
__device__ __forceinline__ void reset_env(const Params& P, Env<V>& e, uint32_t gid_lo, uint32_t gid_hi,
                                          uint32_t episode) {
    constexpr int NB = (V == kE2E) ? 6 : 4;
    float u[4 * NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint32_t o[4];
        philox4x32_10(gid_lo, gid_hi, episode, (uint32_t)b, P.seed_lo, P.seed_hi, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) u[4 * b + k] = u01(o[k]);
    }
    const float pi9 = 0.3490658503988659f, pi = 3.141592653589793f;
    e.s[0] = add_rn(uni(-0.5f, 0.5f, u[0]), P.start[0]);
    e.s[1] = add_rn(uni(-0.5f, 0.5f, u[1]), P.start[1]);
    e.s[2] = add_rn(uni(-0.5f, 0.5f, u[2]), P.start[2]);
    e.s[3] = uni(-0.5f, 0.5f, u[3]);
    e.s[4] = uni(-0.5f, 0.5f, u[4]);
    e.s[5] = uni(-0.5f, 0.5f, u[5]);
    e.s[6] = uni(-pi9, pi9, u[6]);
    e.s[7] = uni(-pi9, pi9, u[7]);
    e.s[8] = uni(-pi, pi, u[8]);
    e.s[9] = uni(-0.1f, 0.1f, u[9]);
    e.s[10] = uni(-0.1f, 0.1f, u[10]);
    e.s[11] = uni(-0.1f, 0.1f, u[11]);
    if constexpr (V == kE2E) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e.s[12 + k] = uni(-1.0f, 1.0f, u[12 + k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) e.d[k] = mul_rn(P.dist_scale, uni(P.dist_lo[k], P.dist_hi[k], u[16 + k]));
    } else {
        e.s[12] = uni(-0.1f, 0.1f, u[12]);
    }
    e.steps = 0;
    e.target = 0;
}

// -------------------------------------------------------------------------------------------------
// Rotation (ZYX Euler, R:97-100) -- shared by the residual-MLP input and the equations of motion
// -------------------------------------------------------------------------------------------------
struct Rot {
    float sph, cph, sth, cth, sps, cps;
    float r00, r10, r20, r01, r11, r21, r02, r12, r22;
};

__device__ __forceinline__ Rot make_rot(float phi, float theta, float psi) {
    Rot R;
    sincosf(phi, &R.sph, &R.cph);
    sincosf(theta, &R.sth, &R.cth);
    sincosf(psi, &R.sps, &R.cps);
    // The library is compiled with -ffp-contract=off: every FMA below is explicit, so the arithmetic is fixed by
    // this source (step kernel and fused rollout kernel are bit-identical by construction).
    const float ss = R.sph * R.sth, cs = R.cph * R.sth;
    R.r00 = R.cps * R.cth;
    R.r10 = R.sps * R.cth;
    R.r20 = -R.sth;
    R.r01 = fmaf(ss, R.cps, -(R.sps * R.cph));
    R.r11 = fmaf(ss, R.sps, R.cph * R.cps);
    R.r21 = R.sph * R.cth;
    R.r02 = fmaf(cs, R.cps, R.sph * R.sps);
    R.r12 = fmaf(cs, R.sps, -(R.sph * R.cps));
    R.r22 = R.cph * R.cth;
    return R;
}

// -------------------------------------------------------------------------------------------------
// Residual thrust / moment MLPs: 7->32->1 and 10->32->3, ReLU (R:227-262).  64 independent accumulator
// chains per lane; weights stream from LDS (broadcast ds_read_b128) one 64-float chunk ahead of their use.
// -------------------------------------------------------------------------------------------------
struct Chunk {
    float4 v[16];
};

__device__ __forceinline__ void mlp_fetch(const float* __restrict__ W, int c, Chunk& dst) {
    const float4* w4 = reinterpret_cast<const float4*>(W) + 16 * c;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst.v[j] = w4[j];
}

__device__ __forceinline__ void fma4(const float4 w, float x, float* h) {
    h[0] = fmaf(w.x, x, h[0]);
    h[1] = fmaf(w.y, x, h[1]);
    h[2] = fmaf(w.z, x, h[2]);
    h[3] = fmaf(w.w, x, h[3]);
}

// One pipeline stage: consume chunk `cur` (64 multiply-accumulates per lane) while fetching chunk `c_next` into
// `nxt`, interleaved one ds_read_b128 per four FMAs so the LDS pipe and the VALU stay busy together (all four
// waves of a workgroup run this in lock-step, so a burst of reads followed by a burst of FMAs would idle one
// unit while the other works).  kind: 0 = 64 units * x0;  1 = units 32..63 * (x0 for the first half, x1 second).
template <int KIND>
__device__ __forceinline__ void mlp_stage(const float* __restrict__ W, int c_next, const Chunk& cur, Chunk& nxt,
                                          float x0, float x1, float* h) {
    const float4* w4 = reinterpret_cast<const float4*>(W) + 16 * c_next;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        nxt.v[j] = w4[j];
        if (KIND == 0) fma4(cur.v[j], x0, h + 4 * j);
        else fma4(cur.v[j], j < 8 ? x0 : x1, h + 32 + 4 * (j & 7));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// dot(W2 row, relu(h)) over 32 hidden units, 4 independent chains
__device__ __forceinline__ float mlp_dot32(const Chunk& w, int half, const float* h) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 ww = w.v[8 * half + j];
        a0 = fmaf(ww.x, h[4 * j + 0], a0);
        a1 = fmaf(ww.y, h[4 * j + 1], a1);
        a2 = fmaf(ww.z, h[4 * j + 2], a2);
        a3 = fmaf(ww.w, h[4 * j + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ void residual_mlp(const float* __restrict__ W, const float x[10], float& thrust,
                                             float moment[3]) {
    Chunk A, B;
    float h[64];
    mlp_fetch(W, 0, A);  // biases
    mlp_fetch(W, 1, B);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        h[4 * j + 0] = A.v[j].x; h[4 * j + 1] = A.v[j].y; h[4 * j + 2] = A.v[j].z; h[4 * j + 3] = A.v[j].w;
    }
    mlp_stage<0>(W, 2, B, A, x[0], 0.0f, h);
    mlp_stage<0>(W, 3, A, B, x[1], 0.0f, h);
    mlp_stage<0>(W, 4, B, A, x[2], 0.0f, h);
    mlp_stage<0>(W, 5, A, B, x[3], 0.0f, h);
    mlp_stage<0>(W, 6, B, A, x[4], 0.0f, h);
    mlp_stage<0>(W, 7, A, B, x[5], 0.0f, h);
    mlp_stage<0>(W, 8, B, A, x[6], 0.0f, h);      // consumes chunk 7 (input vbz), fetches chunk 8
    mlp_stage<1>(W, 9, A, B, x[7], x[8], h);      // chunk 8: p | q  -> units 32..63
    // chunk 9 (in B): first half = input r, second half = thrust output row
    mlp_fetch(W, 10, A);
#pragma unroll
    for (int j = 0; j < 8; ++j) fma4(B.v[j], x[9], h + 32 + 4 * j);
#pragma unroll
    for (int j = 0; j < 64; ++j) h[j] = fmaxf(h[j], 0.0f);
    const float out0 = mlp_dot32(B, 1, h);
    mlp_fetch(W, 11, B);
    const float out1 = mlp_dot32(A, 0, h + 32);
    const float out2 = mlp_dot32(A, 1, h + 32);
    const float out3 = mlp_dot32(B, 0, h + 32);
    const float4 b2 = B.v[8];
    thrust = out0 + b2.x;
    moment[0] = out1 + b2.y;
    moment[1] = out2 + b2.z;
    moment[2] = out3 + b2.w;
}

// -------------------------------------------------------------------------------------------------
// Equations of motion: ds = f(s, u, d)
// -------------------------------------------------------------------------------------------------
// E2E Bebop model (R:57-152; constants pre-folded as in SURVEY Appendix A)
__device__ __forceinline__ void eom_e2e(const float* s, const Rot& R, const float vb[3], const float u[4],
                                        const float M[3], const float F[3], float* ds) {
    const float p = s[9], q = s[10], r = s[11];
    const float w1 = s[12], w2 = s[13], w3 = s[14], w4 = s[15];
    const float W1 = fmaf(4000.0f, w1, 7000.0f), W2 = fmaf(4000.0f, w2, 7000.0f);   // R:106-109
    const float W3 = fmaf(4000.0f, w3, 7000.0f), W4 = fmaf(4000.0f, w4, 7000.0f);
    const float S = (W1 + W2) + (W3 + W4);
    const float W1s = W1 * W1, W2s = W2 * W2, W3s = W3 * W3, W4s = W4 * W4;
    const float Fx = fmaf(-1.07933887e-5f * vb[0], S, F[0]);                          // R:125
    const float Fy = fmaf(-9.65250793e-6f * vb[1], S, F[1]);                          // R:126
    float T = fmaf(-4.36301076e-8f, (W1s + W2s) + (W3s + W4s), F[2]);                 // R:124
    T = fmaf(-2.7862899e-5f * vb[2], S, T);
    T = fmaf(-0.0625501332f, fmaf(vb[0], vb[0], vb[1] * vb[1]), T);
    ds[0] = s[3];
    ds[1] = s[4];
    ds[2] = s[5];
    ds[3] = fmaf(R.r00, Fx, fmaf(R.r01, Fy, R.r02 * T));                              // R:138
    ds[4] = fmaf(R.r10, Fx, fmaf(R.r11, Fy, R.r12 * T));
    ds[5] = fmaf(R.r20, Fx, fmaf(R.r21, Fy, fmaf(R.r22, T, 9.81f)));
    const float inv_cth = 1.0f / R.cth;
    const float tth = R.sth * inv_cth;
    const float qr_mix = fmaf(q, R.sph, r * R.cph);
    ds[6] = fmaf(qr_mix, tth, p);                                                     // R:140
    ds[7] = fmaf(q, R.cph, -(r * R.sph));                                             // R:141
    ds[8] = qr_mix * inv_cth;                                                         // R:142
    ds[9] = fmaf(1103.7527593819f, M[0],
                 fmaf(-0.896247240618101f * q, r,
                      fmaf(-8.79803364238411f, vb[1], 1.55842505518764e-6f * ((W1s - W2s) - (W3s - W4s)))));  // R:144
    ds[10] = fmaf(805.152979066023f, M[1],
                  fmaf(0.924315619967794f * p, r,
                       fmaf(10.4077084541063f, vb[0], 9.79081191626409e-7f * ((W1s + W2s) - (W3s + W4s)))));  // R:145
    ds[11] = fmaf(486.854917234664f, M[2],
                  fmaf(-0.163583252190847f * p, q,
                       fmaf(-0.395780237098345f, r,
                            fmaf(-13.3373373580007f, (u[0] - u[1]) + (u[2] - u[3]),
                                 8.33177659850698f * ((w1 - w2) + (w3 - w4))))));                            // R:146,131
    ds[12] = 16.6666666666667f * (u[0] - w1);                                         // R:112-115
    ds[13] = 16.6666666666667f * (u[1] - w2);
    ds[14] = 16.6666666666667f * (u[2] - w3);
    ds[15] = 16.6666666666667f * (u[3] - w4);
}

// INDI inner-loop model (I:43-110)
__device__ __forceinline__ void eom_indi(const float* s, const Rot& R, const float vb[3], const float u[4],
                                         float* ds) {
    const float p = s[9], q = s[10], r = s[11], Tn = s[12];
    const float Dx = -0.33915248f * vb[0], Dy = -0.4314916f * vb[1];                      // I:73-74,86-87
    const float mT = fmaf(-8.0f, Tn, -8.0f);                                              // I:94, T_max = 16
    ds[0] = s[3];
    ds[1] = s[4];
    ds[2] = s[5];
    ds[3] = fmaf(R.r00, Dx, fmaf(R.r01, Dy, R.r02 * mT));                                 // I:95
    ds[4] = fmaf(R.r10, Dx, fmaf(R.r11, Dy, R.r12 * mT));
    ds[5] = fmaf(R.r20, Dx, fmaf(R.r21, Dy, fmaf(R.r22, mT, 9.81f)));
    const float inv_cth = 1.0f / R.cth;
    const float tth = R.sth * inv_cth;
    const float qr_mix = fmaf(q, R.sph, r * R.cph);
    ds[6] = fmaf(qr_mix, tth, p);                                                         // I:97-99
    ds[7] = fmaf(q, R.cph, -(r * R.sph));
    ds[8] = qr_mix * inv_cth;
    ds[9] = fmaf(-33.3333333333333f, p, 100.0f * u[0]);                                   // I:101-104
    ds[10] = fmaf(-33.3333333333333f, q, 100.0f * u[1]);
    ds[11] = fmaf(-33.3333333333333f, r, 66.6666666666667f * u[2]);
    ds[12] = 33.3333333333333f * (u[3] - Tn);
}

// -------------------------------------------------------------------------------------------------
// Gate-frame observation of one env (update_states_gate, R:365-450 / I:218-265) into o[obs_len]
// -------------------------------------------------------------------------------------------------
template <int V, int GA>
__device__ __forceinline__ void observe(const Params& P, const float* __restrict__ gates, const Env<V>& e,
                                        float* o) {
    constexpr int S = Env<V>::S;
    const float4 g0 = *reinterpret_cast<const float4*>(gates + kGateStride * e.target);      // x y z yaw
    const float2 cs = *reinterpret_cast<const float2*>(gates + kGateStride * e.target + 4);  // cos sin
    const float dx = e.s[0] - g0.x, dy = e.s[1] - g0.y;
    o[0] = fmaf(dx, cs.x, dy * cs.y);            // R:380-382
    o[1] = fmaf(dy, cs.x, -(dx * cs.y));
    o[2] = e.s[2] - g0.z;                        // R:383
    o[3] = fmaf(e.s[3], cs.x, e.s[4] * cs.y);    // R:386-389
    o[4] = fmaf(e.s[4], cs.x, -(e.s[3] * cs.y));
    o[5] = e.s[5];
    o[6] = e.s[6];
    o[7] = e.s[7];
    // yaw relative to the gate, wrapped like NumPy's `%= 2*pi` followed by the > pi fold (R:392-397).
    // fmod is exact, and so is fma(-k, 2pi, x) for the correct integer k.
    const float twopi = 6.283185307179586f, pi = 3.141592653589793f;
    const float x = e.s[8] - g0.w;
    float k = floorf(x * 0.15915494309189535f);
    float yaw = fmaf(-k, twopi, x);
    if (yaw < 0.0f) yaw += twopi;
    if (yaw >= twopi) yaw -= twopi;
    if (yaw > pi) yaw -= twopi;
    o[8] = yaw;
#pragma unroll
    for (int i = 9; i < S; ++i) o[i] = e.s[i];
#pragma unroll
    for (int a = 0; a < GA; ++a) {  // R:406-412
        int idx = e.target + a + 1;
        while (idx >= P.num_gates) idx -= P.num_gates;
        const float4 rel = *reinterpret_cast<const float4*>(gates + kGateStride * idx + 8);
        o[S + 4 * a + 0] = rel.x;
        o[S + 4 * a + 1] = rel.y;
        o[S + 4 * a + 2] = rel.z;
        o[S + 4 * a + 3] = rel.w;
    }
    if constexpr (V == kE2E) {  // R:414-448: (Mx, My, Mz, Fz) mapped to [-1,1] by their ranges
        constexpr int base = S + 4 * GA;
        const int col[4] = {0, 1, 2, 5};
#pragma unroll
        for (int c = 0; c < 4; ++c)  // 2*(d - lo)/(hi - lo) - 1 with the host-precomputed 1/(hi - lo)
            o[base + c] = fmaf(2.0f * (e.d[col[c]] - P.obs_lo[c]), P.obs_inv[c], -1.0f);
    }
}

// -------------------------------------------------------------------------------------------------
// One env step (step_wait, R:501-595 / I:303-385).  Returns reward; sets done / trunc flags.
// On auto-reset `episode_next` is bumped and `did_reset` set so the caller persists the new disturbances.
// -------------------------------------------------------------------------------------------------
template <int V>
__device__ __forceinline__ float step_env(const Params& P, const float* __restrict__ gates,
                                          const float* __restrict__ mlp, Env<V>& e, const float u[4],
                                          uint32_t gid_lo, uint32_t gid_hi, uint32_t* episode_ptr, bool& done,
                                          bool& trunc, bool& did_reset) {
    constexpr int S = Env<V>::S;
    const Rot R = make_rot(e.s[6], e.s[7], e.s[8]);
    QR_TICK(P, 3);
    float vb[3];  // R:103 body velocity = R^T v
    vb[0] = fmaf(e.s[3], R.r00, fmaf(e.s[4], R.r10, e.s[5] * R.r20));
    vb[1] = fmaf(e.s[3], R.r01, fmaf(e.s[4], R.r11, e.s[5] * R.r21));
    vb[2] = fmaf(e.s[3], R.r02, fmaf(e.s[4], R.r12, e.s[5] * R.r22));
    float ds[S];
    if constexpr (V == kE2E) {
        float M[3] = {e.d[0], e.d[1], e.d[2]};
        float F[3] = {e.d[3], e.d[4], e.d[5]};
        if (P.flags & kFlagResidual) {  // R:502-509: residual evaluated on the PRE-step state
            const float x[10] = {e.s[12], e.s[13], e.s[14], e.s[15], vb[0], vb[1], vb[2], e.s[9], e.s[10], e.s[11]};
            float thrust, moment[3];
            residual_mlp(mlp, x, thrust, moment);
            M[0] += moment[0]; M[1] += moment[1]; M[2] += moment[2];
            F[2] += thrust;
        }
        QR_TICK(P, 4);
        eom_e2e(e.s, R, vb, u, M, F, ds);
    } else {
        eom_indi(e.s, R, vb, u, ds);
    }
    float nw[S];
#pragma unroll
    for (int k = 0; k < S; ++k) nw[k] = fmaf(P.dt, ds[k], e.s[k]);  // forward Euler, R:512
    const int steps = e.steps + 1;                                   // R:514

    const float4 g0 = *reinterpret_cast<const float4*>(gates + kGateStride * e.target);      // R:518-519
    const float2 cs = *reinterpret_cast<const float2*>(gates + kGateStride * e.target + 4);
    const float ox = e.s[0] - g0.x, oy = e.s[1] - g0.y, oz = e.s[2] - g0.z;
    const float nx = nw[0] - g0.x, ny = nw[1] - g0.y, nz = nw[2] - g0.z;
    const float d2g_old = sqrtf(fmaf(ox, ox, fmaf(oy, oy, oz * oz)));  // R:522-525
    const float d2g_new = sqrtf(fmaf(nx, nx, fmaf(ny, ny, nz * nz)));
    float reward = d2g_old - d2g_new;
    const float proj_old = fmaf(ox, cs.x, oy * cs.y);               // R:528-532
    const float proj_new = fmaf(nx, cs.x, ny * cs.y);
    const bool crossed = (proj_old < 0.0f) && (proj_new > 0.0f);
    const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
    const bool gate_passed = crossed && (ax < 0.5f) && (ay < 0.5f) && (az < 0.5f);     // R:533
    const bool gate_collision = crossed && ((ax > 0.5f) || (ay > 0.5f) || (az > 0.5f)); // R:534
    if (gate_passed) reward = fmaf(-10.0f, d2g_new, 10.0f);         // R:537
    if (gate_collision) reward = -10.0f;                            // R:540
    const bool ground = nw[2] > 0.0f;                               // R:543-544
    if (ground) reward = -10.0f;
    const bool oob = (fabsf(nw[0]) > 10.0f) || (fabsf(nw[1]) > 10.0f) || (fabsf(nw[9]) > 1000.0f) ||
                     (fabsf(nw[10]) > 1000.0f) || (fabsf(nw[11]) > 1000.0f);           // R:549-550
    if (oob) reward = -10.0f;
    trunc = steps >= P.max_steps;                                   // R:553
    if (gate_passed) e.target = (e.target + 1 == P.num_gates) ? 0 : e.target + 1;      // R:556-557
    done = trunc || ground || gate_collision || oob;                // R:566
    e.steps = steps;
    did_reset = false;
    QR_TICK(P, 5);
    if (P.flags & kFlagPause) {                                     // R:570-572: state not advanced
        done = false;
    } else if (P.flags & kFlagPauseIfCollision) {                   // R:573-578: freeze done envs, no reset
        if (!done) {
#pragma unroll
            for (int k = 0; k < S; ++k) e.s[k] = nw[k];
        }
    } else {                                                        // R:581-585
#pragma unroll
        for (int k = 0; k < S; ++k) e.s[k] = nw[k];
        if (done) {
            const uint32_t ep = *episode_ptr;
            *episode_ptr = ep + 1u;
            reset_env<V>(P, e, gid_lo, gid_hi, ep);
            did_reset = true;
        }
    }
    return reward;
}

}  // namespace qr
