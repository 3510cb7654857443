// quad3d.hip -- the predecessor environments of the reference's "3D quad.ipynb" on gfx950 (include/quad3d.h).
//
//   Quadcopter3DVec (hover, float64)  Q3 cell 6      Quadcopter3DVecGates (float32)  Q3 cell 14      f_func  Q3 cell 2
//
// One lane = one env.  The reference's only state is the row-major `states[N][16]` array, which is also what step_wait
// returns, so HBM holds exactly that array (64 B / 128 B rows; the step kernel moves a block's rows as one contiguous
// slab of coalesced 16-byte accesses and transposes through LDS) plus step / target / episode counters.  Both envs are HBM-bound in
// the limit (about 230 B resp. 420 B per env-step against ~0.6 k resp. ~2 k flops); at 65 536 envs a launch is
// latency-bound like the race env's, which is what q3_step_many (state in registers across K steps) removes.
// Arithmetic follows the lambdified f_func term by term (-ffp-contract=off), in the element type of the reference's
// arrays; actions are float32 in both envs (SB3 hands float32), so constant*action products round in float32.
//
// Resets: Philox4x32-10 keyed (seed, global env id, episode, block) -- the reference draws from NumPy's global
// generator, the distributions are the reference's; Box-Muller with fixed-polynomial log / sin / cos so the CPU oracle
// used by the tests reproduces every draw bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

#include "../../include/quad3d.h"
#include "../../include/quadrace.h"
#include "quadrace_device.hpp"

namespace qr {
int set_last_error(int code, const std::string& msg);  // quadrace_abi.hip
}

namespace {

constexpr int kQ3Block = 256;
constexpr int kQ3MaxGates = 32;

struct Q3Params {
    int n, num_gates, max_steps, pad;
    double dt;
    double pos_thr, vel_thr, ang_thr, rat_thr;
    uint32_t seed_lo, seed_hi;
    uint32_t gid_lo, gid_hi;               // env_id_base
    float start[4];
    float gate[kQ3MaxGates][4];            // x, y, z, yaw (float32, like astype(np.float32))
    float gate_normal[kQ3MaxGates][2];     // cosf / sinf of the float32 yaw
};

template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = float4; static constexpr int kPerRow = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int kPerRow = 8; };

template <typename T>
__device__ __forceinline__ void load_row(const T* __restrict__ base, size_t i, T s[16]) {
    using V = typename Vec16<T>::type;
    const V* p = reinterpret_cast<const V*>(base + 16 * i);
    V v[Vec16<T>::kPerRow];
#pragma unroll
    for (int k = 0; k < Vec16<T>::kPerRow; ++k) v[k] = p[k];
    memcpy(s, v, sizeof(v));
}
template <typename T>
__device__ __forceinline__ void store_row(T* __restrict__ base, size_t i, const T s[16]) {
    using V = typename Vec16<T>::type;
    V* p = reinterpret_cast<V*>(base + 16 * i);
    V v[Vec16<T>::kPerRow];
    memcpy(v, s, sizeof(v));
#pragma unroll
    for (int k = 0; k < Vec16<T>::kPerRow; ++k) p[k] = v[k];
}

__device__ __forceinline__ void trig(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
__device__ __forceinline__ void trig(double x, double& s, double& c) { s = sin(x); c = cos(x); }
__device__ __forceinline__ float tangent(float x) { return tanf(x); }
__device__ __forceinline__ double tangent(double x) { return tan(x); }
// constant * float32 action, rounded in float32 (NumPy: python float times float32 array), then widened to T
template <typename T>
__device__ __forceinline__ T ua(double c, float u) { return (T)((float)c * u); }

// f_func of Q3 cell 2 (w_max = 12000: W = 4500 w + 7500; moments on the world velocities v_y, v_x)
template <typename T>
__device__ __forceinline__ void f_q3(const T* s, const float* u, T* ds) {
    const T vx = s[3], vy = s[4], vz = s[5], p = s[9], q = s[10], r = s[11];
    const T w1 = s[12], w2 = s[13], w3 = s[14], w4 = s[15];
    T sph, cph, sth, cth, sps, cps;
    trig(s[6], sph, cph);
    trig(s[7], sth, cth);
    trig(s[8], sps, cps);
    const T tth = tangent(s[7]);
    const T r01 = sph * sth * cps - sps * cph, r11 = sph * sps * sth + cph * cps;
    const T r02 = sph * sps + sth * cph * cps, r12 = -sph * cps + sps * sth * cph;
    const T S = (T)4500 * w1 + (T)4500 * w2 + (T)4500 * w3 + (T)4500 * w4 + (T)30000;
    const T W1 = (T)4500 * w1 + (T)7500, W2 = (T)4500 * w2 + (T)7500;
    const T W3 = (T)4500 * w3 + (T)7500, W4 = (T)4500 * w4 + (T)7500;
    const T W1s = W1 * W1, W2s = W2 * W2, W3s = W3 * W3, W4s = W4 * W4;
    const T kx = (T)1.07933887e-5, ky = (T)9.65250793e-6, kz = (T)2.7862899e-5;
    const T kw = (T)4.36301076e-8, kh = (T)0.0625501332;
    const T vby = vx * r01 + vy * r11 + vz * sph * cth;
    const T vbx = vx * cps * cth + vy * sps * cth - vz * sth;
    const T Tt = -kw * W1s - kw * W2s - kw * W3s - kw * W4s - (kz * vx * r02 + kz * vy * r12 + kz * vz * cph * cth) * S -
                 kh * (vby * vby) - kh * (vbx * vbx);
    const T Fy = -ky * vx * r01 - ky * vy * r11 - ky * vz * sph * cth;  // times S below, in source order
    const T Fx = -kx * vx * cps * cth - kx * vy * sps * cth + kx * vz * sth;
    ds[0] = vx;
    ds[1] = vy;
    ds[2] = vz;
    ds[3] = r02 * Tt + r01 * Fy * S + Fx * S * cps * cth;
    ds[4] = r12 * Tt + r11 * Fy * S + Fx * S * sps * cth;
    ds[5] = Fy * S * sph * cth - Fx * S * sth + Tt * cph * cth + (T)9.81;
    ds[6] = p + q * sph * tth + r * cph * tth;
    ds[7] = q * cph - r * sph;
    ds[8] = q * sph / cth + r * cph / cth;
    ds[9] = (T)-0.896247240618101 * q * r - (T)8.79803364238411 * vy + (T)1.55842505518764e-6 * W1s -
            (T)1.55842505518764e-6 * W2s - (T)1.55842505518764e-6 * W3s + (T)1.55842505518764e-6 * W4s;
    ds[10] = (T)0.924315619967794 * p * r + (T)10.4077084541063 * vx + (T)9.79081191626409e-7 * W1s +
             (T)9.79081191626409e-7 * W2s - (T)9.79081191626409e-7 * W3s - (T)9.79081191626409e-7 * W4s;
    ds[11] = (T)-0.163583252190847 * p * q - (T)0.395780237098345 * r - ua<T>(15.0045045277507, u[0]) +
             ua<T>(15.0045045277507, u[1]) - ua<T>(15.0045045277507, u[2]) + ua<T>(15.0045045277507, u[3]) +
             (T)9.37324867332035 * w1 - (T)9.37324867332035 * w2 + (T)9.37324867332035 * w3 - (T)9.37324867332035 * w4;
    ds[12] = ua<T>(16.6666666666667, u[0]) - (T)16.6666666666667 * w1;
    ds[13] = ua<T>(16.6666666666667, u[1]) - (T)16.6666666666667 * w2;
    ds[14] = ua<T>(16.6666666666667, u[2]) - (T)16.6666666666667 * w3;
    ds[15] = ua<T>(16.6666666666667, u[3]) - (T)16.6666666666667 * w4;
}

// ---- reset RNG (this build's specification; the tests restate it on the CPU) -------------------------------------------------
__device__ __forceinline__ void q3_block(const Q3Params& P, int i, uint32_t episode, int block, uint32_t o[4]) {
    const uint32_t lo = P.gid_lo + (uint32_t)i;
    const uint32_t hi = P.gid_hi + (lo < P.gid_lo ? 1u : 0u);
    qr::philox4x32_10(lo, hi, episode, (uint32_t)block, P.seed_lo, P.seed_hi, o);
}

__device__ __forceinline__ float q3_log(float x) {  // x in (0, 1]; Cephes logf scheme, float32 operations only
    uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m + -1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m + -1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m + -1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m + -2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    const float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}

__device__ __forceinline__ void q3_normal_pair(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-8f;
    const float t = (float)(b >> 8) * 2.384185791015625e-7f;
    const int quad = (int)t;
    const float x = (t - (float)quad - 0.5f) * 1.5707963267948966f;
    const float z = x * x;
    float sp = -1.9515295891e-4f;
    sp = sp * z + 8.3321608736e-3f;
    sp = sp * z + -1.6666654611e-1f;
    const float sn = x + x * z * sp;
    float cp = 2.443315711809948e-5f;
    cp = cp * z + -1.388731625493765e-3f;
    cp = cp * z + 4.166664568298827e-2f;
    const float cs = 1.0f - 0.5f * z + z * z * cp;
    const float rad = sqrtf(-2.0f * q3_log(u1));
    const float c = quad == 0 ? cs : (quad == 1 ? -sn : (quad == 2 ? -cs : sn));
    const float s = quad == 0 ? sn : (quad == 1 ? cs : (quad == 2 ? -sn : -cs));
    z0 = rad * c;
    z1 = rad * s;
}

template <typename T>
struct Q3Env {
    T s[16];
    int target, steps;
    uint32_t episode;
};

// reset_ of Q3 cell 6: x,y,z U(-5,5); v, phi, theta U(-1,1); psi U(-pi,pi); rates, w U(-1,1)
__device__ __forceinline__ void q3_reset_env(const Q3Params& P, int i, Q3Env<double>& e) {
    const uint32_t ep = e.episode;
    e.episode = ep + 1u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        uint32_t o[4];
        q3_block(P, i, ep, b, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = 4 * b + k;
            const double u = (double)o[k] * 2.3283064365386963e-10;
            const double hi = j < 3 ? 5.0 : (j == 8 ? 3.141592653589793 : 1.0);
            e.s[j] = -hi + (hi - -hi) * u;
        }
    }
    e.steps = 0;
}

// reset_ of Q3 cell 14: random segment midpoint + 0.1 N(0,1); v 0.1 N; angles N; rates 0.1 N; w U(-1,1); target = segment
__device__ __forceinline__ void q3_reset_env(const Q3Params& P, int i, Q3Env<float>& e) {
    const uint32_t ep = e.episode;
    e.episode = ep + 1u;
    float nrm[12];
    uint32_t o[4];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        q3_block(P, i, ep, b, o);
        q3_normal_pair(o[0], o[1], nrm[4 * b + 0], nrm[4 * b + 1]);
        q3_normal_pair(o[2], o[3], nrm[4 * b + 2], nrm[4 * b + 3]);
    }
    q3_block(P, i, ep, 3, o);
#pragma unroll
    for (int k = 0; k < 4; ++k) e.s[12 + k] = -1.0f + 2.0f * ((float)(o[k] >> 8) * 5.9604644775390625e-8f);
    q3_block(P, i, ep, 4, o);
    const int seg = (int)__umulhi(o[0], (uint32_t)P.num_gates);
    const float* p0 = seg == 0 ? P.start : P.gate[seg - 1];
    const float* p1 = P.gate[seg];
#pragma unroll
    for (int k = 0; k < 3; ++k) e.s[k] = 0.1f * nrm[k] + (p0[k] + p1[k]) / 2.0f;
#pragma unroll
    for (int k = 3; k < 6; ++k) e.s[k] = 0.1f * nrm[k];
#pragma unroll
    for (int k = 6; k < 9; ++k) e.s[k] = nrm[k];
#pragma unroll
    for (int k = 9; k < 12; ++k) e.s[k] = 0.1f * nrm[k];
    e.steps = 0;
    e.target = seg;
}

// Quadcopter3DVec.step_wait (Q3 cell 6)
__device__ __forceinline__ double q3_step_env(const Q3Params& P, int i, Q3Env<double>& e, const float u[4], bool& done,
                                              bool& trunc) {
    double ds[16];
    e.steps += 1;
    f_q3<double>(e.s, u, ds);
#pragma unroll
    for (int k = 0; k < 16; ++k) e.s[k] = e.s[k] + P.dt * ds[k];
    const double* s = e.s;
    const double npos = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    const double nvel = sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
    const double nang = sqrt(s[6] * s[6] + s[7] * s[7] + s[8] * s[8]);
    const double nrat = sqrt(s[9] * s[9] + s[10] * s[10] + s[11] * s[11]);
    double reward = -0.002 * npos + -0.002 * nvel + -0.0001 * nang + -0.0001 * nrat;
    const bool ang_ok = fabs(s[6]) < P.ang_thr && fabs(s[7]) < P.ang_thr && fabs(s[8]) < P.ang_thr;
    const bool rat_ok = fabs(s[9]) < P.rat_thr && fabs(s[10]) < P.rat_thr && fabs(s[11]) < P.rat_thr;
    const bool goal = npos < P.pos_thr && nvel < P.vel_thr && ang_ok && rat_ok;
    if (goal) reward = 100.0;
    const bool oob = fabs(s[0]) > 10.0 || fabs(s[1]) > 10.0 || fabs(s[2]) > 10.0 || fabs(s[6]) > 3.141592653589793 ||
                     fabs(s[7]) > 3.141592653589793;
    if (oob) reward = -1.0;
    const bool max_steps = e.steps >= P.max_steps;
    done = goal || oob || max_steps;
    trunc = max_steps || oob;
    if (done) q3_reset_env(P, i, e);
    return reward;
}

// Quadcopter3DVecGates.step_wait (Q3 cell 14)
__device__ __forceinline__ float q3_step_env(const Q3Params& P, int i, Q3Env<float>& e, const float u[4], bool& done,
                                             bool& trunc) {
    float ds[16], ns[16];
    const float dt = (float)P.dt;
    const float* s = e.s;
    e.steps += 1;
    f_q3<float>(s, u, ds);
#pragma unroll
    for (int k = 0; k < 16; ++k) ns[k] = s[k] + dt * ds[k];
    const int g = min(max(e.target, 0), P.num_gates - 1);  // (an out-of-range target_gates raises IndexError upstream)
    const float gx = P.gate[g][0], gy = P.gate[g][1], gz = P.gate[g][2];
    const float ox = s[0] - gx, oy = s[1] - gy, oz = s[2] - gz;
    const float nx = ns[0] - gx, ny = ns[1] - gy, nz = ns[2] - gz;
    const float d2g_old = sqrtf(ox * ox + oy * oy + oz * oz);
    const float d2g_new = sqrtf(nx * nx + ny * ny + nz * nz);
    const float rat_penalty = 0.0001f * sqrtf(ns[9] * ns[9] + ns[10] * ns[10] + ns[11] * ns[11]);
    float reward = d2g_old - d2g_new - rat_penalty;
    const float n0 = P.gate_normal[g][0], n1 = P.gate_normal[g][1];
    const float proj_old = ox * n0 + oy * n1, proj_new = nx * n0 + ny * n1;
    const bool crossed = proj_old < 0.0f && proj_new > 0.0f;
    const bool inside = fabsf(nx) < 0.5f && fabsf(ny) < 0.5f && fabsf(nz) < 0.5f;
    const bool outside = fabsf(nx) > 0.5f || fabsf(ny) > 0.5f || fabsf(nz) > 0.5f;
    const bool gate_passed = crossed && inside, gate_collision = crossed && outside;
    if (gate_collision) reward = -10.0f;
    const bool ground = s[2] > 0.0f;  // the reference tests the PRE-step state here ...
    if (ground) reward = -10.0f;
    const bool oob = fabsf(s[0]) > 10.0f || fabsf(s[1]) > 10.0f || fabsf(s[9]) > 1000.0f || fabsf(s[10]) > 1000.0f ||
                     fabsf(s[11]) > 1000.0f;  // ... and here (no reward override)
    const bool max_steps = e.steps >= P.max_steps;
    if (gate_passed) e.target += 1;
    const bool final_passed = e.target >= P.num_gates;
    if (final_passed) reward = 10.0f;
    done = max_steps || gate_collision || ground || final_passed || oob;
    trunc = max_steps;
#pragma unroll
    for (int k = 0; k < 16; ++k) e.s[k] = ns[k];
    if (done) q3_reset_env(P, i, e);
    return reward;
}

template <typename T>
struct Q3Buffers {
    T* states;          // [N][16]
    int32_t* target;    // [N]
    int32_t* steps;     // [N]
    uint32_t* episode;  // [N]
};

template <typename T>
__device__ __forceinline__ void q3_load(const Q3Buffers<T>& B, int i, Q3Env<T>& e) {
    load_row<T>(B.states, i, e.s);
    e.target = B.target[i];
    e.steps = B.steps[i];
    e.episode = B.episode[i];
}
template <typename T>
__device__ __forceinline__ void q3_store(const Q3Buffers<T>& B, int i, const Q3Env<T>& e) {
    store_row<T>(B.states, i, e.s);
    B.target[i] = e.target;
    B.steps[i] = e.steps;
    B.episode[i] = e.episode;
}

// Block-cooperative row I/O for the per-step kernel: the block's 256 rows are one contiguous slab (16 / 32 KB); it is moved
// with fully coalesced 16-byte accesses (consecutive lanes -> consecutive addresses) and transposed to lane = row through
// LDS (row stride padded by 16 bytes against bank conflicts).  Measured at 1 Mi envs against per-lane row accesses: see
// DESIGN.md section 10.
template <typename T>
struct RowTile {
    using V = typename Vec16<T>::type;
    static constexpr int kCh = Vec16<T>::kPerRow;      // 16-byte chunks per row
    static constexpr int kStride = kCh + 1;            // LDS row stride in chunks (padded)
    static constexpr int kLdsChunks = kQ3Block * kStride;
};

template <typename T>
__device__ __forceinline__ void tile_load_rows(const T* __restrict__ base, int row0, int rows, typename RowTile<T>::V* lds, T s[16]) {
    using R = RowTile<T>;
    using V = typename R::V;
    const V* src = reinterpret_cast<const V*>(base + (size_t)16 * row0);
#pragma unroll
    for (int q = 0; q < R::kCh; ++q) {
        const int cidx = q * kQ3Block + threadIdx.x;   // chunk index within the slab
        if (cidx < rows * R::kCh) lds[(cidx / R::kCh) * R::kStride + (cidx % R::kCh)] = src[cidx];
    }
    __syncthreads();
    V v[R::kCh];
#pragma unroll
    for (int k = 0; k < R::kCh; ++k) v[k] = lds[threadIdx.x * R::kStride + k];
    memcpy(s, v, sizeof(v));
}

// writes the block's rows to up to two destinations (the state array and the caller's states_out)
template <typename T>
__device__ __forceinline__ void tile_store_rows(T* __restrict__ dst0, T* __restrict__ dst1, int row0, int rows,
                                                typename RowTile<T>::V* lds, const T s[16]) {
    using R = RowTile<T>;
    using V = typename R::V;
    V v[R::kCh];
    memcpy(v, s, sizeof(v));
#pragma unroll
    for (int k = 0; k < R::kCh; ++k) lds[threadIdx.x * R::kStride + k] = v[k];
    __syncthreads();
    V* d0 = reinterpret_cast<V*>(dst0 + (size_t)16 * row0);
    V* d1 = dst1 ? reinterpret_cast<V*>(dst1 + (size_t)16 * row0) : nullptr;
#pragma unroll
    for (int q = 0; q < R::kCh; ++q) {
        const int cidx = q * kQ3Block + threadIdx.x;
        if (cidx < rows * R::kCh) {
            const V x = lds[(cidx / R::kCh) * R::kStride + (cidx % R::kCh)];
            d0[cidx] = x;
            if (d1) d1[cidx] = x;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kQ3Block) void q3_step_kernel(Q3Params P, Q3Buffers<T> B, const float4* __restrict__ actions,
                                                           T* __restrict__ states_out, T* __restrict__ rew_out,
                                                           uint8_t* __restrict__ done_out,
                                                           uint8_t* __restrict__ trunc_out) {
    __shared__ typename RowTile<T>::V lds[RowTile<T>::kLdsChunks];
    const int row0 = blockIdx.x * kQ3Block;
    const int rows = min(kQ3Block, P.n - row0);
    const int i = row0 + threadIdx.x;
    const bool active = i < P.n;
    const int ii = active ? i : row0;  // ragged-tail lanes shadow the block's first env (they take part in the barriers)
    Q3Env<T> e;
    tile_load_rows<T>(B.states, row0, rows, lds, e.s);
    if (!active) load_row<T>(B.states, ii, e.s);
    e.target = B.target[ii];
    e.steps = B.steps[ii];
    e.episode = B.episode[ii];
    const float4 a = actions[ii];
    const float u[4] = {a.x, a.y, a.z, a.w};
    bool done, trunc;
    const T reward = q3_step_env(P, ii, e, u, done, trunc);
    tile_store_rows<T>(B.states, states_out, row0, rows, lds, e.s);
    if (!active) return;
    B.target[i] = e.target;
    B.steps[i] = e.steps;
    B.episode[i] = e.episode;
    if (rew_out) rew_out[i] = reward;
    if (done_out) done_out[i] = done ? 1 : 0;
    if (trunc_out) trunc_out[i] = trunc ? 1 : 0;
}

// a wave's 64 rows = one contiguous [64][16] block of a row-major array: transposed through a wave-private LDS tile and written with
// 16-byte-per-lane stores that cover whole cache lines (per-lane row stores would scatter 16-byte pieces over 64 lines per instruction)
template <typename T>
__device__ __forceinline__ void wave_store_rows(T* __restrict__ dst, size_t wave_row0, int rows_in_wave, typename RowTile<T>::V* tile,
                                                int lane, const T s[16]) {
    using R = RowTile<T>;
    using V = typename R::V;
    V v[R::kCh];
    memcpy(v, s, sizeof(v));
#pragma unroll
    for (int k = 0; k < R::kCh; ++k) tile[lane * R::kStride + k] = v[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    V* d = reinterpret_cast<V*>(dst + (size_t)16 * wave_row0);
#pragma unroll
    for (int q = 0; q < R::kCh; ++q) {
        const int cidx = q * 64 + lane;
        if (cidx < rows_in_wave * R::kCh) {   // streaming store: nothing in this launch reads the rows again
            typedef float f32x4n __attribute__((ext_vector_type(4)));
            static_assert(sizeof(V) == sizeof(f32x4n), "16-byte chunks");
            const V x = tile[(cidx / R::kCh) * R::kStride + (cidx % R::kCh)];
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4n, x), reinterpret_cast<f32x4n*>(d + cidx));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the tile may be rewritten
}

// kRows: also write env.states after EVERY step to states_steps [K][N][16] and the truncation flags to trunc_out [K][N] -- the rollout
// a trainer consumes (q3_rollout); without them (q3_step_many) a step moves its action in and reward / done out only.
template <typename T, bool kRows>
__global__ __launch_bounds__(kQ3Block) void q3_rollout_kernel(Q3Params P, Q3Buffers<T> B, const float4* __restrict__ actions,
                                                              int K, T* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                                                              uint8_t* __restrict__ trunc_out, T* __restrict__ states_steps,
                                                              T* __restrict__ states_out) {
    __shared__ typename RowTile<T>::V lds[kRows ? RowTile<T>::kLdsChunks : 1];
    const int i = blockIdx.x * kQ3Block + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave_row0 = i - lane;
    if (wave_row0 >= P.n) return;               // whole wave past the end
    const bool active = i < P.n;
    if (!kRows && !active) return;
    const int ii = active ? i : wave_row0;      // kRows: ragged-tail lanes shadow the wave's first env (they move rows of the tile)
    const int rows_in_wave = min(64, P.n - wave_row0);
    typename RowTile<T>::V* tile = lds + (kRows ? (threadIdx.x >> 6) * 64 * RowTile<T>::kStride : 0);
    Q3Env<T> e;
    q3_load(B, ii, e);
    float4 a = actions[ii];
    for (int k = 0; k < K; ++k) {
        const float u[4] = {a.x, a.y, a.z, a.w};
        if (k + 1 < K) a = actions[(size_t)(k + 1) * P.n + ii];  // next step's action is in flight during this step
        bool done, trunc;
        const T reward = q3_step_env(P, ii, e, u, done, trunc);
        if (active) {
            if (rew_out) rew_out[(size_t)k * P.n + i] = reward;
            if (done_out) done_out[(size_t)k * P.n + i] = done ? 1 : 0;
            if (kRows && trunc_out) trunc_out[(size_t)k * P.n + i] = trunc ? 1 : 0;
        }
        if constexpr (kRows) wave_store_rows<T>(states_steps + (size_t)k * P.n * 16, (size_t)wave_row0, rows_in_wave, tile, lane, e.s);
    }
    if (!active) return;
    q3_store(B, i, e);
    if (states_out) store_row<T>(states_out, i, e.s);
}

template <typename T>
__global__ __launch_bounds__(kQ3Block) void q3_reset_kernel(Q3Params P, Q3Buffers<T> B, const uint8_t* __restrict__ mask,
                                                            T* __restrict__ states_out) {
    const int i = blockIdx.x * kQ3Block + threadIdx.x;
    if (i >= P.n) return;
    Q3Env<T> e;
    q3_load(B, i, e);
    if (!mask || mask[i]) {
        q3_reset_env(P, i, e);
        q3_store(B, i, e);
    }
    if (states_out) store_row<T>(states_out, i, e.s);
}

template <typename T>
__global__ __launch_bounds__(kQ3Block) void q3_copy_state_kernel(int n, Q3Buffers<T> B, T* states_out, int32_t* target_out,
                                                                 int32_t* steps_out, const T* states_in,
                                                                 const int32_t* target_in, const int32_t* steps_in) {
    const int i = blockIdx.x * kQ3Block + threadIdx.x;
    if (i >= n) return;
    T s[16];
    if (states_in) {
        load_row<T>(states_in, i, s);
        store_row<T>(B.states, i, s);
    }
    if (target_in) B.target[i] = target_in[i];
    if (steps_in) B.steps[i] = steps_in[i];
    if (states_out) {
        load_row<T>(B.states, i, s);
        store_row<T>(states_out, i, s);
    }
    if (target_out) target_out[i] = B.target[i];
    if (steps_out) steps_out[i] = B.steps[i];
}

int q3fail(int code, const std::string& m) { return qr::set_last_error(code, m); }

#define Q3_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return q3fail(QR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

}  // namespace

struct q3_env {
    int kind = 0, device = 0;
    bool has_track = false;
    Q3Params P{};
    void* d_states = nullptr;
    int32_t* d_target = nullptr;
    int32_t* d_steps = nullptr;
    uint32_t* d_episode = nullptr;
    template <typename T>
    Q3Buffers<T> buffers() const { return Q3Buffers<T>{static_cast<T*>(d_states), d_target, d_steps, d_episode}; }
    int grid() const { return (P.n + kQ3Block - 1) / kQ3Block; }
};

namespace {
int q3_ready(const q3_env* e) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    if (e->kind == Q3_KIND_GATES && !e->has_track) return q3fail(QR_E_STATE, "q3_set_track has not been called");
    return QR_OK;
}
}  // namespace

extern "C" {

int q3_create(int kind, int num_envs, int device, uint64_t env_id_base, q3_env** out) {
    if (!out) return q3fail(QR_E_INVALID, "out is null");
    *out = nullptr;
    if (kind != Q3_KIND_HOVER && kind != Q3_KIND_GATES) return q3fail(QR_E_INVALID, "kind must be Q3_KIND_HOVER or Q3_KIND_GATES");
    if (num_envs < 1) return q3fail(QR_E_INVALID, "num_envs must be >= 1");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return q3fail(QR_E_NO_DEVICE, "no HIP device visible: libquadrace has no CPU fallback");
    if (device < 0 || device >= count) return q3fail(QR_E_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    Q3_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return q3fail(QR_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    Q3_HIP(hipSetDevice(device));
    q3_env* e = new q3_env();
    e->kind = kind;
    e->device = device;
    e->P.n = num_envs;
    e->P.max_steps = 1000;  // Q3 cell 6 / 14 __init__
    e->P.dt = 0.01;
    e->P.pos_thr = 0.3;
    e->P.vel_thr = 0.3;
    e->P.ang_thr = 10 * 3.141592653589793 / 180;
    e->P.rat_thr = 10 * 3.141592653589793 / 180;
    e->P.gid_lo = (uint32_t)env_id_base;
    e->P.gid_hi = (uint32_t)(env_id_base >> 32);
    const size_t n = (size_t)num_envs, esz = kind == Q3_KIND_HOVER ? 8 : 4;
    hipError_t err = hipMalloc(&e->d_states, n * 16 * esz);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_target), n * 4);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_steps), n * 4);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->d_episode), n * 4);
    if (err == hipSuccess) err = hipMemset(e->d_states, 0, n * 16 * esz);  // np.zeros
    if (err == hipSuccess) err = hipMemset(e->d_target, 0, n * 4);
    if (err == hipSuccess) err = hipMemset(e->d_steps, 0, n * 4);
    if (err == hipSuccess) err = hipMemset(e->d_episode, 0, n * 4);
    if (err != hipSuccess) {
        q3_destroy(e);
        return q3fail(QR_E_HIP, std::string("allocating env state: ") + hipGetErrorString(err));
    }
    *out = e;
    return QR_OK;
}

int q3_destroy(q3_env* e) {
    if (!e) return QR_OK;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(e->d_states);
    (void)hipFree(e->d_target);
    (void)hipFree(e->d_steps);
    (void)hipFree(e->d_episode);
    delete e;
    return QR_OK;
}

int q3_num_envs(const q3_env* e) { return e ? e->P.n : q3fail(QR_E_INVALID, "null q3_env handle"); }
int q3_elem_size(const q3_env* e) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    return e->kind == Q3_KIND_HOVER ? 8 : 4;
}

int q3_set_track(q3_env* e, const float* gate_pos, const float* gate_yaw, int G, const float start_pos[3]) {
    if (!e || !gate_pos || !gate_yaw || !start_pos) return q3fail(QR_E_INVALID, "null argument");
    if (e->kind != Q3_KIND_GATES) return q3fail(QR_E_STATE, "q3_set_track applies to Q3_KIND_GATES only");
    if (G < 1 || G > kQ3MaxGates) return q3fail(QR_E_INVALID, "num_gates must be in 1..32");
    e->P.num_gates = G;
    for (int g = 0; g < G; ++g) {
        for (int k = 0; k < 3; ++k) e->P.gate[g][k] = gate_pos[3 * g + k];
        e->P.gate[g][3] = gate_yaw[g];
        e->P.gate_normal[g][0] = cosf(gate_yaw[g]);  // np.cos / np.sin of the float32 yaw (Q3 cell 14 step_wait)
        e->P.gate_normal[g][1] = sinf(gate_yaw[g]);
    }
    for (int k = 0; k < 3; ++k) e->P.start[k] = start_pos[k];
    e->has_track = true;
    return QR_OK;
}

int q3_set_limits(q3_env* e, int max_steps, double dt) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    if (max_steps < 1 || !(dt > 0.0)) return q3fail(QR_E_INVALID, "max_steps must be >= 1 and dt > 0");
    e->P.max_steps = max_steps;
    e->P.dt = dt;
    return QR_OK;
}

int q3_set_thresholds(q3_env* e, double pos, double vel, double ang, double rat) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    if (e->kind != Q3_KIND_HOVER) return q3fail(QR_E_STATE, "q3_set_thresholds applies to Q3_KIND_HOVER only");
    e->P.pos_thr = pos;
    e->P.vel_thr = vel;
    e->P.ang_thr = ang;
    e->P.rat_thr = rat;
    return QR_OK;
}

int q3_seed(q3_env* e, uint64_t seed) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    Q3_HIP(hipSetDevice(e->device));
    e->P.seed_lo = (uint32_t)seed;
    e->P.seed_hi = (uint32_t)(seed >> 32);
    Q3_HIP(hipDeviceSynchronize());
    Q3_HIP(hipMemset(e->d_episode, 0, (size_t)e->P.n * 4));
    return QR_OK;
}

int q3_reset(q3_env* e, const uint8_t* mask, void* states_out, void* stream) {
    if (int rc = q3_ready(e)) return rc;
    Q3_HIP(hipSetDevice(e->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (e->kind == Q3_KIND_HOVER)
        q3_reset_kernel<double><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<double>(), mask, static_cast<double*>(states_out));
    else
        q3_reset_kernel<float><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<float>(), mask, static_cast<float*>(states_out));
    Q3_HIP(hipGetLastError());
    return QR_OK;
}

int q3_step(q3_env* e, const float* actions, void* states_out, void* rew_out, uint8_t* done_out, uint8_t* trunc_out,
            void* stream) {
    if (int rc = q3_ready(e)) return rc;
    if (!actions) return q3fail(QR_E_INVALID, "actions_dev is null");
    Q3_HIP(hipSetDevice(e->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float4* a = reinterpret_cast<const float4*>(actions);
    if (e->kind == Q3_KIND_HOVER)
        q3_step_kernel<double><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<double>(), a, static_cast<double*>(states_out),
                                                              static_cast<double*>(rew_out), done_out, trunc_out);
    else
        q3_step_kernel<float><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<float>(), a, static_cast<float*>(states_out),
                                                             static_cast<float*>(rew_out), done_out, trunc_out);
    Q3_HIP(hipGetLastError());
    return QR_OK;
}

static int q3_launch_rollout(q3_env* e, const float* actions, int K, void* states_steps, void* rew_out, uint8_t* done_out,
                             uint8_t* trunc_out, void* states_out, void* stream) {
    if (int rc = q3_ready(e)) return rc;
    if (!actions) return q3fail(QR_E_INVALID, "actions_dev is null");
    if (K < 1) return q3fail(QR_E_INVALID, "num_steps must be >= 1");
    Q3_HIP(hipSetDevice(e->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float4* a = reinterpret_cast<const float4*>(actions);
    if (e->kind == Q3_KIND_HOVER) {
        if (states_steps)
            q3_rollout_kernel<double, true><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<double>(), a, K, static_cast<double*>(rew_out), done_out,
                                                                           trunc_out, static_cast<double*>(states_steps), static_cast<double*>(states_out));
        else
            q3_rollout_kernel<double, false><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<double>(), a, K, static_cast<double*>(rew_out), done_out,
                                                                            nullptr, nullptr, static_cast<double*>(states_out));
    } else {
        if (states_steps)
            q3_rollout_kernel<float, true><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<float>(), a, K, static_cast<float*>(rew_out), done_out,
                                                                          trunc_out, static_cast<float*>(states_steps), static_cast<float*>(states_out));
        else
            q3_rollout_kernel<float, false><<<e->grid(), kQ3Block, 0, st>>>(e->P, e->buffers<float>(), a, K, static_cast<float*>(rew_out), done_out,
                                                                           nullptr, nullptr, static_cast<float*>(states_out));
    }
    Q3_HIP(hipGetLastError());
    return QR_OK;
}

int q3_step_many(q3_env* e, const float* actions, int K, void* rew_out, uint8_t* done_out, void* states_out, void* stream) {
    return q3_launch_rollout(e, actions, K, nullptr, rew_out, done_out, nullptr, states_out, stream);
}

int q3_rollout(q3_env* e, const float* actions, int K, void* states_steps_out, void* rew_out, uint8_t* done_out, uint8_t* trunc_out,
               void* stream) {
    if (!states_steps_out) return q3fail(QR_E_INVALID, "states_steps_out_dev is null (q3_step_many is the form without per-step rows)");
    return q3_launch_rollout(e, actions, K, states_steps_out, rew_out, done_out, trunc_out, nullptr, stream);
}

int q3_get_state(q3_env* e, void* states, int32_t* target, int32_t* steps, void* stream) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    Q3_HIP(hipSetDevice(e->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (e->kind == Q3_KIND_HOVER)
        q3_copy_state_kernel<double><<<e->grid(), kQ3Block, 0, st>>>(e->P.n, e->buffers<double>(), static_cast<double*>(states),
                                                                    target, steps, nullptr, nullptr, nullptr);
    else
        q3_copy_state_kernel<float><<<e->grid(), kQ3Block, 0, st>>>(e->P.n, e->buffers<float>(), static_cast<float*>(states),
                                                                   target, steps, nullptr, nullptr, nullptr);
    Q3_HIP(hipGetLastError());
    return QR_OK;
}

int q3_set_state(q3_env* e, const void* states, const int32_t* target, const int32_t* steps, void* stream) {
    if (!e) return q3fail(QR_E_INVALID, "null q3_env handle");
    Q3_HIP(hipSetDevice(e->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (e->kind == Q3_KIND_HOVER)
        q3_copy_state_kernel<double><<<e->grid(), kQ3Block, 0, st>>>(e->P.n, e->buffers<double>(), nullptr, nullptr, nullptr,
                                                                    static_cast<const double*>(states), target, steps);
    else
        q3_copy_state_kernel<float><<<e->grid(), kQ3Block, 0, st>>>(e->P.n, e->buffers<float>(), nullptr, nullptr, nullptr,
                                                                   static_cast<const float*>(states), target, steps);
    Q3_HIP(hipGetLastError());
    return QR_OK;
}

}  // extern "C"
