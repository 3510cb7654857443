// quadrace_device.hpp -- gfx950 device code of the vectorised quadrotor race environment.
//
// One lane simulates one environment.  The step is float32 elementwise ODE work (~550 VALU instructions per env-step)
// plus ONE small per-wave GEMM: the first layer of the residual thrust / moment MLPs runs on the matrix core -- since round 4
// as 10 x v_mfma_f32_32x32x16_f16 per wave-step with both operands split into two f16 pieces (exact products, f32
// accumulation: as accurate as the float32 fmaf chain / f32 matrix instruction of rounds 1-3, at a fifth of its cycles; see
// residual_mlp() for the measurements that led there).  The layout rules that matter are coalesced 16-byte-per-lane accesses,
// LDS-resident constant tables and MLP weights held once per wave in registers.
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
// Residual-MLP table (device image; residual_mlp() below says why layer 1 runs on the f16 matrix core with split operands):
//   tabA [5][64][4 dwords]  layer-1 A operands of v_mfma_f32_32x32x16_f16: lane l holds 8 f16 = hidden row l&31, k-slots 8(l>>5)..+7
//                  of MFMA q.  W = W0 + W1 (f16 pieces, round to nearest even; b likewise), inputs 0..6 = (w1..w4, vbx, vby, vbz),
//                  7..9 = (p, q, r):
//                  q = 0 thrust A : lo [W0t[row][0..6], b0t[row]]                  hi [same with W0t -> the X1 slots: W0t[row][0..6], 0]
//                  q = 1 thrust B : lo [W1t[row][0..6], b1t[row]]                  hi zeros
//                  q = 2 moment A : lo [W0m[row][0..6], b0m[row]]                  hi [W0m[row][0..6], 0]
//                  q = 3 moment B : lo [W1m[row][0..6], b1m[row]]                  hi zeros
//                  q = 4 moment C : lo [W0m[row][7..9], 0, W0m[row][7..9], 0]      hi [W1m[row][7..9], 0, 0, 0, 0, 0]
//   tabW2 [2][64]  layer-2 weights x 2^40 (see relu2_scaled) seen by the lanes of wave half h = lane>>5; accumulator register
//                  r holds hidden row(r,h) = (r&3) + 8*(r>>2) + 4h:   [r] = W2t[0][row]   [16 + 16m + r] = W2m[m][row], m = 0..2
//   b2 [4]         output biases (thrust, moment x/y/z)
constexpr int kMlpQuads = 5;
constexpr int kOffTabA = 0, kOffTabW2 = kMlpQuads * 64 * 4, kOffB2 = kOffTabW2 + 128;
constexpr int kMlpTableFloats = 1424;  // 1412 used, padded to a multiple of 16
static_assert(kOffB2 + 4 <= kMlpTableFloats && kMlpTableFloats % 16 == 0, "MLP table layout");
//
// Reset table [24][4] (host-built, staged to LDS with the gate rows): row t = (lo, hi - lo, add, mul) of the t-th
// reset draw, value = ((lo + (hi - lo) * u) + add) * mul with separately rounded operations:
//   t = 0..2   position: U(-0.5, 0.5) + start_pos[t]        t = 3..5 velocity U(-0.5, 0.5)     (R:455-461)
//   t = 6,7    phi, theta: U(-pi/9, pi/9)   t = 8 psi: U(-pi, pi)   t = 9..11 rates U(-0.1, 0.1) (R:463-469)
//   E2E: t = 12..15 motor speeds U(-1, 1); t = 16..21 disturbances scale * U(range)   (R:471-489)
//   INDI: t = 12 T_norm U(-0.1, 0.1)                                                   (I:285)
// Device table image: [MLP table (784) | reset table (96) | gate rows (G * 12)]
constexpr int kResetTableFloats = 96;
constexpr int kOffResetImage = kMlpTableFloats;
constexpr int kOffGatesImage = kMlpTableFloats + kResetTableFloats;

struct Params {
    // planar state in HBM (structure of float4 arrays, plane stride = n_stride elements)
    float4* ws;         // E2E: 4 planes (x y z vx | vy vz phi theta | psi p q r | w1 w2 w3 w4); INDI: 3 planes
    float* tn;          // INDI: T_norm
    float4* dA;         // E2E: (M_ext_x, M_ext_y, M_ext_z, F_ext_z)
    float2* dB;         // E2E: (F_ext_x, F_ext_y)
    int2* ts;           // (target_gate | episode << 8, step_count); episode = per-env reset counter (RNG position)
    const float* tables;  // device copy of [gate table | mlp table]
    int n, n_stride, num_gates, gates_ahead, max_steps, flags;
    float dt;
    uint32_t seed_lo, seed_hi, gid_lo, gid_hi;  // Philox key, global id of env 0
    float obs_lo[4], obs_inv[4];  // observation scaling of (Mx,My,Mz,Fz): lo and 1/(hi-lo) after the R:419-441 fix-up
    float* term_obs;    // optional (qr_set_terminal_obs): rows [N][obs_len] (per-step kernel) / [K][N][obs_len] (K-step kernels)
                        // receiving the PRE-reset observation of every env that finished at that step
#if defined(QR_PHASE_TIMING) || defined(QR_CLOCK_PROBE)
    unsigned long long* ticks;  // [n_waves][16] shader-clock stamps (profiling builds only, tools/phase_timing.py / clock_probe.py)
    int tick_on;
#endif
};

#ifdef QR_CLOCK_PROBE
// tools/clock_probe.py: un-drained stamps around a kernel's step loop -- shader cycles (s_memtime) AND the constant 100 MHz
// counter (s_memrealtime), so that cycles per step and the EFFECTIVE shader clock under that load both come out; slot 6 = HW_ID
#define QR_CLOCK_STAMP(P, slot)                                                                                   \
    do {                                                                                                          \
        if ((P).ticks && (threadIdx.x & 63) == 0) {                                                               \
            unsigned long long* t_ = (P).ticks + ((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 16; \
            t_[2 * (slot)] = clock64();                                                                           \
            t_[2 * (slot) + 1] = wall_clock64();                                                                  \
        }                                                                                                         \
    } while (0)
#define QR_CLOCK_HWID(P)                                                                                          \
    do {                                                                                                          \
        if ((P).ticks && (threadIdx.x & 63) == 0) {                                                               \
            unsigned hw_, xcc_;                                                                                   \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                     \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                   \
            (P).ticks[((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 16 + 8] =                   \
                (unsigned long long)hw_ | ((unsigned long long)xcc_ << 32);                                       \
        }                                                                                                         \
    } while (0)
#else
#define QR_CLOCK_STAMP(P, slot) do { } while (0)
#define QR_CLOCK_HWID(P) do { } while (0)
#endif

#ifdef QR_PHASE_TIMING
#ifdef QR_PHASE_TIMING_NODRAIN   /* where the wave IS at each stamp (only the LDS / scalar queue drains: s_memtime returns through it) */
#define QR_TICK_DRAIN() do { } while (0)
#else                            /* what each phase costs in isolation: memory queues drained at every stamp */
#define QR_TICK_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#define QR_TICK(P, slot)                                                                      \
    do {                                                                                      \
        if (!(P).tick_on) break;                                                              \
        QR_TICK_DRAIN();                                                                      \
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

// one round of the same generator (for callers that spread the ten rounds over idle issue slots); round r = 0..9 uses the
// key bumped r times, so ten calls with r = 0..9 equal philox4x32_10()
__device__ __forceinline__ void philox4x32_round(uint32_t c[4], uint32_t k0, uint32_t k1, int r) {
    k0 += 0x9E3779B9u * (uint32_t)r;
    k1 += 0xBB67AE85u * (uint32_t)r;
    const uint32_t lo0 = 0xD2511F53u * c[0], hi0 = __umulhi(0xD2511F53u, c[0]);
    const uint32_t lo1 = 0xCD9E8D57u * c[2], hi1 = __umulhi(0xCD9E8D57u, c[2]);
    c[0] = hi1 ^ c[1] ^ k0;
    c[1] = lo1;
    c[2] = hi0 ^ c[3] ^ k1;
    c[3] = lo0;
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
// Streaming ("nt") stores for everything a step writes.  Nothing a step kernel writes is read again by the same
// launch, and per-XCD L2s are written back / invalidated at every kernel boundary anyway (the next step's state
// reads come from the fabric: PMC FETCH_SIZE = the algorithmic read bytes), so keeping these lines as dirty L2
// residents only defers their write-back to the end-of-kernel drain.  Measured with tools/ubench/launch_floor.hip
// (the step kernel's 285 B/env traffic, no arithmetic, 65 536 envs, back-to-back launches): 3.82 us plain
// stores, 3.09-3.11 us nt / write-through stores; an empty kernel of the same shape costs 2.75 us.
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef int i32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store(float4* p, const float4 v) {
    const f32x4s x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4s*>(p));
}
__device__ __forceinline__ void stream_store(float2* p, const float2 v) {
    const f32x2s x = {v.x, v.y};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x2s*>(p));
}
__device__ __forceinline__ void stream_store(int2* p, const int2 v) {
    const i32x2s x = {v.x, v.y};
    __builtin_nontemporal_store(x, reinterpret_cast<i32x2s*>(p));
}
__device__ __forceinline__ void stream_store(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stream_store(uint8_t* p, uint8_t v) { __builtin_nontemporal_store(v, p); }

template <int V>
struct Env {
    static constexpr int S = (V == kE2E) ? 16 : 13;
    float s[S];   // world state
    float d[6];   // constant external disturbances (E2E only)
    float od[4];  // E2E, kernels that keep the env in registers over many steps: the observation columns of (Mx, My, Mz, Fz) -- constant
                  // within an episode, so computed at load / reset instead of in every step (disturbance_obs_values(); observe_with<.., true>)
    int target, steps;
    uint32_t episode;  // resets so far (24 bits are persisted next to the target gate)
};

// R:414-448: (Mx, My, Mz, Fz) mapped to [-1,1] by their ranges: 2*(d - lo)/(hi - lo) - 1 with the host-precomputed 1/(hi - lo)
__device__ __forceinline__ void disturbance_obs_values(const Params& P, const float* d, float* od) {
    const int col[4] = {0, 1, 2, 5};
#pragma unroll
    for (int c = 0; c < 4; ++c) od[c] = fmaf(2.0f * (d[col[c]] - P.obs_lo[c]), P.obs_inv[c], -1.0f);
}

// value of one reset draw from its table row (lo, span, add, mul): ((lo + span*u) + add) * mul, no FMA contraction
__device__ __forceinline__ float reset_value(const float4 row, uint32_t bits) {
    return mul_rn(add_rn(add_rn(row.x, mul_rn(row.y, u01(bits))), row.z), row.w);
}

template <int V>
__device__ __forceinline__ void assign_reset(Env<V>& e, const float* v) {
#pragma unroll
    for (int k = 0; k < Env<V>::S; ++k) e.s[k] = v[k];
    if constexpr (V == kE2E) {
#pragma unroll
        for (int k = 0; k < 6; ++k) e.d[k] = v[16 + k];
    }
    e.steps = 0;
    e.target = 0;
    e.episode = (e.episode + 1u) & 0xFFFFFFu;
}

// reset_ for one env, every lane for itself (reset kernel: whole batches reset at once).
// Distributions of R:455-489 / I:270-296; `rtab` = reset table in LDS.
template <int V>
constexpr int reset_value_count() { return 4 * ((V == kE2E) ? 6 : 4); }
// the reset draws of env (gid) for episode `episode`: 4 values per Philox block, each with its own table row
template <int V>
__device__ __forceinline__ void reset_values(const Params& P, const float* __restrict__ rtab, uint32_t episode, uint32_t gid_lo,
                                             uint32_t gid_hi, float* __restrict__ v) {
    constexpr int NB = reset_value_count<V>() / 4;
    const float4* rows = reinterpret_cast<const float4*>(rtab);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        uint32_t o[4];
        philox4x32_10(gid_lo, gid_hi, episode, (uint32_t)b, P.seed_lo, P.seed_hi, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[4 * b + k] = reset_value(rows[4 * b + k], o[k]);
    }
}
template <int V>
__device__ __forceinline__ void reset_env(const Params& P, const float* __restrict__ rtab, Env<V>& e, uint32_t gid_lo,
                                          uint32_t gid_hi) {
    float v[reset_value_count<V>()];
    reset_values<V>(P, rtab, e.episode, gid_lo, gid_hi, v);
    assign_reset<V>(e, v);
}

// Auto-reset from a per-lane STASH of the lane's own next reset draws (kernels that keep the env in registers over many steps and
// have the registers: 24 / 16 floats): a reset is a masked copy; the stash is refilled for the whole wave, and only when a lane that
// has used its stash up terminates again.  reset_values() for the lane's CURRENT episode = what reset_env() would draw: bit-identical.
// stash_od (E2E, optional): the disturbance observation columns of the stashed episode (Env::od), formed at refill time
template <int V>
__device__ __forceinline__ void reset_from_stash(const Params& P, const float* __restrict__ rtab, bool need, Env<V>& e,
                                                 uint32_t gid_lo, uint32_t gid_hi, float* __restrict__ stash, bool& stash_ok,
                                                 float* __restrict__ stash_od = nullptr) {
    if (__ballot(need) == 0ull) return;                 // wave-uniform
    if (__ballot(need && !stash_ok) != 0ull) {          // refill ALL lanes (a lane whose stash is intact recomputes the same values)
        reset_values<V>(P, rtab, e.episode, gid_lo, gid_hi, stash);
        if constexpr (V == kE2E) {
            if (stash_od) disturbance_obs_values(P, stash + 16, stash_od);
        }
        stash_ok = true;
    }
    if (need) {
        assign_reset<V>(e, stash);
        if constexpr (V == kE2E) {
            if (stash_od) {
#pragma unroll
                for (int c = 0; c < 4; ++c) e.od[c] = stash_od[c];
            }
        }
        stash_ok = false;
    }
}

// Auto-reset inside the step (kernels without the per-lane stash: the per-step kernel, the fused kernel at two workgroups per CU).
// Typically 0-2 of a wave's 64 envs terminate in a step -- but a launch ends with its SLOWEST wave, and among the 1 024 waves of a
// 65 536-env step a few have five or six.  Rounds 1-3 reset one done env at a time (readlane, ten dependent Philox rounds on eight
// lanes, an LDS exchange: ~780 cycles per done env, so the unluckiest wave of a step spent ~4 k cycles here).  Now the wave resets
// up to EIGHT envs per pass: lane j computes Philox block (j & 7) of the (j >> 3)-th done env of the pass -- the done lanes publish
// (lane id, episode) ranked by v_mbcnt through the wave's LDS tile -- and the 24 (16) values reach each env's lane through the
// same tile: one pass costs what one env used to, whatever the number of done envs up to eight.
// Same stream -- (seed, global env id, episode, block) -- and same arithmetic as reset_env(): bit-identical values.
// `tile`: >= 1 280 bytes of wave-private LDS that hold nothing live at the call.
template <int V>
__device__ __forceinline__ void reset_done_lanes(const Params& P, const float* __restrict__ rtab,
                                                 float* __restrict__ tile, int lane, bool done, Env<V>& e,
                                                 uint32_t gid_lo, uint32_t gid_hi) {
    constexpr int NB = (V == kE2E) ? 6 : 4;
    const unsigned long long pending = __ballot(done);
    if (pending == 0ull) return;                        // wave-uniform
    uint32_t* who = reinterpret_cast<uint32_t*>(tile);  // [8] (lane id | episode << 8) of the pass's envs
    float4* t4 = reinterpret_cast<float4*>(tile) + 16;  // [8 envs][8 blocks] reset values
    const int b = lane & 7, s = lane >> 3;
    const float4* rows = reinterpret_cast<const float4*>(rtab) + 4 * (b < NB ? b : 0);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pending >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pending, 0u));
    const int total = __builtin_popcountll(pending);
    // done lanes are active lanes, and an active lane's global id is (global id of the wave's lane 0) + lane
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane(gid_lo) - (uint32_t)__builtin_amdgcn_readfirstlane(lane);
    const uint32_t w_hi = __builtin_amdgcn_readfirstlane(gid_hi);
    for (int base = 0; base < total; base += 8) {       // wave-uniform; one pass unless more than eight envs finished
        const bool mine = done && rank >= base && rank < base + 8;
        if (mine) who[rank - base] = (uint32_t)lane | (e.episode << 8);   // the episode counter has 24 bits
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t info = who[s];                   // a slot past the pass's last env holds stale bits: computed on, never read
        const uint32_t g_lo = w_lo + (info & 0xFFu);
        const uint32_t g_hi = w_hi + (g_lo < w_lo ? 1u : 0u);
        uint32_t o[4];
        philox4x32_10(g_lo, g_hi, info >> 8, (uint32_t)b, P.seed_lo, P.seed_hi, o);
        if (b < NB)
            t4[8 * s + b] = make_float4(reset_value(rows[0], o[0]), reset_value(rows[1], o[1]), reset_value(rows[2], o[2]),
                                        reset_value(rows[3], o[3]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (mine) {
            float v[4 * NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float4 q = t4[8 * (rank - base) + j];
                v[4 * j + 0] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
            }
            assign_reset<V>(e, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Auto-reset from a POOL of reset draws in the wave's LDS (the fused kernel at two workgroups per CU: no registers for a per-lane
// stash).  reset_done_lanes() above runs one Philox pass in every step in which some env of the wave finishes -- 0.69 passes per step
// at the square track's termination rate, each serving 1.6 envs on average although it costs the same for eight.  Here a pass is only
// forced when an env finishes whose pool row is empty, and the slots such a pass has left over (8 - forced) fill the rows of OTHER
// lanes with the draws of THEIR current episode, ahead of time: a lane whose row is full resets with six 16-byte LDS reads.  A row holds
// reset_values(gid, episode) of the lane's current episode -- what reset_env() would draw -- and is invalidated by the reset that uses
// it (the episode counter moves on): bit-identical values, same stream.  Steady state: a pass every ~8 steps instead of every 1.45.
// `who`: 8 wave-private dwords; `pool`: [64 lanes][NB] float4, wave-private, persistent over the kernel; `ok`: this lane's row is full.
template <int V>
__device__ __forceinline__ void reset_pooled(const Params& P, const float* __restrict__ rtab, uint32_t* __restrict__ who,
                                             float4* __restrict__ pool, int lane, bool done, Env<V>& e, uint32_t gid_lo,
                                             uint32_t gid_hi, bool& ok) {
    constexpr int NB = (V == kE2E) ? 6 : 4;
    if (__ballot(done) == 0ull) return;                 // wave-uniform
    unsigned long long must = __ballot(done && !ok);
    if (must != 0ull) {
        const int b = lane & 7, s = lane >> 3;
        const float4* rows = reinterpret_cast<const float4*>(rtab) + 4 * (b < NB ? b : 0);
        // a served lane is an active lane or a lane that never resets: (global id of the wave's lane 0) + lane, as in reset_done_lanes()
        const uint32_t w_lo = __builtin_amdgcn_readfirstlane(gid_lo) - (uint32_t)__builtin_amdgcn_readfirstlane(lane);
        const uint32_t w_hi = __builtin_amdgcn_readfirstlane(gid_hi);
        do {                                            // wave-uniform; one pass unless more than eight empty rows are needed at once
            const unsigned long long spare = __ballot(!ok && !done);
            const int n_must = __builtin_popcountll(must);
            const bool forced = done && !ok;
            const unsigned long long below = forced ? must : spare;
            const int rank = (forced ? 0 : n_must) +
                             (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(below >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)below, 0u));
            const int total = n_must + __builtin_popcountll(spare);
            const int count = total < 8 ? total : 8;
            const bool mine = !ok && rank < 8;
            if (mine) who[rank] = (uint32_t)lane | (e.episode << 8);   // the episode counter has 24 bits
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t info = who[s];               // a slot past `count` holds stale bits: computed on, never stored
            const uint32_t tl = info & 0xFFu;
            const uint32_t g_lo = w_lo + tl;
            const uint32_t g_hi = w_hi + (g_lo < w_lo ? 1u : 0u);
            uint32_t o[4];
            philox4x32_10(g_lo, g_hi, info >> 8, (uint32_t)b, P.seed_lo, P.seed_hi, o);
            if (b < NB && s < count)
                pool[tl * NB + b] = make_float4(reset_value(rows[0], o[0]), reset_value(rows[1], o[1]), reset_value(rows[2], o[2]),
                                                reset_value(rows[3], o[3]));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (mine) ok = true;
            must = __ballot(done && !ok);
        } while (must != 0ull);
    }
    if (done) {
        float v[4 * NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float4 q = pool[lane * NB + j];
            v[4 * j + 0] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
        }
        assign_reset<V>(e, v);
        ok = false;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// -------------------------------------------------------------------------------------------------
// Rotation (ZYX Euler, R:97-100) -- shared by the residual-MLP input and the equations of motion
// -------------------------------------------------------------------------------------------------
struct Rot {
    float sph, cph, sth, cth, sps, cps;
    float r00, r10, r20, r01, r11, r21, r02, r12, r22;
};

// Branch-free float32 sincos: Cody-Waite reduction by pi/2 with a three-term split (FMA keeps each step exactly
// rounded), fdlibm's float minimax kernels on [-pi/4, pi/4], quadrant selection by k & 3.  Absolute error <= 7.3e-8
// for |x| <= 2e4 (checked against float64 in tools/check_sincos.py; NumPy's float32 sin/cos, which the reference
// evaluates, is at 6.2-7.0e-8 on the same inputs), ~25 instructions instead of the library's large-argument path.
// |psi| can grow while a drone spins, but |r| <= 1000 rad/s (out-of-bounds guard) for at most max_steps steps.
__device__ __forceinline__ void qr_sincos(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.6366197723675814f);
    float r = fmaf(-k, 1.5707963705062866f, x);       // pi/2 = 1.5707963705062866 - 4.371139000186241e-8 - 1.7151245100059e-15
    r = fmaf(-k, -4.371139000186241e-8f, r);
    r = fmaf(-k, -1.7151245100059e-15f, r);
    const float z = r * r;
    float sp = fmaf(z, 2.7183114939898219e-6f, -0.00019839334836563469f);
    sp = fmaf(z, sp, 0.0083333375930786133f);
    sp = fmaf(z, sp, -0.16666667163372040f);
    const float s = fmaf(r * z, sp, r);
    float cp = fmaf(z, 2.4390448796277409e-5f, -0.0013886763774609929f);
    cp = fmaf(z, cp, 0.041666623323739063f);
    cp = fmaf(z, cp, -0.49999999725103100f);
    const float c = fmaf(z, cp, 1.0f);
    const int q = (int)k;
    const float a = (q & 1) ? c : s;   // sin: s, c, -s, -c
    const float b = (q & 1) ? s : c;   // cos: c, -s, -c, s
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

__device__ __forceinline__ Rot make_rot(float phi, float theta, float psi) {
    Rot R;
    qr_sincos(phi, R.sph, R.cph);
    qr_sincos(theta, R.sth, R.cth);
    qr_sincos(psi, R.sps, R.cps);
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
// Residual thrust / moment MLPs: 7->32->1 and 10->32->3, ReLU (R:227-262).
//
// One lane = one env means every lane needs all 740 weights.  Feeding them as wave-uniform operands was measured on MI355X
// (N = 65 536, one wave per SIMD): broadcast ds_read_b128 from LDS is bandwidth-bound at 16 unique bytes per ~4.8 cycles per CU
// (5.9-8.5 k of the step kernel's 18 k wave cycles, pipelined or not), scalar loads expose ~24 scalar-cache round trips (8.3 k
// cycles).  The first layer is a genuine small GEMM per wave -- H^T[64 hidden x 64 envs] = W1[64 x 10(+bias)] * X^T -- so it runs
// on the matrix core, which also distributes each weight (held ONCE per wave) to all envs.
//
// Rounds 1-3 used v_mfma_f32_32x32x2_f32 (bit-exactly a k-ordered fmaf chain): 20 instructions of 64 cycles each that run at the
// f32 VECTOR rate and block the wave's VALU while they do (tools/ubench/valu_rate.hip: one f32 MFMA + 15 independent v_fma take
// 153 cycles, i.e. nothing overlaps) -- 1 500 of the fused step's 5 200 cycles.  Round 4: the same layer on
// v_mfma_f32_32x32x16_f16 (32 cycles per instruction on the separate matrix pipe) with BOTH operands split into two f16 pieces,
//     x = X0 + X1,  X0 = f16(x), X1 = f16(x - X0)         (x - X0 is exact in f32; |x - X0 - X1| <= 2^-22 |x|)
//     w = W0 + W1   (host side, once)
//     w x  ~  X0 W0 + X0 W1 + X1 W0                        (the dropped X1 W1 is <= 2^-22 |w x|)
// Every f16 x f16 product is exact in f32 and the matrix core adds the 16 products of an instruction and the accumulator with
// <= 3.3e-8 relative error of the sum of magnitudes (tools/ubench/mfma_f16_denorm.hip; f16 subnormals are honoured on both
// operands, so small inputs and weights lose nothing).  Against the float64 value of the layer the result is AS ACCURATE AS the f32
// fmaf chain it replaces: on the reference's 257 fixture rows (F1) max relative output error 1.4e-6 vs 1.3e-6 for the chain
// (tools/check_mlp_split.py; tests/test_gpu_round4.py pins both against float64).  33 products per hidden unit (thrust 7 x 3 + 2
// bias slots, moment 10 x 3 + 2) -> 5 instructions of 16 k-slots per 32-env tile, 10 per wave-step, 320 matrix-pipe cycles.
//   A (weights): lane l holds hidden row l&31, k-slots 8(l>>5)..+7 of instruction q (tabA, loop invariant, 5 x 4 registers)
//   B (inputs):  lane l holds k-slots 8(l>>5)..+7 of env tile*32 + (l&31): the lane-per-env quads (4 registers = 8 f16) of the
//                low and the high k-half meet in one v_permlane32_swap per register, which yields BOTH env tiles' operands
//   D: lane l, register r = hidden row (r&3) + 8*(r>>2) + 4*(l>>5) of env tile*32 + (l&31)   (as before)
// The 128 output-layer MACs stay on the VALU in that layout (each lane owns 16 hidden rows per tile) and the two wave halves are
// combined with v_permlane32_swap, which also returns every env's result to its own lane.
// -------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct MlpRegs {   // loop-invariant per-lane registers
    u32x4 a[kMlpQuads];  // tabA[q][lane] -- or, where registers decide the occupancy (the fused kernel at two workgroups per CU), read
    const u32x4* a_lds = nullptr;   // from an LDS copy of tabA every step: five conflict-free ds_read_b128 instead of 20 registers
    float w2[64];        // tabW2[lane>>5][..]
    float b2[4];
};
// stage tabA into LDS (kMlpQuads * 64 x 16 B) for MlpRegs::a_lds; the caller's barrier publishes it
__device__ __forceinline__ void mlp_stage_a(const float* __restrict__ tab, u32x4* __restrict__ lds_a) {
    const u32x4* src = reinterpret_cast<const u32x4*>(tab + kOffTabA);
    for (int i = threadIdx.x; i < kMlpQuads * 64; i += blockDim.x) lds_a[i] = src[i];
}

__device__ __forceinline__ void mlp_load_regs(const float* __restrict__ tab, int lane, MlpRegs& m, bool a_in_regs = true) {
    // MODE.DX10_CLAMP = 0 for the rest of the wave's life: the ReLU below is a CLAMP modifier, and with the (default) DX10 treatment a
    // NaN pre-activation would clamp to 0 -- the reference's torch ReLU propagates NaN (an env with a NaN state component gets NaN
    // residual forces and moments, fixture F11).  With the bit clear the clamp passes NaN through; finite values are unaffected.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0");
    if (a_in_regs) {
        const u32x4* a4 = reinterpret_cast<const u32x4*>(tab + kOffTabA) + lane;
#pragma unroll
        for (int q = 0; q < kMlpQuads; ++q) m.a[q] = a4[64 * q];
    }
    const float4* w4 = reinterpret_cast<const float4*>(tab + kOffTabW2 + 64 * (lane >> 5));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 w = w4[j];
        m.w2[4 * j + 0] = w.x; m.w2[4 * j + 1] = w.y; m.w2[4 * j + 2] = w.z; m.w2[4 * j + 3] = w.w;
    }
    const float4 b = *reinterpret_cast<const float4*>(tab + kOffB2);
    m.b2[0] = b.x; m.b2[1] = b.y; m.b2[2] = b.z; m.b2[3] = b.w;
}

// (a, b) lane-per-env  ->  lo = {a[0..31], b[0..31]} (operand of env tile 0), hi = {a[32..63], b[32..63]} (tile 1)
__device__ __forceinline__ void pair_to_tiles(float a, float b, float& lo, float& hi) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo = __uint_as_float(r.x);
    hi = __uint_as_float(r.y);
}
__device__ __forceinline__ void pair_to_tiles(uint32_t a, uint32_t b, uint32_t& lo, uint32_t& hi) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    lo = r.x;
    hi = r.y;
}

// two f32 -> one register of two f16, round to nearest even (v_cvt_pk_f16_f32), and back
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ void unpack_f16(uint32_t p, float& a, float& b) {
    const f16x2 h = __builtin_bit_cast(f16x2, p);
    a = (float)h[0];
    b = (float)h[1];
}

// ReLU of TWO accumulator rows in ONE instruction: v_pk_mul_f32 with the CLAMP modifier, clamp(x * 2^-40) in [0, 1].  For
// |x| < 2^40 the upper clamp never acts and the power-of-two scaling is exact, so the result is max(x, 0) * 2^-40 exactly; the
// output-layer weights are stored pre-multiplied by 2^40 (host side, exact), and the fused multiply-add rounds w' * t + s =
// w * max(x, 0) + s once -- bit for bit what fmaf(w, max(x, 0), s) gives.  (A hidden pre-activation of 1e12 does not occur: inputs
// are motor commands in [-1, 1], body velocities and rates the out-of-bounds guard keeps below 1000; a NaN pre-activation stays
// NaN like torch's ReLU: mlp_load_regs() clears MODE.DX10_CLAMP.)  There is no v_pk_max_f32 on gfx950: the v_max form cost 64 single-lane-pair instructions per env step, this
// costs 32.  fmaxf() would cost two each: LLVM first canonicalises an operand it cannot prove quiet.
// MFMA -> VALU read hazards: the wait states after a matrix instruction are inserted by the compiler's hazard recogniser,
// which does not look inside inline asm.  `ready` is the result of acc_ready(): a compiler-visible VALU read of the same
// accumulator (so the wait states are inserted before IT), and passing it in orders every relu2 of that accumulator behind it.
constexpr float kReluDown = 0x1p-40f, kReluUp = 0x1p40f;   // tabW2 holds W2 * kReluUp
__device__ __forceinline__ f32x2v relu2_scaled(f32x2v x, f32x2v down, uint32_t ready) {
    f32x2v r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(x), "v"(down), "s"(ready));
    return r;
}
template <class Acc>
__device__ __forceinline__ uint32_t acc_ready(const Acc& acc) {
    return (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(acc[0]));
}

// dot(w[0..15], relu(acc[0..15])) over this lane's 16 hidden rows as four independent chains (two packed pairs), in 4-row chunks
struct DotAcc {
    f32x2v s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
    // (s01.x + s23.x) + (s01.y + s23.y): one packed add and one scalar add.  The scalar add is inline asm on purpose: written in C++,
    // the SLP vectoriser pairs the final adds of DIFFERENT dot products into v_pk_add_f32 and pays for it with three v_mov per pair
    // (18 moves + 12 packed adds for the eight sums of a step; now 8 + 8 instructions)
    // (The add itself stays VISIBLE to the compiler -- its result feeds v_permlane32_swap, and a VALU write -> permlane read needs wait
    // states that the hazard recogniser only inserts for instructions it can see; as `asm("v_add_f32 ...")`, the first form, nothing
    // guaranteed them.  The empty asm only makes the RESULT opaque, which is enough to keep the vectoriser from pairing the adds.)
    __device__ __forceinline__ float sum() const {
        const f32x2v t = s01 + s23;
        float r = t.x + t.y;
        asm("" : "+v"(r));
        return r;
    }
};
__device__ __forceinline__ void dot_chunk(const float* w, const f32x16& acc, int r, DotAcc& d, uint32_t ready) {
    const f32x2v down = {kReluDown, kReluDown};
    const f32x2v a01 = {acc[r + 0], acc[r + 1]}, a23 = {acc[r + 2], acc[r + 3]};
    const f32x2v w01 = {w[r + 0], w[r + 1]}, w23 = {w[r + 2], w[r + 3]};
    d.s01 = __builtin_elementwise_fma(w01, relu2_scaled(a01, down, ready), d.s01);
    d.s23 = __builtin_elementwise_fma(w23, relu2_scaled(a23, down, ready), d.s23);
}

// kMode: 0 = layer-1 weight operands in registers (MlpRegs::a);  1 = operands re-read from LDS (MlpRegs::a_lds) every call -- the fused
//        form for two workgroups per CU, where 256 registers per wave are the budget.
template <int kMode = 0>
__device__ __forceinline__ void residual_mlp(const MlpRegs& m, int lane, const float x[10], float& thrust,
                                             float moment[3]) {
    constexpr bool kALds = (kMode == 1);
    // ---- split the inputs: P0 = f16 pairs of x, P1 = f16 pairs of x - X0 (exact difference) ----
    const uint32_t p0_01 = pack_f16(x[0], x[1]), p0_23 = pack_f16(x[2], x[3]), p0_45 = pack_f16(x[4], x[5]);
    const uint32_t p0_6o = pack_f16(x[6], 1.0f);   // k-slot 7 multiplies the bias
    const uint32_t p0_78 = pack_f16(x[7], x[8]), p0_9z = pack_f16(x[9], 0.0f);
    float h[12];
    unpack_f16(p0_01, h[0], h[1]); unpack_f16(p0_23, h[2], h[3]); unpack_f16(p0_45, h[4], h[5]);
    unpack_f16(p0_6o, h[6], h[10]); unpack_f16(p0_78, h[7], h[8]); unpack_f16(p0_9z, h[9], h[11]);
    float r[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) r[k] = x[k] - h[k];
    const uint32_t p1_01 = pack_f16(r[0], r[1]), p1_23 = pack_f16(r[2], r[3]), p1_45 = pack_f16(r[4], r[5]);
    const uint32_t p1_6z = pack_f16(r[6], 0.0f), p1_78 = pack_f16(r[7], r[8]), p1_9z = pack_f16(r[9], 0.0f);
    // ---- B operands of both env tiles: O1 = (X0 of inputs 0..6 + bias | X1 of inputs 0..6), O2 = (p, q, r terms, both halves) ----
    uint32_t a1[4], b1[4], a2[4], b2[4];   // [k-slot pair] of tile 0 (a) / tile 1 (b)
    pair_to_tiles(p0_01, p1_01, a1[0], b1[0]);
    pair_to_tiles(p0_23, p1_23, a1[1], b1[1]);
    pair_to_tiles(p0_45, p1_45, a1[2], b1[2]);
    pair_to_tiles(p0_6o, p1_6z, a1[3], b1[3]);
    pair_to_tiles(p0_78, p0_78, a2[0], b2[0]);
    pair_to_tiles(p0_9z, p0_9z, a2[1], b2[1]);
    pair_to_tiles(p1_78, p1_78, a2[2], b2[2]);
    pair_to_tiles(p1_9z, p1_9z, a2[3], b2[3]);
    const u32x4 o1[2] = {{a1[0], a1[1], a1[2], a1[3]}, {b1[0], b1[1], b1[2], b1[3]}};
    const u32x4 o2[2] = {{a2[0], a2[1], a2[2], a2[3]}, {b2[0], b2[1], b2[2], b2[3]}};
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    f32x16 hT0, hM0, hT1, hM1;
    u32x4 aq0, aq1, aq2, aq3, aq4;
    if constexpr (kALds) {
        int li = lane;
        asm volatile("" : "+v"(li));   // opaque per call: keeps the five reads INSIDE the step loop (hoisted, they are 20 registers again)
        aq0 = m.a_lds[li]; aq1 = m.a_lds[64 + li]; aq2 = m.a_lds[128 + li]; aq3 = m.a_lds[192 + li]; aq4 = m.a_lds[256 + li];
    } else {
        aq0 = m.a[0]; aq1 = m.a[1]; aq2 = m.a[2]; aq3 = m.a[3]; aq4 = m.a[4];
    }
#define QR_MFMA(ACC, Q, B, C) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq##Q), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)
    // ten instructions on four accumulators, round robin: an accumulator's next instruction is three others (>= 96 cycles) away
    QR_MFMA(hT0, 0, o1[0], zero); QR_MFMA(hM0, 2, o1[0], zero); QR_MFMA(hT1, 0, o1[1], zero); QR_MFMA(hM1, 2, o1[1], zero);
    // instructions 1 / 3 (the W1 pieces) read the same operand: their high k-half weights are zero
    QR_MFMA(hT0, 1, o1[0], hT0); QR_MFMA(hM0, 3, o1[0], hM0); QR_MFMA(hT1, 1, o1[1], hT1); QR_MFMA(hM1, 3, o1[1], hM1);
    QR_MFMA(hM0, 4, o2[0], hM0); QR_MFMA(hM1, 4, o2[1], hM1);
#undef QR_MFMA
    // (Tried in round 4 and dropped: sched_group_barrier patterns "one MFMA, then 4 / 7 VALU instructions" to pull the MLP-independent
    // part of the step between the ten matrix instructions -- 4 176 instead of 3 997 cycles per fused step: the compiler's own
    // back-to-back issue with the VALU work behind it is the better schedule here.)
    // (Rounds 4's "guard" -- 32 wait states and operand keep-alives behind the matrix block for kernels with two waves per SIMD -- is gone:
    // the corruption it papered over was not in this block at all.  MI355X computes a packed-f32 instruction whose second source feeds
    // its HIGH dword to the LOW result half wrongly in lanes 48-63 when another wave of the SIMD issues an f16 matrix instruction at
    // the wrong moment; the SLP vectoriser had produced one such instruction in the equations of motion, and the guard merely shifted
    // the two waves' phase.  The build now rewrites that instruction form everywhere: isa_lint.py)
    DotAcc dT0, dT1, dM0[3], dM1[3];
    const uint32_t rT0 = acc_ready(hT0);
#pragma unroll
    for (int q = 0; q < 16; q += 4) dot_chunk(m.w2 + 0, hT0, q, dT0, rT0);
    const uint32_t rT1 = acc_ready(hT1);
#pragma unroll
    for (int q = 0; q < 16; q += 4) dot_chunk(m.w2 + 0, hT1, q, dT1, rT1);
    const uint32_t rM0 = acc_ready(hM0);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < 16; q += 4) dot_chunk(m.w2 + 16 + 16 * g, hM0, q, dM0[g], rM0);
    const uint32_t rM1 = acc_ready(hM1);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int q = 0; q < 16; q += 4) dot_chunk(m.w2 + 16 + 16 * g, hM1, q, dM1[g], rM1);
    float part[2][4];
    part[0][0] = dT0.sum();
    part[1][0] = dT1.sum();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        part[0][1 + g] = dM0[g].sum();
        part[1][1 + g] = dM1[g].sum();
    }
    float out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {  // lanes 0..31: tile-0 halves; lanes 32..63: tile-1 halves -> env = lane
        float lo, hi;
        pair_to_tiles(part[0][o], part[1][o], lo, hi);
        out[o] = (lo + hi) + m.b2[o];
    }
    thrust = out[0];
    moment[0] = out[1];
    moment[1] = out[2];
    moment[2] = out[3];
}

// 1/x and sqrt(x) as hardware approximation + one Newton step (error <= 1 ulp; the library forms spend ~10
// instructions on exact rounding and denormal scaling that this path cannot use: cos(theta) and squared distances
// are far from the denormal range; at theta = +-pi/2 both forms give inf/NaN exactly like the reference's 1/cos).
__device__ __forceinline__ float fast_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float fast_sqrt(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(y);
    const float c = fmaf(fmaf(-y, y, x), h, y);      // y + (x - y*y) / (2y)
    return x > 0.0f ? c : y;                         // sqrt(0) = 0 (avoid 0 * inf)
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
    const float inv_cth = fast_rcp(R.cth);
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
    const float inv_cth = fast_rcp(R.cth);
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
// the target gate's table row as a step / an observation uses it, and the rows of the gates ahead
struct GateRow {
    float4 g0;   // x y z yaw
    float2 cs;   // cos(yaw) sin(yaw)
};
__device__ __forceinline__ GateRow read_gate_row(const float* __restrict__ gates, int target) {
    GateRow g;
    const float* row = gates + __mul24(kGateStride, target);
    g.g0 = *reinterpret_cast<const float4*>(row);
    g.cs = *reinterpret_cast<const float2*>(row + 4);
    return g;
}
// relative position / yaw of the GA gates after `target` (R:406-412), indices modulo the gate count
template <int GA>
__device__ __forceinline__ void read_gates_ahead(const Params& P, const float* __restrict__ gates, int target, float4* rel) {
#pragma unroll
    for (int a = 0; a < GA; ++a) {
        int idx = target + a + 1;
        // target < num_gates, so one conditional subtraction suffices unless the track has fewer gates than a + 1 (wave-uniform)
        if (idx >= P.num_gates) idx -= P.num_gates;
        if (P.num_gates <= a) {
            while (idx >= P.num_gates) idx -= P.num_gates;
        }
        rel[a] = *reinterpret_cast<const float4*>(gates + __mul24(kGateStride, idx) + 8);
    }
}

template <int V, int GA, bool kCachedOd = false>
__device__ __forceinline__ void observe_with(const Params& P, const GateRow& g, const float4* rel, const Env<V>& e, float* o) {
    constexpr int S = Env<V>::S;
    const float4 g0 = g.g0;
    const float2 cs = g.cs;
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
        o[S + 4 * a + 0] = rel[a].x;
        o[S + 4 * a + 1] = rel[a].y;
        o[S + 4 * a + 2] = rel[a].z;
        o[S + 4 * a + 3] = rel[a].w;
    }
    if constexpr (V == kE2E) {  // R:414-448: (Mx, My, Mz, Fz) mapped to [-1,1] by their ranges
        constexpr int base = S + 4 * GA;
        if constexpr (kCachedOd) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[base + c] = e.od[c];
        } else {
            disturbance_obs_values(P, e.d, o + base);
        }
    }
}
template <int V, int GA>
__device__ __forceinline__ void observe(const Params& P, const float* __restrict__ gates, const Env<V>& e,
                                        float* o) {
    const GateRow g = read_gate_row(gates, e.target);
    float4 rel[GA > 0 ? GA : 1];
    read_gates_ahead<GA>(P, gates, e.target, rel);
    observe_with<V, GA>(P, g, rel, e, o);
}

// -------------------------------------------------------------------------------------------------
// One env step (step_wait, R:501-595 / I:303-385).  Returns reward; sets done / trunc flags.
// On auto-reset `did_reset` is set so the caller persists the new disturbances.  Must be called by all 64 lanes.
// -------------------------------------------------------------------------------------------------
// `before_reset(done)` is invoked (all lanes) after the state update and before the auto-reset -- the caller's hook
// for the terminal observation SB3 bootstraps time-limit truncations from (R:589-594).
// `do_reset(need)` performs the auto-reset of the lanes with `need` (all lanes call it); the default is reset_done_lanes().
// the arithmetic of one step from the pre-step state: new state `nw`, reward, flags, the target after the step
template <int V, int kMode = 0>
__device__ __forceinline__ float step_dynamics(const Params& P, const GateRow& gate, const MlpRegs& mlp, bool use_mlp, int lane,
                                               const Env<V>& e, const float u[4], float* nw, int& new_target, bool& done,
                                               bool& trunc) {
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
        if (use_mlp) {  // R:502-509: residual evaluated on the PRE-step state
            const float x[10] = {e.s[12], e.s[13], e.s[14], e.s[15], vb[0], vb[1], vb[2], e.s[9], e.s[10], e.s[11]};
            float thrust, moment[3];
            residual_mlp<kMode>(mlp, lane, x, thrust, moment);
            M[0] += moment[0]; M[1] += moment[1]; M[2] += moment[2];
            F[2] += thrust;
        }
        QR_TICK(P, 4);
        eom_e2e(e.s, R, vb, u, M, F, ds);
    } else {
        eom_indi(e.s, R, vb, u, ds);
    }
#pragma unroll
    for (int k = 0; k < S; ++k) nw[k] = fmaf(P.dt, ds[k], e.s[k]);  // forward Euler, R:512
    const int steps = e.steps + 1;                                   // R:514

    const float4 g0 = gate.g0;                                       // R:518-519
    const float2 cs = gate.cs;
    const float ox = e.s[0] - g0.x, oy = e.s[1] - g0.y, oz = e.s[2] - g0.z;
    const float nx = nw[0] - g0.x, ny = nw[1] - g0.y, nz = nw[2] - g0.z;
    const float d2g_old = fast_sqrt(fmaf(ox, ox, fmaf(oy, oy, oz * oz)));  // R:522-525
    const float d2g_new = fast_sqrt(fmaf(nx, nx, fmaf(ny, ny, nz * nz)));
    // R:524-525: rewards = d2g_old - d2g_new - rat_penalty with rat_penalty = 0 * 0.01 * |new rates| -- zero for finite rates, NaN when a
    // new rate is NaN or inf (0 * NaN, 0 * inf): the reward of an env with NaN attitude or rates is NaN although its position is finite
    // (fixture F11).  0 * (sum of squares) is NaN / zero under exactly the same conditions as 0 * the float32 norm.
    const float rat_penalty = 0.0f * fmaf(nw[9], nw[9], fmaf(nw[10], nw[10], nw[11] * nw[11]));
    float reward = (d2g_old - d2g_new) - rat_penalty;
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
    new_target = e.target;
    if (gate_passed) new_target = (e.target + 1 == P.num_gates) ? 0 : e.target + 1;    // R:556-557
    done = trunc || ground || gate_collision || oob;                // R:566
    return reward;
}

template <int V, int kMode = 0, class BeforeReset, class DoReset>
__device__ __forceinline__ float step_env(const Params& P, const float* __restrict__ gates,
                                          const float* __restrict__ rtab, float* __restrict__ tile, const MlpRegs& mlp,
                                          int lane, bool active, Env<V>& e, const float u[4], uint32_t gid_lo,
                                          uint32_t gid_hi, bool& done, bool& trunc, bool& did_reset,
                                          BeforeReset&& before_reset, DoReset&& do_reset) {
    constexpr int S = Env<V>::S;
    float nw[S];
    int new_target;
    const GateRow gate = read_gate_row(gates, e.target);
    const float reward = step_dynamics<V, kMode>(P, gate, mlp, (V == kE2E) && (P.flags & kFlagResidual), lane, e, u, nw, new_target, done, trunc);
    e.target = new_target;
    e.steps = e.steps + 1;
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
        did_reset = done;
        before_reset(done);
        do_reset(done && active);  // shadow lanes are not reset
    }
    return reward;
}
template <int V, int kMode = 0, class BeforeReset>
__device__ __forceinline__ float step_env(const Params& P, const float* __restrict__ gates,
                                          const float* __restrict__ rtab, float* __restrict__ tile, const MlpRegs& mlp,
                                          int lane, bool active, Env<V>& e, const float u[4], uint32_t gid_lo,
                                          uint32_t gid_hi, bool& done, bool& trunc, bool& did_reset,
                                          BeforeReset&& before_reset) {
    return step_env<V, kMode>(P, gates, rtab, tile, mlp, lane, active, e, u, gid_lo, gid_hi, done, trunc, did_reset,
                       static_cast<BeforeReset&&>(before_reset),
                       [&](bool need) { reset_done_lanes<V>(P, rtab, tile, lane, need, e, gid_lo, gid_hi); });
}

}  // namespace qr
