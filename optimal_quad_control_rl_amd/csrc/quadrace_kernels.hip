// quadrace_kernels.hip -- gfx950 kernels: fused env step, reset, observe, state import/export.
//
// Launch shape: 1 lane = 1 env, 256-thread workgroups (4 wave64).  At N = 65 536 that is 256 workgroups
// = one per CU; larger N simply adds workgroups (block b lands on XCD b % 8, and consecutive blocks touch
// consecutive 4 KiB slabs of every plane, so each XCD's L2 sees disjoint, fully-used lines).
// Per workgroup the gate table (indexed per lane by the env's target gate) is staged once into LDS; each wave
// also owns an LDS tile for coalesced observation stores.  Residual-MLP weights live in registers (layer 1: five MFMA A operands
// of eight f16 per lane; layer 2: 64 floats per lane half; see quadrace_device.hpp).
#include "quadrace_device.hpp"
#include "quadrace_policy.hpp"

namespace qr {

// copy `count` floats starting at float offset `src_off` of the device table image [MLP table | gate rows] to lds[0..)
__device__ __forceinline__ void stage_tables(const Params& P, float* lds, int src_off, int count) {
    // offsets / counts are multiples of 4; the tables pointer is 16-byte aligned
    const float4* src = reinterpret_cast<const float4*>(P.tables + src_off);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < count / 4; i += kBlock) dst[i] = src[i];
}

template <int V>
__device__ __forceinline__ void load_env(const Params& P, int i, Env<V>& e) {
    const float4 a = P.ws[i], b = P.ws[P.n_stride + i], c = P.ws[2 * P.n_stride + i];
    e.s[0] = a.x; e.s[1] = a.y; e.s[2] = a.z; e.s[3] = a.w;
    e.s[4] = b.x; e.s[5] = b.y; e.s[6] = b.z; e.s[7] = b.w;
    e.s[8] = c.x; e.s[9] = c.y; e.s[10] = c.z; e.s[11] = c.w;
    if constexpr (V == kE2E) {
        const float4 d = P.ws[3 * P.n_stride + i];
        e.s[12] = d.x; e.s[13] = d.y; e.s[14] = d.z; e.s[15] = d.w;
        const float4 dA = P.dA[i];
        const float2 dB = P.dB[i];
        e.d[0] = dA.x; e.d[1] = dA.y; e.d[2] = dA.z; e.d[5] = dA.w;
        e.d[3] = dB.x; e.d[4] = dB.y;
    } else {
        e.s[12] = P.tn[i];
    }
    const int2 ts = P.ts[i];
    e.target = ts.x & 0xFF;
    e.episode = (uint32_t)ts.x >> 8;
    e.steps = ts.y;
}

template <int V>
__device__ __forceinline__ int2 pack_ts(const Env<V>& e) {
    return make_int2((int)((uint32_t)e.target | (e.episode << 8)), e.steps);
}

template <int V>
__device__ __forceinline__ void store_world(const Params& P, int i, const Env<V>& e) {
    stream_store(P.ws + i, make_float4(e.s[0], e.s[1], e.s[2], e.s[3]));
    stream_store(P.ws + P.n_stride + i, make_float4(e.s[4], e.s[5], e.s[6], e.s[7]));
    stream_store(P.ws + 2 * P.n_stride + i, make_float4(e.s[8], e.s[9], e.s[10], e.s[11]));
    if constexpr (V == kE2E) {
        stream_store(P.ws + 3 * P.n_stride + i, make_float4(e.s[12], e.s[13], e.s[14], e.s[15]));
    } else {
        stream_store(P.tn + i, e.s[12]);
    }
}

template <int V>
__device__ __forceinline__ void store_dist(const Params& P, int i, const Env<V>& e) {
    if constexpr (V == kE2E) {
        stream_store(P.dA + i, make_float4(e.d[0], e.d[1], e.d[2], e.d[5]));
        stream_store(P.dB + i, make_float2(e.d[3], e.d[4]));
    }
}

// The final state of a K-step kernel leaves in 16-byte tuples.  Left alone, the compiler forms those tuples INSIDE the step loop: the
// loop's exit values (a phi of the reset / no-reset paths) were copied into four register quads on every step -- 18 moves of ~700
// instructions, 60 on a step with a reset -- for a store that happens once per launch.  Passing the values through an empty asm
// defines them behind the loop (round 6: rollout_fast_mlp_kernel 701 -> 686 instructions per step, 277 -> 163 v_mov in the kernel).
template <int V>
__device__ __forceinline__ void define_exit_values(Env<V>& e) {
#pragma unroll
    for (int q = 0; q < Env<V>::S; ++q) asm volatile("" : "+v"(e.s[q]));
    if constexpr (V == kE2E) {
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("" : "+v"(e.d[q]));
    }
}

template <int V, int GA>
constexpr int obs_len() { return (V == kE2E) ? 16 + 4 * GA + 4 : 13 + 4 * GA; }

// obs row -> caller's row-major [N][L] buffer.  E2E rows (20+4*GA floats) are 16-byte aligned.
template <int V, int GA>
__device__ __forceinline__ void store_obs(float* __restrict__ obs_out, int i, const float* o) {
    constexpr int L = obs_len<V, GA>();
    float* row = obs_out + (size_t)i * L;
    if constexpr (V == kE2E) {
        float4* r4 = reinterpret_cast<float4*>(row);
#pragma unroll
        for (int k = 0; k < L / 4; ++k) stream_store(r4 + k, make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]));
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) stream_store(row + k, o[k]);
    }
}

// The 64 observation rows of a full wave are one contiguous [64][L] block of the caller's row-major buffer.
// Per-lane row stores would scatter 16-byte pieces over 64 different cache lines per instruction, so the wave
// transposes through a wave-private LDS tile and writes the block with fully coalesced 16-byte-per-lane stores
// (1 KiB per instruction).  LDS operations of one wave execute in order; the fences only pin the compiler.
template <int V, int GA>
__device__ __forceinline__ void obs_tile_write(float* __restrict__ tile, int lane, const float* o) {
    constexpr int L = obs_len<V, GA>();
    float* row = tile + lane * L;
    if constexpr (L % 4 == 0) {
        float4* r4 = reinterpret_cast<float4*>(row);
#pragma unroll
        for (int k = 0; k < L / 4; ++k) r4[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) row[k] = o[k];  // odd row stride: conflict-free ds_write_b32
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int V, int GA>
__device__ __forceinline__ void obs_tile_flush(const float* __restrict__ tile, float* __restrict__ obs_out, size_t wave_first_env,
                                               int lane) {
    constexpr int L = obs_len<V, GA>();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int kVec = 16 * L;  // float4 elements in the block (64*L floats; 64*L*4 bytes is a multiple of 16)
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    float4* g4 = reinterpret_cast<float4*>(obs_out + wave_first_env * L);
#pragma unroll
    for (int t = 0; t < (kVec + 63) / 64; ++t) {
        const int e = t * 64 + lane;
        if ((t + 1) * 64 <= kVec || e < kVec) stream_store(g4 + e, t4[e]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int V, int GA>
__device__ __forceinline__ void store_obs_coalesced(float* __restrict__ tile, float* __restrict__ obs_out,
                                                    size_t wave_first_env, int lane, const float* o) {
    obs_tile_write<V, GA>(tile, lane, o);
    obs_tile_flush<V, GA>(tile, obs_out, wave_first_env, lane);
}

// Terminal observation (optional, Params::term_obs): the gate-frame observation of the final state of an episode, written
// before the auto-reset replaces that state.  Finished envs are rare (~1 % of the lanes per step), so the divergent
// observe + row store costs next to nothing; rows of envs that did not finish are left untouched.
template <int V, int GA>
__device__ __forceinline__ void store_terminal_obs(const Params& P, const float* __restrict__ gates, const Env<V>& e,
                                                   size_t row_base, int i, bool write) {
    if (P.term_obs == nullptr || !write) return;
    float to[obs_len<V, GA>()];
    observe<V, GA>(P, gates, e, to);
    store_obs<V, GA>(P.term_obs + row_base * obs_len<V, GA>(), i, to);
}

#ifndef QR_FULL_OK
#define QR_FULL_OK true   /* -DQR_FULL_OK=false: A/B build without the full-grid copies */
#endif
// ---------------------------------------------------------------------------------------------------
// Fused step: residual MLP -> EoM -> Euler -> reward/termination -> auto-reset -> gate-frame observation
// ---------------------------------------------------------------------------------------------------
// kFull: every workgroup of the launch is full (n a multiple of the workgroup size) -- `active` is compile-time true and the EXEC-mask
// sequences of the ragged tail leave the code; step_kernel holds both copies behind a launch-uniform branch (see rollout_fast_body).
template <int V, int GA, bool kFull>
__device__ __forceinline__ void step_body(Params P, const float4* __restrict__ actions, float* __restrict__ obs_out,
                                          float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out,
                                          float* __restrict__ lds) {
    // The MLP table is staged through LDS with the reset / gate rows (two 16-byte loads per thread), then 22 LDS reads per lane fill the
    // weight registers behind the barrier.  (Round 4 A/B, same box: loading the registers straight from global memory instead -- 22
    // loads per lane through the texture path -- costs 0.84 us per launch, 6.60 vs 5.77 us; that form is gone.)
    constexpr int kTab = (V == kE2E) ? kMlpTableFloats : 0;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // Lanes past the end of a ragged batch stay ACTIVE (they shadow env 0) because the residual MLP uses
    // wave-wide operations (MFMA, permlane swap); only their stores are suppressed.
    const bool active = kFull || i < P.n;
    const int ii = active ? i : 0;
    QR_TICK(P, 0);
    // Prologue ordering (one wave per SIMD at N = 65 536: every exposed latency is paid in full).  Loads return in
    // issue order (one vmcnt counter), so the table loads -- L2 hits, needed first: they go through LDS and a
    // workgroup barrier -- are issued BEFORE the lane's state loads (HBM round trip): the LDS writes, the barrier and
    // the 22 LDS reads that fill the residual-MLP weight registers all complete in the shadow of the state loads.
    const bool use_mlp = (V == kE2E) && (P.flags & kFlagResidual);
    float* rtab = lds + kTab;                 // [reset table | gate rows | obs tiles]
    float* gates = rtab + kResetTableFloats;
    const int tab_off = use_mlp ? 0 : kOffResetImage;
    const int tab_vec = ((use_mlp ? kOffGatesImage : kResetTableFloats) + P.num_gates * kGateStride) / 4;  // <= 476
    const float4* tsrc = reinterpret_cast<const float4*>(P.tables + tab_off);
    float4* tdst = reinterpret_cast<float4*>(use_mlp ? lds : rtab);
    const int t0 = threadIdx.x, t1 = threadIdx.x + kBlock;
    const float4 tv0 = tsrc[t0 < tab_vec ? t0 : 0];
    const float4 tv1 = tsrc[t1 < tab_vec ? t1 : 0];
    Env<V> e;
    load_env<V>(P, ii, e);
    const float4 act = actions[ii];
    QR_TICK(P, 1);
    tdst[t0] = tv0;
    tdst[t1] = tv1;
    __syncthreads();
    MlpRegs mlp;
    if (use_mlp) mlp_load_regs(lds, lane, mlp);
    float* tile = gates + kMaxGates * kGateStride + (threadIdx.x >> 6) * 64 * obs_len<V, GA>();
    QR_TICK(P, 2);
    const float u[4] = {act.x, act.y, act.z, act.w};
    const uint32_t gid_lo = P.gid_lo + (uint32_t)ii;
    const uint32_t gid_hi = P.gid_hi + (gid_lo < P.gid_lo ? 1u : 0u);
    bool done, trunc, did_reset;
    const float reward = step_env<V>(P, gates, rtab, tile, mlp, lane, active, e, u, gid_lo, gid_hi, done, trunc, did_reset,
                                        [&](bool fin) { store_terminal_obs<V, GA>(P, gates, e, 0, i, fin && active); });
    if (active) {
        stream_store(rew_out + i, reward);
        stream_store(done_out + i, (uint8_t)(done ? 1 : 0));
        if (trunc_out) stream_store(trunc_out + i, (uint8_t)(trunc ? 1 : 0));
        stream_store(P.ts + i, pack_ts<V>(e));
    }
    if (P.flags & kFlagPause) return;  // world state and observation untouched (R:570-572)
    QR_TICK(P, 6);
    if (active) {
        store_world<V>(P, i, e);
        if (did_reset) store_dist<V>(P, i, e);
    }
    float o[obs_len<V, GA>()];
    observe<V, GA>(P, gates, e, o);
    const int wave_first = i - lane;
    if (wave_first + 64 <= P.n) {  // full wave (wave-uniform): coalesced block store through the LDS tile
        store_obs_coalesced<V, GA>(tile, obs_out, (size_t)wave_first, lane, o);
    } else if (active) {
        store_obs<V, GA>(obs_out, i, o);
    }
    QR_TICK(P, 7);
}

// ---------------------------------------------------------------------------------------------------
// Fused K-step rollout (qr_step_many): the same step_env() applied K times with the env state (and the MLP
// weights) held in registers.  Per step a lane only needs its action and writes obs / reward / done; there is no
// launch boundary, state round trip or end-of-kernel L2 write-back per step.  Bit-identical to K x step_kernel.
//
// Actions are staged kActChunk steps at a time into a lane-private LDS slot.  On gfx9-family hardware loads and
// stores share one in-order counter (vmcnt), so ANY global load inside the step loop makes the wave wait for the
// previous step's stores to be acknowledged (measured: the loop ran at one store round trip, ~1.8 us, per step
// with ~300 instructions in it).  With the loads hoisted to one burst per chunk, the per-step stores simply
// stream out behind the arithmetic.
// ---------------------------------------------------------------------------------------------------
template <int V, int GA>
constexpr int act_chunk() {  // steps of actions staged per burst, sized so the static LDS stays <= 64 KiB
    return (65536 - 4 * (kResetTableFloats + kMaxGates * kGateStride) - 4 * kBlock * obs_len<V, GA>() - 16 * kMlpQuads * 64) / (16 * kBlock) >= 8 ? 8 : 4;
}
template <int V, int GA>
__global__ void __launch_bounds__(kBlock)
step_kernel(Params P, const float4* __restrict__ actions, float* __restrict__ obs_out,
            float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    constexpr int kTab = (V == kE2E) ? kMlpTableFloats : 0;
    __shared__ __attribute__((aligned(16))) float lds[kTab + kResetTableFloats + kMaxGates * kGateStride + kBlock * obs_len<V, GA>()];
    if (QR_FULL_OK && P.n % kBlock == 0) step_body<V, GA, true>(P, actions, obs_out, rew_out, done_out, trunc_out, lds);
    else step_body<V, GA, false>(P, actions, obs_out, rew_out, done_out, trunc_out, lds);
}

// kStash (round 3; launches with at most one workgroup per CU, where the register budget is free): every lane keeps the draws of
// ITS OWN next reset -- 24 (16) floats: Philox blocks 0..5 (0..3) of (seed, global env id, current episode) -- and an auto-reset
// is a masked register copy.  The stash is refilled for all 64 lanes at once, and only when a lane that has used its stash up
// terminates again (about every 14 steps at a 1.2 % termination rate), instead of the wave walking its done lanes one by one
// through reset_done_lanes() in 56 % of the steps: measured 0.31 us of a 2.80 us step (tools probe: 2.49 us with resets off,
// +0.21 us per per cent of terminating lanes).  Same stream, same arithmetic as reset_env(): bit-identical.
template <int V, int GA, bool kStash, bool kFull>
__device__ __forceinline__ void rollout_body_impl(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                                                  float* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                                                  uint8_t* __restrict__ trunc_out, float* __restrict__ lds) {
    constexpr int kActChunk = act_chunk<V, GA>();
    constexpr int L = obs_len<V, GA>();
    // the plain form runs two workgroups per CU, where 256 registers is the limit: the layer-1 weight operands (20 registers) live in
    // LDS there and are re-read every step (MlpRegs::a_lds); the stash form has the register file of a whole SIMD per wave
    constexpr bool kALds = (V == kE2E) && !kStash;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = kFull || i < P.n;  // ragged tail lanes stay active (wave-wide MLP ops) and shadow env 0
    const int ii = active ? i : 0;
    QR_CLOCK_STAMP(P, 0);
    QR_CLOCK_HWID(P);
    Env<V> e;
    load_env<V>(P, ii, e);
    MlpRegs mlp;  // weights stay in registers for all K steps
    const bool use_mlp = (V == kE2E) && (P.flags & kFlagResidual);
    if (use_mlp) mlp_load_regs(P.tables, lane, mlp, !kALds);
    float* rtab = lds;                        // [reset table | gate rows | obs tiles | action slots | layer-1 A operands]
    float* gates = lds + kResetTableFloats;
    stage_tables(P, lds, kOffResetImage, kResetTableFloats + P.num_gates * kGateStride);
    if constexpr (kALds) {
        u32x4* lds_a = reinterpret_cast<u32x4*>(lds + kResetTableFloats + kMaxGates * kGateStride + kBlock * L + 4 * kBlock * kActChunk);
        if (use_mlp) mlp_stage_a(P.tables, lds_a);
        mlp.a_lds = lds_a;
    }
    __syncthreads();
    const uint32_t gid_lo = P.gid_lo + (uint32_t)ii;
    const uint32_t gid_hi = P.gid_hi + (gid_lo < P.gid_lo ? 1u : 0u);
    const size_t n = (size_t)P.n;
    const int wave_first = i - lane;
    const bool full_wave = kFull || wave_first + 64 <= P.n;
    float* tile = gates + kMaxGates * kGateStride + (threadIdx.x >> 6) * 64 * L;
    // lane-private action slots: element (j, thread) at [j * kBlock + threadIdx.x] (consecutive lanes = consecutive
    // 16 B, conflict-free); each lane only reads back what it wrote itself
    float4* act_slot = reinterpret_cast<float4*>(gates + kMaxGates * kGateStride + kBlock * L) + threadIdx.x;
    bool any_reset = false;
    float stash[kStash ? reset_value_count<V>() : 1];
    bool stash_ok = false;
    QR_CLOCK_STAMP(P, 1);
    for (int k0 = 0; k0 < K; k0 += kActChunk) {
        const int c = (K - k0 < kActChunk) ? K - k0 : kActChunk;
        float4 burst[kActChunk];  // all loads first (clamped step index keeps them unconditional), then the LDS writes
#pragma unroll
        for (int j = 0; j < kActChunk; ++j) {
            const int kk = (k0 + j < K) ? k0 + j : K - 1;
            burst[j] = actions[(size_t)kk * n + ii];
        }
#pragma unroll
        for (int j = 0; j < kActChunk; ++j) act_slot[j * kBlock] = burst[j];
        for (int j = 0; j < c; ++j) {
            const int k = k0 + j;
#ifdef QR_PHASE_TIMING
            P.tick_on = (k == K / 2);
#endif
            QR_TICK(P, 2);
            const float4 act = act_slot[j * kBlock];
            const float u[4] = {act.x, act.y, act.z, act.w};
            bool done, trunc, did_reset;
            const float reward = step_env<V, kALds ? 1 : 0>(P, gates, rtab, tile, mlp, lane, active, e, u, gid_lo, gid_hi, done, trunc,
                                             did_reset, [&](bool fin) {
                                                 store_terminal_obs<V, GA>(P, gates, e, (size_t)k * n, i, fin && active);
                                             }, [&](bool need) {
                                                 if constexpr (kStash) {
                                                     reset_from_stash<V>(P, rtab, need, e, gid_lo, gid_hi, stash, stash_ok);
                                                 } else {
                                                     reset_done_lanes<V>(P, rtab, tile, lane, need, e, gid_lo, gid_hi);
                                                 }
                                             });
            any_reset |= did_reset;
            if (active) {
                stream_store(rew_out + (size_t)k * n + i, reward);
                stream_store(done_out + (size_t)k * n + i, (uint8_t)(done ? 1 : 0));
                if (trunc_out) stream_store(trunc_out + (size_t)k * n + i, (uint8_t)(trunc ? 1 : 0));
            }
            QR_TICK(P, 6);
            if (!(P.flags & kFlagPause)) {
                float o[L];
                observe<V, GA>(P, gates, e, o);
                if (full_wave) store_obs_coalesced<V, GA>(tile, obs_out + (size_t)k * n * L, (size_t)wave_first, lane, o);
                else if (active) store_obs<V, GA>(obs_out + (size_t)k * n * L, i, o);
            }
            QR_TICK(P, 7);
        }
    }
    QR_CLOCK_STAMP(P, 2);
    if (!active) return;
    define_exit_values<V>(e);
    P.ts[i] = pack_ts<V>(e);
    if (P.flags & kFlagPause) return;
    store_world<V>(P, i, e);
    if (any_reset) store_dist<V>(P, i, e);
    QR_CLOCK_STAMP(P, 3);
}

template <int V, int GA, bool kStash>
__device__ __forceinline__ void rollout_body(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                                             float* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                                             uint8_t* __restrict__ trunc_out) {
    constexpr bool kALds = (V == kE2E) && !kStash;
    __shared__ __attribute__((aligned(16))) float lds[kResetTableFloats + kMaxGates * kGateStride + kBlock * obs_len<V, GA>() +
                                                       4 * kBlock * act_chunk<V, GA>() + (kALds ? 4 * kMlpQuads * 64 : 0)];
    if (QR_FULL_OK && P.n % kBlock == 0) rollout_body_impl<V, GA, kStash, true>(P, K, actions, obs_out, rew_out, done_out, trunc_out, lds);
    else rollout_body_impl<V, GA, kStash, false>(P, K, actions, obs_out, rew_out, done_out, trunc_out, lds);
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the fused rollout for the DEFAULT mode (no pause flags, no terminal-observation buffer; residual on / off is a
// template parameter), launches with at most one workgroup per CU.  Same step_dynamics() / observe_with() / reset_from_stash()
// as every other kernel -- bit-identical results -- but the loop is one basic block as far as the modes go, and the data
// movement around the arithmetic is re-planned for a wave that has its SIMD to itself (every exposed latency is paid in full):
//   * the actions of chunk c + 1 are requested at the top of chunk c into registers (the wave has 512 of them here) and parked in
//     the lane's LDS slots at the top of chunk c + 1: the loop used to wait for a full HBM round trip BEHIND all of its own
//     outstanding stores once per chunk (loads and stores return in issue order);
//   * one gate-table read per step instead of two: the row the observation of step k is built with is the row step k + 1 starts
//     from (the target only changes inside a step);
//   * the observation tile of step k is written to LDS at the end of step k and streamed out in the middle of step k + 1 (reads
//     issued at the top of the step, stores behind the rotation matrix): the LDS round trip is no longer on the chain;
//   * per-step output addresses are scalar bases advanced with scalar adds + a constant per-lane 32-bit offset;
//   * the reset stash is filled in the prologue, in the shadow of the state loads (it only needs the episode counter), so that the
//     first terminating lane of a launch does not stall the wave for six Philox blocks.
// ---------------------------------------------------------------------------------------------------
template <int V, int GA>
__device__ __forceinline__ void obs_tile_store_rows(float* __restrict__ tile, int lane, const float* o) {
    constexpr int L = obs_len<V, GA>();
    float* row = tile + lane * L;
    if constexpr (L % 4 == 0) {
        float4* r4 = reinterpret_cast<float4*>(row);
#pragma unroll
        for (int k = 0; k < L / 4; ++k) r4[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) row[k] = o[k];
    }
}

// A chunk of actions is held in registers ACROSS iterations of the chunk loop: eight named float4s, not an array (an array that is
// live around the loop's back edge stayed in scratch: it is only indexable by constants after the inner loops are unrolled).
#define QR_BURST8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int V, int GA>   // lean form: steps of actions per burst
constexpr int lean_act_chunk() { return obs_len<V, GA>() > 32 ? 2 : 4; }
template <int V, int GA, bool kMlp>   // floats of (dynamic) LDS of the lean form: tables, observation tiles, action slots, layer-1 operands, reset pool
constexpr int lean_lds_floats() {
    return kResetTableFloats + kMaxGates * kGateStride + kBlock * obs_len<V, GA>() + 4 * kBlock * lean_act_chunk<V, GA>() +
           (kMlp ? 4 * kMlpQuads * 64 : 0) + 4 * 16 + 4 * 64 * reset_value_count<V>();
}

// kLean = the same loop for launches with MORE than one workgroup per CU, where 256 registers (two waves per SIMD) is the budget
// and LDS takes over what the registers hold in the other form (73-81 KB per workgroup, dynamic):
//   * no per-lane reset stash: a POOL of reset draws per wave in LDS, filled eight envs per Philox pass, ahead of need
//     (reset_pooled(): a pass every ~8 steps instead of the 0.69 passes per step of the batched reset_done_lanes());
//   * layer-1 weight operands re-read from LDS every step;
//   * the observation block of the previous step is read from LDS next to its stores, BEHIND the dynamics: read at the top of the
//     step (as the other form does, to take the LDS latency off a lone wave's chain) its 24 registers were live through the residual
//     MLPs and the allocator spilled two address pairs -- and the reload of a spilled value is a vector-memory wait (vmcnt(0)) that
//     drains the wave's whole queue of outstanding stores once per step;
//   * actions: the next 4-step chunk is requested into registers a chunk ahead, like the one-wave form.  (Round 4 fed the MLP form
//     through a RING of LDS slots filled by LDS-DMA with a COUNTED s_waitcnt vmcnt(N) on the consumer side: 1-3 % faster, and unsound --
//     loads and stores share vmcnt on gfx9-family hardware and do not retire in issue order against each other, so "N younger
//     operations" guarantees nothing; the no-MLP form was caught reading slots the load had not reached.  The compiler's own waits
//     for the register prefetch are conservative for exactly that reason (vmcnt(0) at the top of a chunk).  The ring is gone:
//     round 5 A/B on one box, 1 Mi envs 37.6 -> 37.1 G env-steps/s, profiles/r05_unguarded_ab.txt.)
// 1 Mi envs: 36.9 -> 39.6 G env-steps/s A/B'd on one box (profiles/r04_lean_ab.txt), every build checked against K x step_kernel under
// full-chip load (tools/lean_stress.py, profiles/r04_lean_stress.txt, tests/test_gpu_round4.py::test_lean_forms_agree...).
// kFull = every workgroup of the launch is full (n is a multiple of the workgroup size: the common case, and every benchmark
// size): `active` and `full_wave` are compile-time true and the EXEC-mask sequences around the ragged tail's stores leave the loop
// (46 of the loop's ~715 instructions; a lone wave pays ~5 cycles for every instruction it issues, scalar or vector).  Both copies
// live in ONE kernel behind a launch-uniform branch (rollout_fast_body): same symbols, same registers, same arithmetic.
template <int V, int GA, bool kMlp, bool kLean, bool kFull>
__device__ __forceinline__ void rollout_fast_body_impl(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                                                       float* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                                                       uint8_t* __restrict__ trunc_out, float* __restrict__ lds) {
    constexpr int kActChunk = kLean ? lean_act_chunk<V, GA>() : act_chunk<V, GA>();
    constexpr int L = obs_len<V, GA>();
    constexpr int S = Env<V>::S;
    constexpr int kVec = 16 * L;                 // float4 elements of a wave's [64][L] observation block
    constexpr int kFlush = (kVec + 63) / 64;     // store instructions per block
    constexpr bool kALds = kLean && kMlp;
    constexpr int kOffA = kResetTableFloats + kMaxGates * kGateStride + kBlock * L + 4 * kBlock * kActChunk;
    constexpr int kOffWho = kOffA + (kALds ? 4 * kMlpQuads * 64 : 0);   // lean: [4 waves][16] dwords, then the reset pool [4][64][NB] float4
    constexpr int kOffPool = kOffWho + 4 * 16;
    static_assert(!kLean || kOffPool + 4 * 64 * reset_value_count<V>() == lean_lds_floats<V, GA, kMlp>(), "lean LDS layout");
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = kFull || i < P.n;  // ragged tail lanes stay active (wave-wide MLP ops) and shadow env 0
    const int ii = active ? i : 0;
    const size_t n = (size_t)P.n;
    QR_CLOCK_STAMP(P, 0);
    QR_CLOCK_HWID(P);
    // ---- prologue: every load of the launch is requested before anything waits, in the order the data is needed (loads return in
    // issue order): tables (L2 hits; they go through LDS and a barrier) -> episode counters (the reset stash needs nothing else) ->
    // weight registers (L2) -> first action chunk and the state (HBM).  The stash -- six Philox blocks, ~2.5 k cycles of integer
    // work -- is then filled while the HBM loads are still in flight.
    float* rtab = lds;                        // [reset table | gate rows | obs tiles | action slots]
    float* gates = lds + kResetTableFloats;
    const int tab_vec = (kResetTableFloats + P.num_gates * kGateStride) / 4;   // <= 120 float4: one load per thread
    const float4* tsrc = reinterpret_cast<const float4*>(P.tables + kOffResetImage);
    const float4 tv = tsrc[(int)threadIdx.x < tab_vec ? threadIdx.x : 0];
    // (the non-lean form stages the MLP table through the action-slot area, which nothing uses before the first chunk: two 16-byte
    // loads per thread + LDS reads instead of 22 global loads per lane -- the per-step kernel measured 0.84 us for that difference)
    constexpr bool kMlpViaLds = kMlp && !kLean;
    constexpr int kMlpVec = kMlpTableFloats / 4;   // 356 float4
    static_assert(!kMlpViaLds || 4 * kBlock * kActChunk >= kMlpTableFloats, "the action-slot area holds the MLP table");
    const float4* msrc = reinterpret_cast<const float4*>(P.tables);
    float4 mv0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), mv1 = mv0;
    if constexpr (kMlpViaLds) {
        mv0 = msrc[threadIdx.x];
        mv1 = msrc[(int)threadIdx.x + kBlock < kMlpVec ? threadIdx.x + kBlock : 0];
    }
    const int2 ts0 = P.ts[ii];
    MlpRegs mlp;
    if (kMlp && !kMlpViaLds) mlp_load_regs(P.tables, lane, mlp, !kALds);
    float4 b0, b1, b2, b3, b4, b5, b6, b7;
    b0 = b1 = b2 = b3 = b4 = b5 = b6 = b7 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#define QR_X(J) if constexpr (J < kActChunk) b##J = actions[(size_t)(J < K ? J : K - 1) * n + ii];
    QR_BURST8(QR_X)   // first chunk of actions
#undef QR_X
    Env<V> e;
    load_env<V>(P, ii, e);
    if ((int)threadIdx.x < tab_vec) reinterpret_cast<float4*>(lds)[threadIdx.x] = tv;
    if constexpr (kALds) {
        mlp_stage_a(P.tables, reinterpret_cast<u32x4*>(lds + kOffA));
        mlp.a_lds = reinterpret_cast<const u32x4*>(lds + kOffA);
    }
    if constexpr (kMlpViaLds) {
        float4* mdst = reinterpret_cast<float4*>(lds + kResetTableFloats + kMaxGates * kGateStride + kBlock * L);   // = the action slots
        mdst[threadIdx.x] = mv0;
        if ((int)threadIdx.x + kBlock < kMlpVec) mdst[threadIdx.x + kBlock] = mv1;
    }
    __syncthreads();
    if constexpr (kMlpViaLds) {
        mlp_load_regs(lds + kResetTableFloats + kMaxGates * kGateStride + kBlock * L, lane, mlp);
        __syncthreads();   // every wave has its weight registers: the area is free for the action slots
    }
    // lane-private action slots (consecutive lanes = consecutive 16 B, conflict-free); each lane only reads back what it wrote itself
    float4* const act_slot = reinterpret_cast<float4*>(lds + kResetTableFloats + kMaxGates * kGateStride + kBlock * L) + threadIdx.x;
    const uint32_t gid_lo = P.gid_lo + (uint32_t)ii;
    const uint32_t gid_hi = P.gid_hi + (gid_lo < P.gid_lo ? 1u : 0u);
    float stash[kLean ? 1 : reset_value_count<V>()];
    float stash_od[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // E2E: the stashed episode's disturbance observation columns (Env::od)
    bool stash_ok = false;
    if constexpr (!kLean) {
        reset_values<V>(P, rtab, (uint32_t)ts0.x >> 8, gid_lo, gid_hi, stash);   // = what reset_from_stash() would draw on first use
        if constexpr (V == kE2E) disturbance_obs_values(P, stash + 16, stash_od);
        stash_ok = true;
    }
    if constexpr (V == kE2E) disturbance_obs_values(P, e.d, e.od);   // constant within an episode: refreshed by the resets below
    uint32_t* who = nullptr;
    float4* pool = nullptr;
    if constexpr (kLean) {
        who = reinterpret_cast<uint32_t*>(lds + kOffWho) + (threadIdx.x >> 6) * 16;
        pool = reinterpret_cast<float4*>(lds + kOffPool) + (threadIdx.x >> 6) * 16 * reset_value_count<V>();
    }
    bool pool_ok = false;                     // lean: this lane's pool row holds the draws of its current episode
    const int wave_first = i - lane;
    const bool full_wave = kFull || wave_first + 64 <= P.n;
    float* tile = gates + kMaxGates * kGateStride + (threadIdx.x >> 6) * 64 * L;
    // per-step output rows: scalar bases (advanced by scalar adds) + constant per-lane offsets
    const float4* tile4 = reinterpret_cast<const float4*>(tile);
    float* obs_step = obs_out;                 // row k of [K][n][L]
    float* rew_step = rew_out;
    uint8_t* done_step = done_out;
    uint8_t* trunc_step = trunc_out;
    GateRow gate = read_gate_row(gates, e.target);
    float4 rel[GA > 0 ? GA : 1];
    read_gates_ahead<GA>(P, gates, e.target, rel);
    bool any_reset = false;
    bool pending = false;                      // a tile written by the previous step waits to be streamed out (full waves)
    QR_CLOCK_STAMP(P, 1);
    for (int k0 = 0; k0 < K; k0 += kActChunk) {
        const int c = (K - k0 < kActChunk) ? K - k0 : kActChunk;
#define QR_X(J) if constexpr (J < kActChunk) act_slot[J * kBlock] = b##J;
        QR_BURST8(QR_X)
#undef QR_X
        if (k0 + kActChunk < K) {   // request the next chunk now; it lands while this chunk is simulated
#define QR_X(J) if constexpr (J < kActChunk) b##J = actions[(size_t)((k0 + kActChunk + J < K) ? k0 + kActChunk + J : K - 1) * n + ii];
            QR_BURST8(QR_X)
#undef QR_X
        }
        for (int j = 0; j < c; ++j) {
#ifdef QR_PHASE_TIMING
            P.tick_on = (k0 + j == K / 2);
#endif
            QR_TICK(P, 2);
            const float4 act = act_slot[j * kBlock];
            // stream the previous step's observation block out: LDS reads here, global stores after the rotation matrix
            // (lean: the reads happen next to the stores, behind the dynamics -- 24 registers that would otherwise be live through the
            // residual MLPs' peak, where this form has none to spare: they spilled, and every reload of a spilled value is a vector-memory
            // wait that drains the wave's whole store queue; the co-resident wave covers the LDS latency)
            float4 blk[kFlush];
            auto read_block = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int t = 0; t < kFlush; ++t) {
                    const int el = t * 64 + lane;
                    blk[t] = tile4[((t + 1) * 64 <= kVec || el < kVec) ? el : 0];
                }
            };
            if (!kLean && pending) read_block();
            const float u[4] = {act.x, act.y, act.z, act.w};
            float nw[S];
            int new_target;
            bool done, trunc;
            const float reward = step_dynamics<V, kALds ? 1 : 0>(P, gate, mlp, kMlp, lane, e, u, nw, new_target, done, trunc);
            QR_TICK(P, 5);
            if (pending) {
                if constexpr (kLean) read_block();
                float4* g4 = reinterpret_cast<float4*>(obs_step - n * L + (size_t)wave_first * L);
#pragma unroll
                for (int t = 0; t < kFlush; ++t) {
                    const int el = t * 64 + lane;
                    if ((t + 1) * 64 <= kVec || el < kVec) stream_store(g4 + el, blk[t]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            e.target = new_target;
            e.steps = e.steps + 1;
#pragma unroll
            for (int q = 0; q < S; ++q) e.s[q] = nw[q];
            any_reset |= done;
            if constexpr (kLean) {
                reset_pooled<V>(P, rtab, who, pool, lane, done && active, e, gid_lo, gid_hi, pool_ok);
                if constexpr (V == kE2E) {
                    if (done && active) disturbance_obs_values(P, e.d, e.od);
                }
            } else {
                reset_from_stash<V>(P, rtab, done && active, e, gid_lo, gid_hi, stash, stash_ok, (V == kE2E) ? stash_od : nullptr);
            }
            if (active) {
                stream_store(rew_step + i, reward);
                stream_store(done_step + i, (uint8_t)(done ? 1 : 0));
                if (trunc_step) stream_store(trunc_step + i, (uint8_t)(trunc ? 1 : 0));
            }
            QR_TICK(P, 6);
            // the row of the (possibly new) target: this step's observation and the next step's gate
            gate = read_gate_row(gates, e.target);
            read_gates_ahead<GA>(P, gates, e.target, rel);
            float o[L];
            observe_with<V, GA, true>(P, gate, rel, e, o);
            if (full_wave) {
                obs_tile_store_rows<V, GA>(tile, lane, o);   // streamed out in the middle of the NEXT step (worth ~600 cycles per step, r04)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                pending = true;
            } else if (active) {
                store_obs<V, GA>(obs_step, i, o);
            }
            QR_TICK(P, 7);
            obs_step += n * L;
            rew_step += n;
            done_step += n;
            if (trunc_step) trunc_step += n;
        }
    }
    QR_CLOCK_STAMP(P, 2);
    if (pending) obs_tile_flush<V, GA>(tile, obs_step - n * L, (size_t)wave_first, lane);
    if (!active) return;
    define_exit_values<V>(e);
    P.ts[i] = pack_ts<V>(e);
    store_world<V>(P, i, e);
    if (any_reset) store_dist<V>(P, i, e);
    QR_CLOCK_STAMP(P, 3);
}
template <int V, int GA, bool kMlp, bool kLean>
__device__ __forceinline__ void rollout_fast_body(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                                                  float* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                                                  uint8_t* __restrict__ trunc_out) {
    constexpr int kActChunk = kLean ? lean_act_chunk<V, GA>() : act_chunk<V, GA>();
    constexpr int kOffWho = kResetTableFloats + kMaxGates * kGateStride + kBlock * obs_len<V, GA>() + 4 * kBlock * kActChunk +
                            ((kLean && kMlp) ? 4 * kMlpQuads * 64 : 0);
    float* lds;
    if constexpr (kLean) {   // 73-81 KB per workgroup, two workgroups per CU: dynamic LDS (launch_rollout_lean sets the limit)
        extern __shared__ __attribute__((aligned(16))) float lds_lean[];
        lds = lds_lean;
    } else {
        __shared__ __attribute__((aligned(16))) float lds_fast[kOffWho];
        static_assert(sizeof(float) * kOffWho <= 65536, "static LDS");
        lds = lds_fast;
    }
    if (QR_FULL_OK && P.n % kBlock == 0) rollout_fast_body_impl<V, GA, kMlp, kLean, true>(P, K, actions, obs_out, rew_out, done_out, trunc_out, lds);
    else rollout_fast_body_impl<V, GA, kMlp, kLean, false>(P, K, actions, obs_out, rew_out, done_out, trunc_out, lds);
}
template <int V, int GA>
__global__ void __launch_bounds__(kBlock)
rollout_fast_kernel(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                    float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    rollout_fast_body<V, GA, false, false>(P, K, actions, obs_out, rew_out, done_out, trunc_out);
}
template <int V, int GA>   // E2E with the residual MLPs
__global__ void __launch_bounds__(kBlock)
rollout_fast_mlp_kernel(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                        float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    static_assert(V == kE2E, "residual MLPs belong to the E2E model");
    rollout_fast_body<V, GA, true, false>(P, K, actions, obs_out, rew_out, done_out, trunc_out);
}
template <int V, int GA>   // E2E with the residual MLPs, more than one workgroup per CU
__global__ void __launch_bounds__(kBlock, 2)
rollout_lean_mlp_kernel(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                        float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    static_assert(V == kE2E, "residual MLPs belong to the E2E model");
    rollout_fast_body<V, GA, true, true>(P, K, actions, obs_out, rew_out, done_out, trunc_out);
}

template <int V, int GA>   // INDI / E2E without the residual MLPs, more than one workgroup per CU
__global__ void __launch_bounds__(kBlock, 2)
rollout_lean_kernel(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                    float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    rollout_fast_body<V, GA, false, true>(P, K, actions, obs_out, rew_out, done_out, trunc_out);
}

template <int V, int GA>
__global__ void __launch_bounds__(kBlock)
rollout_kernel(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
               float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    rollout_body<V, GA, false>(P, K, actions, obs_out, rew_out, done_out, trunc_out);
}
template <int V, int GA>
__global__ void __launch_bounds__(kBlock)
rollout_stash_kernel(Params P, int K, const float4* __restrict__ actions, float* __restrict__ obs_out,
                     float* __restrict__ rew_out, uint8_t* __restrict__ done_out, uint8_t* __restrict__ trunc_out) {
    rollout_body<V, GA, true>(P, K, actions, obs_out, rew_out, done_out, trunc_out);
}

// ---------------------------------------------------------------------------------------------------
// Closed-loop rollout (qr_rollout_policy): policy network + Gaussian action sampling + env step, K times in one
// kernel.  Per step: obs (registers) -> MFMA policy -> mean; action = mean + std * N(0,1) (Philox + Box-Muller keyed
// by (noise seed, global env id, global step)); the buffer gets (obs_t, action_t, log-prob_t), the env gets the action
// clipped to the Box [-1, 1] (what SB3 does, R:785); reward_t / done_t follow; the post-step observation feeds the
// next step.  This is PPO's collect phase (R:820 -> SB3 collect_rollouts) without leaving the chip.
// ---------------------------------------------------------------------------------------------------
// kF32: the reference-precision forward (policy_forward_f32class: every operand as two f16 pieces, the low-piece image read from global
// memory) instead of the hand-scheduled f16-operand one; noise and the observation store then simply run in front of it.
template <int V, int GA, bool kF32 = false>
__global__ void __launch_bounds__(kBlock, 1)
rollout_policy_kernel(Params P, PolicyArgs A, int K, float* __restrict__ obs_out, float4* __restrict__ act_out,
                      float* __restrict__ logp_out, float* __restrict__ rew_out, uint8_t* __restrict__ done_out,
                      uint8_t* __restrict__ trunc_out, float* __restrict__ last_obs_out) {
    constexpr int L = obs_len<V, GA>();
    using D = PolicyDims<L>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* W = reinterpret_cast<half8*>(smem);                                   // policy weights (f16)
    float* rtab = reinterpret_cast<float*>(smem + (size_t)D::kTotalHalf8 * 16);  // reset table | gate rows | obs tiles
    float* gates = rtab + kResetTableFloats;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = i < P.n;
    const int ii = active ? i : 0;
    Env<V> e;
    load_env<V>(P, ii, e);
    MlpRegs mlp;
    const bool use_mlp = (V == kE2E) && (P.flags & kFlagResidual);
    if (use_mlp) mlp_load_regs(P.tables, lane, mlp);
    {
        const float4* s4 = reinterpret_cast<const float4*>(A.weights);
        float4* d4 = reinterpret_cast<float4*>(W);
        for (int j = threadIdx.x; j < D::kTotalHalf8; j += kBlock) d4[j] = s4[j];
    }
    stage_tables(P, rtab, kOffResetImage, kResetTableFloats + P.num_gates * kGateStride);
    __syncthreads();
    const uint32_t gid_lo = P.gid_lo + (uint32_t)ii;
    const uint32_t gid_hi = P.gid_hi + (gid_lo < P.gid_lo ? 1u : 0u);
    const size_t n = (size_t)P.n;
    const int wave_first = i - lane;
    const bool full_wave = wave_first + 64 <= P.n;
    float* tile = gates + kMaxGates * kGateStride + (threadIdx.x >> 6) * 64 * L;
    bool any_reset = false;
    float stash[reset_value_count<V>()];   // the lane's own next reset draws (reset_from_stash; this kernel always has the registers)
    bool stash_ok = false;
    float o[L];
    observe<V, GA>(P, gates, e, o);
    for (int k = 0; k < K; ++k) {
#ifdef QR_PHASE_TIMING
        P.tick_on = (k == K / 2);
#endif
        QR_TICK(P, 8);
        // Action noise eps ~ N(0, 1)^4 (Philox4x32-10 keyed by (noise seed, global env id, global step) + Box-Muller) does not
        // depend on the policy output: its ~300 VALU instructions are cut into slices that policy_forward() places between
        // the MFMAs of the second hidden layer, where the wave otherwise only waits for the matrix core.
        float mean[4];
        float eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t pc[4];
        float bm_u1a, bm_u2a, bm_u1b, bm_u2b, bm_ra, bm_rb, bm_sa, bm_ca, bm_sb, bm_cb;
        // (drawn in deterministic mode too and then multiplied out: a branch would split the pinned MFMA schedule)
        // Each slice first passes the values it reads through an empty volatile asm: that emits nothing, but it is ordered
        // with the sched_barrier pins of policy_layer (both are side-effecting), which is what keeps the slice in ITS slot --
        // plain arithmetic would all be hoisted in front of the layer.
        auto pin_u = [](uint32_t& x) { asm volatile("" : "+v"(x)); };
        auto pin_f = [](float& x) { asm volatile("" : "+v"(x)); };
        auto noise_slice = [&](int slot) {
            if (slot >= 1 && slot <= 11) { pin_u(pc[0]); pin_u(pc[1]); pin_u(pc[2]); pin_u(pc[3]); }
            if (slot == 12) pin_f(bm_u1a);
            if (slot == 13) pin_f(bm_u1b);
            if (slot == 14) pin_f(bm_u2a);
            if (slot == 15) pin_f(bm_u2b);
            if (slot == 16) { pin_f(bm_ra); pin_f(bm_rb); pin_f(bm_sa); pin_f(bm_sb); }
            if (slot == 0) {
                const uint32_t s_lo = A.step_lo + (uint32_t)k;
                pc[0] = gid_lo; pc[1] = gid_hi; pc[2] = s_lo; pc[3] = A.step_hi + (s_lo < A.step_lo ? 1u : 0u);
            } else if (slot <= 10) {
                philox4x32_round(pc, A.seed_lo, A.seed_hi, slot - 1);
            } else if (slot == 11) {  // Box-Muller: two pairs of normals from four uniforms (u1 in (0,1], u2 in [0,1))
                bm_u1a = (float)((pc[0] >> 8) + 1u) * 5.9604644775390625e-8f; bm_u2a = u01(pc[1]);
                bm_u1b = (float)((pc[2] >> 8) + 1u) * 5.9604644775390625e-8f; bm_u2b = u01(pc[3]);
            } else if (slot == 12) {
                bm_ra = fast_sqrt(-2.0f * __logf(bm_u1a));
            } else if (slot == 13) {
                bm_rb = fast_sqrt(-2.0f * __logf(bm_u1b));
            } else if (slot == 14) {
                qr_sincos(6.283185307179586f * bm_u2a, bm_sa, bm_ca);
            } else if (slot == 15) {
                qr_sincos(6.283185307179586f * bm_u2b, bm_sb, bm_cb);
            } else if (slot == 16) {
                eps[0] = bm_ra * bm_ca; eps[1] = bm_ra * bm_sa; eps[2] = bm_rb * bm_cb; eps[3] = bm_rb * bm_sb;
            }
        };
        // The observation row of this step (the policy's input) is stored under the third layer's MFMAs: LDS transpose in
        // slot 0, coalesced block store in slot 2 (full waves; the ragged tail wave stores its rows afterwards).
        auto obs_slice = [&](int slot) {
            if (slot == 0 && full_wave) obs_tile_write<V, GA>(tile, lane, o);
            if (slot == 2 && full_wave) obs_tile_flush<V, GA>(tile, obs_out + (size_t)k * n * L, (size_t)wave_first, lane);
        };
        if constexpr (kF32) {
#pragma unroll
            for (int slot = 0; slot <= 16; ++slot) noise_slice(slot);
            obs_slice(0);
            obs_slice(2);
            policy_forward_f32class<L>(W, A.weights_lo, lane, o, mean);
        } else {
            policy_forward<L>(W, lane, o, mean, noise_slice, obs_slice);
        }
        QR_TICK(P, 9);
        float a[4] = {mean[0], mean[1], mean[2], mean[3]};
        float logp = A.logp_const;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float e = A.deterministic ? 0.0f : eps[c];   // fmaf(std, 0, mean) = mean, fmaf(-0, 0, logp) = logp
            a[c] = fmaf(A.std[c], e, mean[c]);
            logp = fmaf(-0.5f * e, e, logp);
        }
        QR_TICK(P, 10);
        // rollout buffer row t: the observation the action was computed from (stored above), the unclipped action, its log-prob
        if (!full_wave && active) store_obs<V, GA>(obs_out + (size_t)k * n * L, i, o);
        if (active) {
            stream_store(act_out + (size_t)k * n + i, make_float4(a[0], a[1], a[2], a[3]));
            stream_store(logp_out + (size_t)k * n + i, logp);
        }
        QR_TICK(P, 11);
        const float u[4] = {fminf(fmaxf(a[0], -1.0f), 1.0f), fminf(fmaxf(a[1], -1.0f), 1.0f),
                            fminf(fmaxf(a[2], -1.0f), 1.0f), fminf(fmaxf(a[3], -1.0f), 1.0f)};
        bool done, trunc, did_reset;
        const float reward = step_env<V>(P, gates, rtab, tile, mlp, lane, active, e, u, gid_lo, gid_hi, done, trunc,
                                         did_reset, [&](bool fin) {
                                             store_terminal_obs<V, GA>(P, gates, e, (size_t)k * n, i, fin && active);
                                         }, [&](bool need) { reset_from_stash<V>(P, rtab, need, e, gid_lo, gid_hi, stash, stash_ok); });
        any_reset |= did_reset;
        if (active) {
            stream_store(rew_out + (size_t)k * n + i, reward);
            stream_store(done_out + (size_t)k * n + i, (uint8_t)(done ? 1 : 0));
            if (trunc_out) stream_store(trunc_out + (size_t)k * n + i, (uint8_t)(trunc ? 1 : 0));
        }
        QR_TICK(P, 12);
        observe<V, GA>(P, gates, e, o);
        QR_TICK(P, 13);
    }
    if (last_obs_out) {
        if (full_wave) store_obs_coalesced<V, GA>(tile, last_obs_out, (size_t)wave_first, lane, o);
        else if (active) store_obs<V, GA>(last_obs_out, i, o);
    }
    if (!active) return;
    define_exit_values<V>(e);
    P.ts[i] = pack_ts<V>(e);
    store_world<V>(P, i, e);
    if (any_reset) store_dist<V>(P, i, e);
}

// reset_(mask) + update_states for ALL envs (R:452-496)
template <int V, int GA>
__global__ void __launch_bounds__(kBlock)
reset_kernel(Params P, const uint8_t* __restrict__ mask, float* __restrict__ obs_out) {
    __shared__ __attribute__((aligned(16))) float lds_all[kResetTableFloats + kMaxGates * kGateStride];
    const float* lds = lds_all + kResetTableFloats;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    stage_tables(P, lds_all, kOffResetImage, kResetTableFloats + P.num_gates * kGateStride);
    __syncthreads();
    if (i >= P.n) return;
    Env<V> e;
    load_env<V>(P, i, e);
    if (!mask || mask[i]) {
        const uint32_t gid_lo = P.gid_lo + (uint32_t)i;
        const uint32_t gid_hi = P.gid_hi + (gid_lo < P.gid_lo ? 1u : 0u);
        reset_env<V>(P, lds_all, e, gid_lo, gid_hi);
        store_world<V>(P, i, e);
        store_dist<V>(P, i, e);
        P.ts[i] = pack_ts<V>(e);
    }
    if (obs_out) {
        float o[obs_len<V, GA>()];
        observe<V, GA>(P, lds, e, o);
        store_obs<V, GA>(obs_out, i, o);
    }
}

// update_states(): observation from the current state
template <int V, int GA>
__global__ void __launch_bounds__(kBlock)
observe_kernel(Params P, float* __restrict__ obs_out) {
    __shared__ __attribute__((aligned(16))) float lds[kMaxGates * kGateStride];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    stage_tables(P, lds, kOffGatesImage, P.num_gates * kGateStride);
    __syncthreads();
    if (i >= P.n) return;
    Env<V> e;
    load_env<V>(P, i, e);
    float o[obs_len<V, GA>()];
    observe<V, GA>(P, lds, e, o);
    store_obs<V, GA>(obs_out, i, o);
}

// planar <-> row-major state export / import (attribute access in the adapter; parity injection)
template <int V>
__global__ void __launch_bounds__(kBlock)
get_state_kernel(Params P, float* __restrict__ world, float* __restrict__ dist, int32_t* __restrict__ target,
                 int32_t* __restrict__ steps, uint32_t* __restrict__ episode) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env<V> e;
    load_env<V>(P, i, e);
    constexpr int S = Env<V>::S;
    if (world) {
#pragma unroll
        for (int k = 0; k < S; ++k) world[(size_t)i * S + k] = e.s[k];
    }
    if (V == kE2E && dist) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dist[(size_t)i * 6 + k] = e.d[k];
    }
    if (target) target[i] = e.target;
    if (steps) steps[i] = e.steps;
    if (episode) episode[i] = e.episode;
}

template <int V>
__global__ void __launch_bounds__(kBlock)
set_state_kernel(Params P, const float* __restrict__ world, const float* __restrict__ dist,
                 const int32_t* __restrict__ target, const int32_t* __restrict__ steps,
                 const uint32_t* __restrict__ episode) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    Env<V> e;
    load_env<V>(P, i, e);
    constexpr int S = Env<V>::S;
    if (world) {
#pragma unroll
        for (int k = 0; k < S; ++k) e.s[k] = world[(size_t)i * S + k];
        store_world<V>(P, i, e);
    }
    if (V == kE2E && dist) {
#pragma unroll
        for (int k = 0; k < 6; ++k) e.d[k] = dist[(size_t)i * 6 + k];
        store_dist<V>(P, i, e);
    }
    if (target) {  // the reference indexes with target % num_gates (R:367-368); keep the invariant 0 <= t < G
        int t = target[i] % P.num_gates;
        if (t < 0) t += P.num_gates;
        e.target = t;
    }
    if (steps) e.steps = steps[i];
    if (episode) e.episode = episode[i] & 0xFFFFFFu;
    P.ts[i] = pack_ts<V>(e);
}

#ifndef QR_TU_MLP_ROLLOUT   // (everything from here on belongs to the main translation unit, except launch_rollout_mlp at the end)
// qr_probe_residual: body velocity (R:103) and the residual thrust / moment MLP outputs (R:254-262) of the CURRENT state
// of every env, row [vbx vby vbz thrust Mx My Mz] -- the same device functions the step kernels inline, exposed so that
// parity tests can pin them directly against the reference's fixture rows instead of through finite differences.
__global__ void __launch_bounds__(kBlock) residual_probe_kernel(Params P, float* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = i < P.n;   // MFMA / permlane are wave-wide: tail lanes shadow env 0
    Env<kE2E> e;
    load_env<kE2E>(P, active ? i : 0, e);
    MlpRegs mlp;
    mlp_load_regs(P.tables, lane, mlp);
    const Rot R = make_rot(e.s[6], e.s[7], e.s[8]);
    float vb[3];
    vb[0] = fmaf(e.s[3], R.r00, fmaf(e.s[4], R.r10, e.s[5] * R.r20));
    vb[1] = fmaf(e.s[3], R.r01, fmaf(e.s[4], R.r11, e.s[5] * R.r21));
    vb[2] = fmaf(e.s[3], R.r02, fmaf(e.s[4], R.r12, e.s[5] * R.r22));
    const float x[10] = {e.s[12], e.s[13], e.s[14], e.s[15], vb[0], vb[1], vb[2], e.s[9], e.s[10], e.s[11]};
    float thrust, moment[3];
    residual_mlp(mlp, lane, x, thrust, moment);
    if (!active) return;
    float* o = out + (size_t)i * 7;
    o[0] = vb[0]; o[1] = vb[1]; o[2] = vb[2]; o[3] = thrust; o[4] = moment[0]; o[5] = moment[1]; o[6] = moment[2];
}

// qr_seed: restart every env's reset stream (episode counter = 0)
__global__ void __launch_bounds__(kBlock) clear_episode_kernel(Params P) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    int2 ts = P.ts[i];
    ts.x &= 0xFF;
    P.ts[i] = ts;
}

#endif  // !QR_TU_MLP_ROLLOUT

// ---------------------------------------------------------------------------------------------------
// host-callable launchers (used by quadrace_abi.hip)
// ---------------------------------------------------------------------------------------------------
static inline dim3 grid_for(int n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// The two fused E2E + residual-MLP kernels (rollout_fast_mlp_kernel, rollout_lean_mlp_kernel) are instantiated in a translation unit
// of their own, quadrace_kernels_mlp.hip = this file with QR_TU_MLP_ROLLOUT defined, compiled WITHOUT the SLP vectoriser
// (build.py PER_SOURCE_FLAGS): next to their matrix instructions, and above all at two waves per SIMD where a packed-f32
// instruction costs 1.3 x a scalar one (profiles/r04_valu_rate.txt), the vectoriser's packed operations and the ~80 register moves
// that feed them are a net loss there: 1 Mi envs 40.5 -> 42.3 G env-steps/s, 65 536 envs + 1.5 % (profiles/r05_slp_ab.txt).  The INDI
// and per-step kernels keep it (INDI at 65 536 envs loses 8 % without).  Same arithmetic either way: the vectoriser packs, it does not
// re-associate (-ffp-contract=off, explicit fmaf) -- the forms stay bit-identical (tests/test_gpu_round4.py).
hipError_t launch_rollout_mlp(bool lean, const Params& P, int K, const float4* a4, float* obs, float* rew, uint8_t* done,
                              uint8_t* trunc, hipStream_t st);

// compile-time (variant, gates_ahead) dispatch: keeps every observation index static (registers, no scratch)
#ifdef QR_GA_ONLY  // developer builds (ISA inspection, tools/phase_timing.py): instantiate one gates_ahead value only
#define QR_DISPATCH_GA(V, KERNEL, ...)                                                                  \
    if (P.gates_ahead != QR_GA_ONLY) return hipErrorInvalidValue;                                       \
    hipLaunchKernelGGL((KERNEL<V, QR_GA_ONLY>), grid_for(P.n), dim3(kBlock), 0, st, __VA_ARGS__);
#else
#define QR_DISPATCH_GA(V, KERNEL, ...)                                                                  \
    switch (P.gates_ahead) {                                                                            \
        case 0: hipLaunchKernelGGL((KERNEL<V, 0>), grid_for(P.n), dim3(kBlock), 0, st, __VA_ARGS__); break; \
        case 1: hipLaunchKernelGGL((KERNEL<V, 1>), grid_for(P.n), dim3(kBlock), 0, st, __VA_ARGS__); break; \
        case 2: hipLaunchKernelGGL((KERNEL<V, 2>), grid_for(P.n), dim3(kBlock), 0, st, __VA_ARGS__); break; \
        case 3: hipLaunchKernelGGL((KERNEL<V, 3>), grid_for(P.n), dim3(kBlock), 0, st, __VA_ARGS__); break; \
        case 4: hipLaunchKernelGGL((KERNEL<V, 4>), grid_for(P.n), dim3(kBlock), 0, st, __VA_ARGS__); break; \
        default: return hipErrorInvalidValue;                                                           \
    }
#endif

#ifndef QR_TU_MLP_ROLLOUT
hipError_t launch_step(int variant, const Params& P, const float* actions, float* obs, float* rew, uint8_t* done,
                       uint8_t* trunc, hipStream_t st) {
    const float4* a4 = reinterpret_cast<const float4*>(actions);
    if (variant == kE2E) { QR_DISPATCH_GA(kE2E, step_kernel, P, a4, obs, rew, done, trunc) }
    else { QR_DISPATCH_GA(kINDI, step_kernel, P, a4, obs, rew, done, trunc) }
    return hipGetLastError();
}

static int n_wgs(int n) { return (n + kBlock - 1) / kBlock; }
static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}
// Which kernel runs a K-step call -- ONE selection, used by the launcher and by qr_rollout_kernel_name() (bench.py prints the symbol
// rocprofv3 will show, and looks its counter evidence up under that name):
//   at most one workgroup per CU, default mode (no pause flags, no terminal-observation rows):
//       E2E + residual MLPs -> rollout_fast_mlp_kernel, E2E without -> rollout_fast_kernel, INDI -> rollout_stash_kernel (HBM-bound at
//       65 536 envs: the general stash kernel is as fast per step, 2 443 vs 2 503 cycles, and has the shorter prologue)
//   at most one workgroup per CU, any other mode -> rollout_stash_kernel
//   more workgroups, default mode: E2E + residual MLPs -> rollout_lean_mlp_kernel, INDI / E2E without -> rollout_lean_kernel up to four
//       workgroups per CU;
//   more workgroups, anything else -> rollout_kernel
// `form` = qr_set_rollout_form(), two flags: QR_ROLLOUT_MULTI_WAVE (bit 0: the forms built for more than one workgroup per CU at any
// env count), QR_ROLLOUT_GENERAL (bit 1: rollout_stash_kernel / rollout_kernel for every launch); 0 = QR_ROLLOUT_AUTO (the table above).  All forms are
// bit-identical (tests/test_gpu_round4.py); the setter exists for tests and A/B runs -- there is no environment variable.
enum RolloutKernel { kRkFastMlp, kRkFast, kRkStash, kRkLeanMlp, kRkLean, kRkPlain };
static RolloutKernel select_rollout(int variant, const Params& P, int form) {
    const bool plain_mode = !(P.flags & (kFlagPause | kFlagPauseIfCollision)) && P.term_obs == nullptr && !(form & 2);
    const bool mlp = variant == kE2E && (P.flags & kFlagResidual);
    if ((form & 4) || (n_wgs(P.n) <= device_cus() && !(form & 1))) {   // one wave per SIMD: the register file of a whole SIMD per wave (reset stash, operands in registers)
        if (plain_mode && mlp) return kRkFastMlp;
        if (plain_mode && variant == kE2E) return kRkFast;
        return kRkStash;
    }
    if (plain_mode && mlp) return kRkLeanMlp;
    // without the MLPs the model is memory-bound, and the lean form's LDS (52 KB: three workgroups per CU) is its occupancy limit where
    // the general form has four: INDI 61.8 vs 56.4 G env-steps/s at 262 144 envs (4 workgroups of work per CU), 56.4 vs 58.0 at 1 Mi
    if (plain_mode && (n_wgs(P.n) <= 4 * device_cus())) return kRkLean;
    return kRkPlain;
}

const char* rollout_kernel_name(int variant, const Params& P, int form) {
    switch (select_rollout(variant, P, form)) {
        case kRkFastMlp: return "rollout_fast_mlp_kernel";
        case kRkFast: return "rollout_fast_kernel";
        case kRkStash: return "rollout_stash_kernel";
        case kRkLeanMlp: return "rollout_lean_mlp_kernel";
        case kRkLean: return "rollout_lean_kernel";
        default: return "rollout_kernel";
    }
}

#endif  // !QR_TU_MLP_ROLLOUT

// the lean forms' LDS is dynamic (more than the 64 KB a static array may have): limit set once per device and instantiation
template <int V, int GA, bool kMlp>
static hipError_t launch_rollout_lean_vg(const Params& P, int K, const float4* a4, float* obs, float* rew, uint8_t* done,
                                         uint8_t* trunc, hipStream_t st) {
    constexpr size_t lds_need = sizeof(float) * lean_lds_floats<V, GA, kMlp>();
    static_assert(2 * lds_need <= 160 * 1024, "two workgroups per CU");
    constexpr size_t lds = lds_need;
    static unsigned long long configured = 0;   // per device ordinal
    if constexpr (kMlp) {
#ifdef QR_TU_MLP_ROLLOUT
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(rollout_lean_mlp_kernel<V, GA>), lds, configured)) return e;
        hipLaunchKernelGGL((rollout_lean_mlp_kernel<V, GA>), grid_for(P.n), dim3(kBlock), lds, st, P, K, a4, obs, rew, done, trunc);
#else
        return hipErrorInvalidValue;   // (instantiated in quadrace_kernels_mlp.hip only)
#endif
    } else {
#ifndef QR_TU_MLP_ROLLOUT
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(rollout_lean_kernel<V, GA>), lds, configured)) return e;
        hipLaunchKernelGGL((rollout_lean_kernel<V, GA>), grid_for(P.n), dim3(kBlock), lds, st, P, K, a4, obs, rew, done, trunc);
#endif
    }
    return hipGetLastError();
}
template <int V, bool kMlp>
static hipError_t launch_rollout_lean(const Params& P, int K, const float4* a4, float* obs, float* rew, uint8_t* done,
                                      uint8_t* trunc, hipStream_t st) {
#ifdef QR_GA_ONLY
    if (P.gates_ahead != QR_GA_ONLY) return hipErrorInvalidValue;
    return launch_rollout_lean_vg<V, QR_GA_ONLY, kMlp>(P, K, a4, obs, rew, done, trunc, st);
#else
    switch (P.gates_ahead) {
        case 0: return launch_rollout_lean_vg<V, 0, kMlp>(P, K, a4, obs, rew, done, trunc, st);
        case 1: return launch_rollout_lean_vg<V, 1, kMlp>(P, K, a4, obs, rew, done, trunc, st);
        case 2: return launch_rollout_lean_vg<V, 2, kMlp>(P, K, a4, obs, rew, done, trunc, st);
        case 3: return launch_rollout_lean_vg<V, 3, kMlp>(P, K, a4, obs, rew, done, trunc, st);
        case 4: return launch_rollout_lean_vg<V, 4, kMlp>(P, K, a4, obs, rew, done, trunc, st);
        default: return hipErrorInvalidValue;
    }
#endif
}

#ifdef QR_TU_MLP_ROLLOUT
hipError_t launch_rollout_mlp(bool lean, const Params& P, int K, const float4* a4, float* obs, float* rew, uint8_t* done,
                              uint8_t* trunc, hipStream_t st) {
    if (lean) return launch_rollout_lean<kE2E, true>(P, K, a4, obs, rew, done, trunc, st);
    QR_DISPATCH_GA(kE2E, rollout_fast_mlp_kernel, P, K, a4, obs, rew, done, trunc)
    return hipGetLastError();
}
#else
hipError_t launch_rollout(int variant, const Params& P, int form, int K, const float* actions, float* obs, float* rew,
                          uint8_t* done, uint8_t* trunc, hipStream_t st) {
    const float4* a4 = reinterpret_cast<const float4*>(actions);
    switch (select_rollout(variant, P, form)) {
        case kRkFastMlp: return launch_rollout_mlp(false, P, K, a4, obs, rew, done, trunc, st);
        case kRkFast: { QR_DISPATCH_GA(kE2E, rollout_fast_kernel, P, K, a4, obs, rew, done, trunc) } break;
        case kRkLeanMlp: return launch_rollout_mlp(true, P, K, a4, obs, rew, done, trunc, st);
        case kRkLean:
            if (variant == kE2E) return launch_rollout_lean<kE2E, false>(P, K, a4, obs, rew, done, trunc, st);
            return launch_rollout_lean<kINDI, false>(P, K, a4, obs, rew, done, trunc, st);
        case kRkStash:
            if (variant == kE2E) { QR_DISPATCH_GA(kE2E, rollout_stash_kernel, P, K, a4, obs, rew, done, trunc) }
            else { QR_DISPATCH_GA(kINDI, rollout_stash_kernel, P, K, a4, obs, rew, done, trunc) }
            break;
        default:
            if (variant == kE2E) { QR_DISPATCH_GA(kE2E, rollout_kernel, P, K, a4, obs, rew, done, trunc) }
            else { QR_DISPATCH_GA(kINDI, rollout_kernel, P, K, a4, obs, rew, done, trunc) }
    }
    return hipGetLastError();
}

template <int V, int GA>
hipError_t launch_rollout_policy_vg(const Params& P, const PolicyArgs& A, int K, float* obs, float* act, float* logp,
                                    float* rew, uint8_t* done, uint8_t* trunc, float* last_obs, hipStream_t st) {
    constexpr int L = obs_len<V, GA>();
    const size_t lds = (size_t)PolicyDims<L>::kTotalHalf8 * 16 +
                       sizeof(float) * (kResetTableFloats + kMaxGates * kGateStride + kBlock * L);
    static unsigned long long configured = 0, configured32 = 0;   // per device ordinal
    if (A.f32class) {
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(rollout_policy_kernel<V, GA, true>), lds, configured32)) return e;
        hipLaunchKernelGGL((rollout_policy_kernel<V, GA, true>), grid_for(P.n), dim3(kBlock), lds, st, P, A, K, obs,
                           reinterpret_cast<float4*>(act), logp, rew, done, trunc, last_obs);
        return hipGetLastError();
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(rollout_policy_kernel<V, GA>), lds, configured)) return e;
    hipLaunchKernelGGL((rollout_policy_kernel<V, GA>), grid_for(P.n), dim3(kBlock), lds, st, P, A, K, obs,
                       reinterpret_cast<float4*>(act), logp, rew, done, trunc, last_obs);
    return hipGetLastError();
}

hipError_t launch_rollout_policy(int variant, const Params& P, const PolicyArgs& A, int K, float* obs, float* act,
                                 float* logp, float* rew, uint8_t* done, uint8_t* trunc, float* last_obs,
                                 hipStream_t st) {
#define QR_RP(V, GA) return launch_rollout_policy_vg<V, GA>(P, A, K, obs, act, logp, rew, done, trunc, last_obs, st)
#ifdef QR_GA_ONLY
    if (P.gates_ahead != QR_GA_ONLY) return hipErrorInvalidValue;
    if (variant == kE2E) QR_RP(kE2E, QR_GA_ONLY);
    QR_RP(kINDI, QR_GA_ONLY);
#else
    if (variant == kE2E) {
        switch (P.gates_ahead) { case 0: QR_RP(kE2E, 0); case 1: QR_RP(kE2E, 1); case 2: QR_RP(kE2E, 2);
                                 case 3: QR_RP(kE2E, 3); case 4: QR_RP(kE2E, 4); }
    } else {
        switch (P.gates_ahead) { case 0: QR_RP(kINDI, 0); case 1: QR_RP(kINDI, 1); case 2: QR_RP(kINDI, 2);
                                 case 3: QR_RP(kINDI, 3); case 4: QR_RP(kINDI, 4); }
    }
#endif
#undef QR_RP
    return hipErrorInvalidValue;
}

hipError_t launch_reset(int variant, const Params& P, const uint8_t* mask, float* obs, hipStream_t st) {
    if (variant == kE2E) { QR_DISPATCH_GA(kE2E, reset_kernel, P, mask, obs) }
    else { QR_DISPATCH_GA(kINDI, reset_kernel, P, mask, obs) }
    return hipGetLastError();
}

hipError_t launch_observe(int variant, const Params& P, float* obs, hipStream_t st) {
    if (variant == kE2E) { QR_DISPATCH_GA(kE2E, observe_kernel, P, obs) }
    else { QR_DISPATCH_GA(kINDI, observe_kernel, P, obs) }
    return hipGetLastError();
}

hipError_t launch_residual_probe(const Params& P, float* out, hipStream_t st) {
    hipLaunchKernelGGL(residual_probe_kernel, grid_for(P.n), dim3(kBlock), 0, st, P, out);
    return hipGetLastError();
}

hipError_t launch_clear_episode(const Params& P, hipStream_t st) {
    hipLaunchKernelGGL(clear_episode_kernel, grid_for(P.n), dim3(kBlock), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_get_state(int variant, const Params& P, float* world, float* dist, int32_t* target, int32_t* steps,
                            uint32_t* episode, hipStream_t st) {
    if (variant == kE2E)
        hipLaunchKernelGGL(get_state_kernel<kE2E>, grid_for(P.n), dim3(kBlock), 0, st, P, world, dist, target, steps,
                           episode);
    else
        hipLaunchKernelGGL(get_state_kernel<kINDI>, grid_for(P.n), dim3(kBlock), 0, st, P, world, dist, target, steps,
                           episode);
    return hipGetLastError();
}

hipError_t launch_set_state(int variant, const Params& P, const float* world, const float* dist,
                            const int32_t* target, const int32_t* steps, const uint32_t* episode, hipStream_t st) {
    if (variant == kE2E)
        hipLaunchKernelGGL(set_state_kernel<kE2E>, grid_for(P.n), dim3(kBlock), 0, st, P, world, dist, target, steps,
                           episode);
    else
        hipLaunchKernelGGL(set_state_kernel<kINDI>, grid_for(P.n), dim3(kBlock), 0, st, P, world, dist, target, steps,
                           episode);
    return hipGetLastError();
}

#endif  // QR_TU_MLP_ROLLOUT

}  // namespace qr
