// quadrace_ppo.hip -- PPO minibatch update on the gfx950 matrix cores (BASELINE config 5).
//
// The reference trains with SB3's PPO (R:783-795): two separate ReLU MLPs  obs[L] -> 120 -> 120 -> 120 -> {4 | 1}
// (policy mean / value), a state-independent log-std, clipped surrogate + vf_coef * MSE value loss, per-minibatch
// advantage normalisation, global grad-norm clipping, Adam.  With the collect phase fused into one kernel
// (qr_rollout_policy) torch's minibatch update was > 95 % of training time: ~100 small launches around
// 16 k x 120 x 120 GEMMs.  Here one minibatch is two launches (grad, apply) plus one adv_stats launch per EPOCH:
//
//   adv_stats   sum / sum of squares of the advantages of EVERY minibatch of an epoch in one launch (qr_ppo_epoch_begin;
//               SB3 normalises per minibatch; the gradient kernel finishes the maths)
//   grad        ppo_grad_kernel: a workgroup = 8 waves = 128 samples of one net per pass.  Chain waves 0-3: one 32-sample tile each --
//               forward (f16 MFMA chain of quadrace_policy.hpp), per-sample loss gradients, backward through W^T read out of the SAME
//               LDS image (ds_read_b64_tr_b16); activations h_l and deltas d_l stay in registers in the "lane = sample" form and are
//               published per layer as natural packs to a 64 KB exchange area in LDS.  dW waves 4-7 read them back transposed
//               (lane = unit, k = sample) and form dW_l = d_l^T x h_(l-1): 2 x 2 weight tiles per wave, k = the workgroup's 128
//               samples, plain stores to partial[workgroup][slot] in accumulator order -- bf16 by default (the partials' trip
//               through the fabric is what bounds a 16 384-row update; QR_PPO_PARTIAL_F32 keeps f32) -- widened again and summed in
//               f32 in a fixed order by `apply`.  Biases ride along as the constant-1 unit of every layer.  No atomics anywhere:
//               log-std gradients and loss statistics leave as per-wave sums.
//   apply       ONE kernel: sums the partials and the per-wave sums into the gradient, accumulates its squared norm, crosses a
//               grid-wide barrier (247 co-resident workgroups), then clip scale, torch.optim.Adam arithmetic, and each thread
//               scatters its new parameter as f16 into the operand image of the next minibatch; SB3's target-KL early stop is
//               decided here, on the device, before the step is taken
//   (pack       the gather form of the same image layout: initial images / after external changes of theta;
//    reduce     gradient + minibatch statistics only, for the data-parallel path: qr_ppo_grad -> all-reduce -> qr_ppo_apply)
// The earlier forms of `grad` (rounds 1-2: a two-kernel split through an HBM scratch buffer; a 4-wave fused kernel with one dependent
// chain per wave) are gone since round 6; docs/history/DESIGN_rounds1-4.md describes them.
//
// Operand layouts are those of quadrace_policy.hpp (verified on MI355X with tools/ubench/mfma_layout.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "../../include/quadrace.h"
#include "quadrace_policy.hpp"

namespace qr {

constexpr int kH = kPolHidden;  // 120

struct NetOff { int w1, b1, w2, b2, w3, b3, w4, b4, total; };
__host__ __device__ inline NetOff net_off(int L, int O) {
    NetOff o;
    o.w1 = 0;
    o.b1 = o.w1 + kH * L;
    o.w2 = o.b1 + kH;
    o.b2 = o.w2 + kH * kH;
    o.w3 = o.b2 + kH;
    o.b3 = o.w3 + kH * kH;
    o.w4 = o.b3 + kH;
    o.b4 = o.w4 + O * kH;
    o.total = o.b4 + O;
    return o;
}
// flat parameter vector: [policy net (4 outputs) | value net (1 output) | log_std[4]]
__host__ __device__ inline int ppo_num_params(int L) { return net_off(L, 4).total + net_off(L, 1).total + 4; }
__host__ __device__ inline int rho_(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int L>
struct PpoDims {
    using P = PolicyDims<L>;
    static constexpr int kIT = (L + 1 + 31) / 32;  // 32-wide tiles of the layer-1 input (incl. the constant 1)
    // half8 per net: the forward operand image of quadrace_policy.hpp and nothing else.  The backward pass needs W^T operands
    // (lane = input unit, k = output units): it reads them out of the SAME image with ds_read_b64_tr_b16 (see lds_tr_pair).
    static constexpr int kImage = P::kTotalHalf8;
    // Partials of the role-split gradient kernel in ACCUMULATOR ORDER: one 32 x 32 weight tile = 1024 slots
    // [quad a = r >> 2][lane][k = r & 3] -- what a dW wave holds, written as four 1 KB runs per tile (store_dw_tile_raw);
    // tiles per net: layer 1 [out tile][in tile], layers 2 and 3 [out tile][in tile] (4 x 4), layer 4 [in tile]
    static constexpr int kRawT2 = 4 * kIT, kRawT3 = kRawT2 + 16, kRawT4 = kRawT3 + 16, kRawTiles = kRawT4 + 4;
    static constexpr int kRawSlots = kRawTiles * 1024;   // per net and workgroup (L = 24: 40 tiles = 160 KB of f32)
};

// The apply kernel's threads enumerate the REAL slots of one net's accumulator-order partial densely, in storage order: layer by
// layer, tile by tile, and inside a tile [row quad (a, h)][column c][k] over the tile's real rows and columns only (a layer's 120
// output rows end in the fourth tile row after 6 of its 8 quads, its in_dim + 1 columns somewhere in the last tile column) -- so a
// wave's 64 threads read (nearly) one contiguous run per chunk, and no thread is spent on padding: kDenseThreads = the net's
// parameter count (value net: + 363, the three unused rows of the output layer's stored quad).
// t -> slot in the partial and index inside the net's block of the flat parameter vector (-1: the unused rows just mentioned).
template <int L>
struct DenseMap {
    using D = PpoDims<L>;
    static constexpr int W1 = L + 1, WH = kH + 1;
    static constexpr int kN1 = kH * W1, kNH = kH * WH, kN4 = 4 * WH;
    static constexpr int kDenseThreads = kN1 + 2 * kNH + kN4;
    // one layer: W = in_dim + 1 columns in NTI tile columns, row blocks of 8 quads (the last: QL)
    template <int W, int NTI, bool kOutputLayer>
    static __device__ __forceinline__ void decode(int u, int& to, int& ti, int& a4, int& h, int& c, int& k) {
        constexpr int VL = W - 32 * (NTI - 1);   // real columns of the last tile column
        int quads = 1;
        to = 0;
        if constexpr (!kOutputLayer) {
            constexpr int full = 8 * W * 4;      // dense threads of a full row block (32 rows)
            to = u / full;
            to = to > 3 ? 3 : to;
            u -= to * full;
            quads = to < 3 ? 8 : 6;              // 120 = 3 x 32 + 24
        }
        const int tsz = quads * 128;             // a full-width tile of this row block
        ti = u / tsz;
        ti = ti > NTI - 1 ? NTI - 1 : ti;
        u -= ti * tsz;
        k = u & 3;
        const int cc = u >> 2;
        const int ah = ti == NTI - 1 ? cc / VL : cc >> 5;
        c = cc - ah * (ti == NTI - 1 ? VL : 32);
        a4 = ah >> 1;
        h = ah & 1;
    }
    static __device__ __forceinline__ void map(int t, int O, int& slot, int& param) {
        const NetOff o = net_off(L, O);
        int to, ti, a4, h, c, k, T, in_dim, out_dim, woff, boff;
        if (t < kN1) { decode<W1, D::kIT, false>(t, to, ti, a4, h, c, k); T = to * D::kIT + ti; in_dim = L; out_dim = kH; woff = o.w1; boff = o.b1; }
        else if (t < kN1 + kNH) { decode<WH, 4, false>(t - kN1, to, ti, a4, h, c, k); T = D::kRawT2 + 4 * to + ti; in_dim = kH; out_dim = kH; woff = o.w2; boff = o.b2; }
        else if (t < kN1 + 2 * kNH) { decode<WH, 4, false>(t - kN1 - kNH, to, ti, a4, h, c, k); T = D::kRawT3 + 4 * to + ti; in_dim = kH; out_dim = kH; woff = o.w3; boff = o.b3; }
        else { decode<WH, 4, true>(t - kN1 - 2 * kNH, to, ti, a4, h, c, k); T = D::kRawT4 + ti; in_dim = kH; out_dim = O; woff = o.w4; boff = o.b4; }
        slot = ((T * 4 + a4) * 64 + 32 * h + c) * 4 + k;
        const int row = 32 * to + k + 8 * a4 + 4 * h, col = 32 * ti + c;
        param = row < out_dim ? (col == in_dim ? boff + row : woff + row * in_dim + col) : -1;
    }
};

// ---- pack: f32 parameters -> f16 operand images -----------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(256) ppo_pack_kernel(const float* __restrict__ theta, half8* __restrict__ images) {
    using D = PpoDims<L>;
    using P = PolicyDims<L>;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int net = blockIdx.y;
    if (e >= D::kImage) return;
    const int O = net == 0 ? 4 : 1;
    const float* th = theta + (net == 0 ? 0 : net_off(L, 4).total);
    const NetOff o = net_off(L, O);
    const int lane = e & 63, c = lane & 31, h = lane >> 5;
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float val = 0.0f;
        if (e < P::kOff2) {  // layer 1: row = output unit, k = input index (natural order), input L = constant 1
            const int t = e / (P::kSteps1 * 64), s = (e / 64) % P::kSteps1;
            const int row = 32 * t + c, k = 16 * s + 8 * h + j;
            if (row < kH) val = k < L ? th[o.w1 + row * L + k] : (k == L ? th[o.b1 + row] : 0.0f);
            else val = (row == kPolBiasUnit && k == L) ? 1.0f : 0.0f;
        } else if (e < P::kOff4) {  // layers 2, 3: k-slots named in accumulator-row order
            const bool third = e >= P::kOff3;
            const int e2 = e - (third ? P::kOff3 : P::kOff2);
            const int t = e2 / 512, sp = (e2 / 64) % 8;
            const int row = 32 * t + c, hid = 32 * (sp >> 1) + rho_(8 * (sp & 1) + j, h);
            const int ow = third ? o.w3 : o.w2, ob = third ? o.b3 : o.b2;
            if (row < kH) val = hid < kH ? th[ow + row * kH + hid] : (hid == kPolBiasUnit ? th[ob + row] : 0.0f);
            else val = (row == kPolBiasUnit && hid == kPolBiasUnit) ? 1.0f : 0.0f;
        } else {  // output layer: rows 0..O-1 of one 32-row tile
            const int sp = (e - P::kOff4) / 64;
            const int hid = 32 * (sp >> 1) + rho_(8 * (sp & 1) + j, h);
            if (c < O) val = hid < kH ? th[o.w4 + c * kH + hid] : (hid == kPolBiasUnit ? th[o.b4 + c] : 0.0f);
        }
        v[j] = (_Float16)val;
    }
    images[(size_t)net * D::kImage + e] = v;
}

// ---- adv_stats: sum and sum of squares of the advantages of minibatch mb = blockIdx.y, rows idx[mb * B + 0..B), into
// table[mb][0..1] (the table is zeroed by a memset before the launch; the gradient kernel turns the sums into mean / rstd) ----------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the block of up to two values: wave shuffles, then the first wave adds the per-wave sums; valid in thread 0
template <int kThreads>
__device__ __forceinline__ void block_sum2_f64(double& a, double& b) {
    __shared__ double red[2][kThreads / 64];
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        a = wave_sum_f64((int)threadIdx.x < kThreads / 64 ? red[0][threadIdx.x] : 0.0);
        b = wave_sum_f64((int)threadIdx.x < kThreads / 64 ? red[1][threadIdx.x] : 0.0);
    }
}

// ---- epoch permutation on the device: a keyed bijection instead of a sort ---------------------------------------------------------
// SB3 draws np.random.permutation(rows) per epoch; torch.randperm on the GPU is a radix sort of random keys (0.24 ms for 2 Mi rows:
// 5 % of an epoch of matrix-core updates).  perm[i] = E_k(i) with E_k an 8-round alternating (unbalanced) Feistel network over
// ceil(log2 n) bits, cycle-walked into [0, n): every round XORs one half with a keyed hash of the other, so E_k is a bijection for
// ANY key; the key is (seed, epoch count) -- the count lives on the device (PpoCtrl::shuffle_count, bumped by the statistics kernel
// that follows in the same graph), so a replayed graph shuffles differently every epoch.  O(1) per element, no scratch, 20 us.
__device__ __forceinline__ uint32_t shuffle_mix(uint32_t x, uint32_t k) {
    uint32_t h = x * 0x9E3779B1u + k;
    h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
    return h;
}
// element i of the permutation of [0, n) keyed by (seed, c)
__device__ __forceinline__ unsigned int shuffle_index(unsigned int i, unsigned int n, unsigned long long seed, unsigned long long c) {
    unsigned int bits = 2;
    while ((1ull << bits) < n) ++bits;
    const unsigned int lb = bits >> 1, rb = bits - lb;
    const unsigned int lmask = (1u << lb) - 1u, rmask = (1u << rb) - 1u;
    uint32_t key[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {   // round keys: SplitMix64-style finaliser of (seed, epoch count, round) -- wave-uniform: scalar code
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (c * 8ull + (unsigned long long)r + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        key[r] = (uint32_t)(z ^ (z >> 31));
    }
    unsigned int x = i;
    do {   // cycle walking: the image of a value < n is revisited until it lands below n again (expected < 2 iterations)
        unsigned int l = x >> rb, r_ = x & rmask;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            l ^= shuffle_mix(r_, key[r]) & lmask;
            r_ ^= shuffle_mix(l, key[r + 1]) & rmask;
        }
        x = (l << rb) | r_;
    } while (x >= n);
    return x;
}
__global__ void __launch_bounds__(256) ppo_shuffle_kernel(int* __restrict__ perm, unsigned int n, unsigned long long seed,
                                                          const unsigned long long* __restrict__ count) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    perm[i] = (int)shuffle_index(i, n, seed, *count);
}

// Clears the advantage-sum table ahead of ppo_adv_stats_kernel.  A KERNEL, not hipMemsetAsync: inside qr_ppo_epoch's captured graph
// a memset node was seen to lose its ordering against the kernel nodes around it on REPLAY (ROCm 7.2: whole epochs whose sums
// were cleared mid-accumulation -> non-finite gradients, every update of the epoch skipped; tests/test_gpu_round2.py config-5 loop).
// PpoCtrl::lr <- lr (qr_ppo_epoch): a kernel node, not a host-to-device copy -- the value travels in the kernel arguments, so nothing on
// the host has to outlive the launch, and captured into a caller's graph the node replays the value it was captured with
__global__ void ppo_set_lr_kernel(float* __restrict__ dst, float lr) { *dst = lr; }

__global__ void __launch_bounds__(256) ppo_zero_table_kernel(double* __restrict__ table, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) table[i] = 0.0;
}

// (same-address f64 atomics serialise at ~15 ns each: one pair per 1024-thread block, not one per wave)
__global__ void __launch_bounds__(1024) ppo_adv_stats_kernel(const float* __restrict__ adv, const int* __restrict__ idx, int B,
                                                             double* __restrict__ table, unsigned long long* __restrict__ bump) {
    if (bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump += 1ull;   // the shuffle kernel BEFORE this launch has read it
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const double x = i < B ? (double)adv[idx[(size_t)blockIdx.y * B + i]] : 0.0;
    double s1 = x, s2 = x * x;
    block_sum2_f64<1024>(s1, s2);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(table + 2 * blockIdx.y + 0, s1);
        unsafeAtomicAdd(table + 2 * blockIdx.y + 1, s2);
    }
}

// Device-resident control block of one qr_ppo handle
struct PpoCtrl {
    // Grid barrier of ppo_apply_kernel.  arrive[b] = (generation workgroup b has arrived in) << 32 | f32 bits of its squared-gradient
    // sum: ONE 64-bit store per workgroup carries flag and value (no same-address read-modify-write atomics: 247 of them serialise
    // at ~100 ns each), so the master has the sums the moment it has seen the flags.  go = generation << 32 | f32 bits of the total,
    // with the decision bits below in the generation word: every workgroup learns "released", the norm and the verdict from one load.
    unsigned long long arrive[512];
    unsigned long long go;
    unsigned int gen;             // completed barrier generations (= apply launches that were not skipped)
    unsigned int pad_;
    int stop;                     // sticky: a minibatch exceeded 1.5 x target_kl (SB3's early stop); cleared by qr_ppo_control
    int applied;                  // optimiser steps taken since the last qr_ppo_control
    int skipped_nonfinite;        // updates dropped because the gradient norm was not finite
    int barrier_timeouts;         // must stay 0
    // Device-resident optimiser state for launches whose arguments must not change between replays of a captured graph
    // (qr_ppo_epoch) -- and so that the Adam step count advances only when a step was really TAKEN (a launch turned into a no-op by
    // the early stop or a non-finite norm does not count: torch.optim.Adam / SB3 count real steps).
    int adam_t;                   // optimiser steps taken over the life of the parameters (bias correction uses adam_t + 1)
    float lr;                     // learning rate of the launches qr_ppo_epoch enqueues (ApplyArgs::device_lr)
    unsigned long long shuffle_count;   // epochs shuffled on the device so far (ppo_shuffle_kernel's stream position)
};
constexpr unsigned int kGoStop = 0x40000000u, kGoNonFinite = 0x80000000u, kGoGenMask = 0x3FFFFFFFu;

// ---- one minibatch as the gradient kernel sees it ---------------------------------------------------------------------------
struct PpoBatch {
    const float* obs;       // [rows][L]
    const float* act;       // [rows][4]
    const float* old_logp;  // [rows]
    const float* adv;       // [rows]
    const float* ret;       // [rows]
    const int* idx;         // [B] rows of this minibatch
    int B, G;               // G = B / 64 groups
    float clip, vf_coef, ent_coef;
    const double* acc;       // [sum adv, sum adv^2] of this minibatch (ppo_adv_stats_kernel)
    const int* stop;         // PpoCtrl::stop: set by an earlier launch when the target-KL early stop hit -> nothing left to do
    const float* theta;      // flat parameters (log_std is read from here)
    const half8* images;     // [2][kImage]
    float* wave_out;         // [2][2 G][8] per-wave sums: policy waves {d log_std[4] / B, surrogate loss, approx kl, clipped, -},
                             // value waves {-, -, -, -, squared error, ...}; reduced by the norm kernel (no atomics)
    float* stats;            // [0] sum surrogate loss, [1] sum squared value error, [2] sum approx kl, [3] clipped count
#ifdef QR_PHASE_TIMING
    unsigned long long* ticks;  // [waves][16] shader-clock stamps (profiling build only, tools/ppo_phase_timing.py)
#endif
};

#ifdef QR_PHASE_TIMING
#define QR_TICK_GATE true   /* ppo_grad_kernel: -DQR_TICK_PASS=p stamps pass p only (a later pass overwrites an earlier one's stamps) */
#ifdef QR_PHASE_TIMING_NODRAIN   /* stamps without draining the queues: where the waves ARE, not what a stage costs in isolation */
#define PPO_TICK_DRAIN() asm volatile("" ::: "memory")
#else
#define PPO_TICK_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#define PPO_TICK(a, slot)                                                                                       \
    do {                                                                                                        \
        PPO_TICK_DRAIN();                                                                                       \
        if ((a).ticks && (threadIdx.x & 63) == 0 && QR_TICK_GATE)                                               \
            (a).ticks[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6)) * 16 + (slot)] = clock64(); \
    } while (0)
// constant-rate (100 MHz) device-wide clock: workgroup start / end skew across the grid, and the apply kernel's stages
#define PPO_WALL(ptr, index) do { if ((ptr) && (threadIdx.x & 63) == 0) (ptr)[index] = wall_clock64(); } while (0)
#else
#define PPO_TICK(a, slot) do { } while (0)
#define PPO_WALL(ptr, index) do { } while (0)
#endif

typedef float f32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half8 plain_pack(const f32x16p& acc, int s) {
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)acc[8 * s + j];
    return b;
}
// ReLU masks as bits, built and applied on packed f16 pairs.  One 32-bit word per pair of output tiles: the dword d of
// pack s of tile t (accumulator registers 8 s + 2 d and 8 s + 2 d + 1) owns bit  sh = 8 (t & 1) + 4 s + d  for its low half
// and bit 16 + sh for its high half.
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
typedef unsigned short ushort2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 mask_pack(const f32x16p& acc, uint32_t word, int t, int s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = acc[8 * s + j];
    u32x4p p = __builtin_bit_cast(u32x4p, sat_pack(v));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t on = (word >> (8 * (t & 1) + 4 * s + d)) & 0x00010001u;
        asm("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(p[d]) : "v"(on));   // bits x {0, 1} per half (see epilogue_dword)
    }
    return __builtin_bit_cast(half8, p);
}

// One dword (two values) of a layer's epilogue, d = 0..15 within a pair of output tiles: tile-in-pair ti = d >> 3, pack
// sh = (d >> 2) & 1, dword dd = d & 3 -> accumulator registers 8 sh + 2 dd (+1) of acc[ti].  Forward: relu + saturation +
// f16 pack, and the unit-active bits into `word`; backward: saturation + pack, zeroed where the forward unit was inactive.
// The f32 -> f16 conversion stays compiler-visible (it reads MFMA results: hazard wait states); the max / min pair is inline
// asm so that the dword stays in the issue slot the source gives it (see mlp_layer).
template <bool BWD>
__device__ __forceinline__ uint32_t epilogue_dword(const f32x16p (&acc)[2], int d, uint32_t& word) {
    const int ti = d >> 3, sh = (d >> 2) & 1, dd = d & 3;
    const int shift = 8 * ti + 4 * sh + dd;
    if (!BWD) {
        const uint32_t r = relu_pack2(acc[ti][8 * sh + 2 * dd], acc[ti][8 * sh + 2 * dd + 1]);
        // relu output >= 0: its f16 bit pattern is non-zero iff the unit is active; min(bits, 1) per half -> 0 / 1.  ONE packed
        // instruction, written out: the compiler expanded the vector min into two 16-bit compares, two selects and a v_perm per dword
        // (6 issue slots where the forward pass of a lone wave is issue-bound: 9 VALU per dword against 2 MFMAs per 2 dwords)
        // (the OR into `word` sits in the same asm: left to the compiler the 32 ORs of a layer were re-associated into a tree at the
        // END of the layer and the 32 intermediate values spilled)
        uint32_t on;
        asm("v_pk_min_u16 %1, %2, %3\n\tv_lshl_or_b32 %0, %1, %4, %0" : "+v"(word), "=&v"(on) : "v"(r), "s"(0x00010001u), "n"(shift));
        return r;
    }
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    const half2p c = {(_Float16)acc[ti][8 * sh + 2 * dd], (_Float16)acc[ti][8 * sh + 2 * dd + 1]};
    uint32_t r = __builtin_bit_cast(uint32_t, c);
    asm("v_pk_max_f16 %0, %0, %1\n\tv_pk_min_f16 %0, %0, %2" : "+v"(r) : "v"(0xFBFFFBFFu), "v"(0x7BFF7BFFu));  // +-65504
    const uint32_t on = (word >> shift) & 0x00010001u;
    uint32_t out;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(out) : "v"(r), "v"(on));   // bits x {0, 1} per half: the value or zero, one instruction
    return out;
}

// Backward epilogue dword with the ReLU derivative taken from the FORWARD ACTIVATION itself (`hdw` = the same dword of h, two f16 >= 0:
// active iff its bits are non-zero -- the criterion the mask bits record): saturation + pack, zero where the unit was inactive.
__device__ __forceinline__ uint32_t epilogue_dword_h(const f32x16p (&acc)[2], int d, uint32_t hdw) {
    const int ti = d >> 3, sh = (d >> 2) & 1, dd = d & 3;
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    const half2p c = {(_Float16)acc[ti][8 * sh + 2 * dd], (_Float16)acc[ti][8 * sh + 2 * dd + 1]};
    uint32_t r = __builtin_bit_cast(uint32_t, c);
    asm("v_pk_max_f16 %0, %0, %1\n\tv_pk_min_f16 %0, %0, %2" : "+v"(r) : "v"(0xFBFFFBFFu), "v"(0x7BFF7BFFu));  // +-65504
    uint32_t on, out;
    asm("v_pk_min_u16 %1, %2, %3\n\tv_pk_mul_lo_u16 %0, %4, %1" : "=v"(out), "=&v"(on) : "v"(hdw), "s"(0x00010001u), "v"(r));
    return out;
}
__device__ __forceinline__ half8 mask_pack_h(const f32x16p& acc, int s, const u32x4p& hpack) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = acc[8 * s + j];
    u32x4p p = __builtin_bit_cast(u32x4p, sat_pack(v));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t on;
        asm("v_pk_min_u16 %1, %2, %3\n\tv_pk_mul_lo_u16 %0, %0, %1" : "+v"(p[d]), "=&v"(on) : "v"(hpack[d]), "s"(0x00010001u));
    }
    return __builtin_bit_cast(half8, p);
}

// ---- W^T operands out of the forward image: ds_read_b64_tr_b16 (gfx950) ------------------------------------------------------
// The backward pass multiplies with W^T: A operand lane (c', h') = input unit i = 32 t' + c' of the layer, its 8 halves = the
// OUTPUT units oo = 16 sp' + 4 ((j >> 2) * 2 + h') + (j & 3) -- the same k-slot naming as the forward operands.  The forward
// image stores, for an output row, runs of 4 consecutive input units as 8 contiguous bytes (input 16 sp + 4 q + jj of output row
// 32 t + c sits at half8 index off + 512 t + 64 sp + 32 (q & 1) + c, halves 4 (q >> 1) + jj), and ds_read_b64_tr_b16 is a 4 x 16
// transpose inside every 16-lane group (measured, tools/ubench/tr_read.hip): lane n receives, as its k-th half, element n % 4 of
// the 8 bytes addressed by lane 4 k + n / 4.  So lane 4 k + m of a group points at [output unit oo_base + k][input units
// i0 + 4 m ..+3] and lane n gets W[oo_base + 0..3][i0 + n]: one read = halves j = 0..3, a second one 128 bytes on (8 output rows
// later) = halves 4..7.  Addresses = one lane constant + an immediate per operand; no second copy of the weights in LDS, none
// to stage (68 of 148 KB per workgroup) and none to re-pack after the optimiser step.
typedef unsigned u32x2p __attribute__((ext_vector_type(2)));
template <int kByteOff>
__device__ __forceinline__ u32x2p lds_tr_read(unsigned addr) {
    u32x2p v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(kByteOff));
    return v;
}
template <int kByteOff, int kSecond>
__device__ __forceinline__ half8 lds_tr_pair(unsigned addr) {
    const u32x2p lo = lds_tr_read<kByteOff>(addr), hi = lds_tr_read<kByteOff + kSecond>(addr);
    const u32x4p r = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(half8, r);
}
template <int kByteOff>
__device__ __forceinline__ half8 lds_tr_pair2(unsigned addr0, unsigned addr1) {
    const u32x2p lo = lds_tr_read<kByteOff>(addr0), hi = lds_tr_read<kByteOff>(addr1);
    const u32x4p r = {lo.x, lo.y, hi.x, hi.y};
    return __builtin_bit_cast(half8, r);
}
// Lane constants of the addresses (bytes, relative to the start of the layer images): hidden layers / output layer, for the first
// (second = 0) and second read of an operand pair.  Chunk index bits: k = source row (bits 0-1), then h' and `second` (hidden
// layers: bit 2 = h', bit 3 = second; output layer: bit 2 = second, bit 3 = h'), b = lane & 1 (bit 5), a = 16-lane group parity
// (bit 6).  Swizzled image: bits 2-3 ^= (b, a).
__device__ __forceinline__ unsigned tr_lane_hidden(int lane, int second = 0, bool swz = false) {
    const int a = (lane >> 4) & 1, b = lane & 1, hh = lane >> 5, k = (lane >> 2) & 3, mhi = (lane >> 1) & 1;
    const int bit2 = swz ? hh ^ b : hh, bit3 = swz ? second ^ a : second;
    return 16u * (unsigned)(64 * a + 32 * b + 4 * bit2 + 8 * bit3 + k) + 8u * (unsigned)mhi;
}
__device__ __forceinline__ unsigned tr_lane_out(int lane, int second = 0, bool swz = false) {
    const int a = (lane >> 4) & 1, b = lane & 1, hh = lane >> 5, k = (lane >> 2) & 3, mhi = (lane >> 1) & 1;
    const int bit2 = swz ? second ^ b : second, bit3 = swz ? hh ^ a : hh;
    return 16u * (unsigned)(64 * a + 32 * b + 4 * bit2 + 8 * bit3 + k) + 8u * (unsigned)mhi;
}
// The compiler does not see these reads: before the operands of one K-step group are used, wait until at most `kLater` LDS
// operations issued after them are outstanding (LDS returns in order); tying the wait to the registers keeps the MFMAs behind it.
template <int kLater>
__device__ __forceinline__ void lds_tr_wait(half8& a0, half8& a1) {
    u32x4p x = __builtin_bit_cast(u32x4p, a0), y = __builtin_bit_cast(u32x4p, a1);
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(kLater));
    a0 = __builtin_bit_cast(half8, x);
    a1 = __builtin_bit_cast(half8, y);
}

// Backward through one hidden layer for ONE 32-sample tile: out = (W^T in) where the forward unit was active (`mask`), W^T
// operands read out of the layer's forward image at half8 offset kFwdOff.  Same pinned schedule as mlp_layer (two output tiles
// side by side, epilogue of the previous pair in the shadow of the MFMAs), operand ring 3 deep = 12 reads in flight.
// Two lane-address registers (first / second read of an operand pair): for the plain image they differ by 128 bytes, for the
// swizzled one (ppo_grad_kernel) by the swizzle too -- see tr_lane_hidden().
// kHMask (ppo_grad_kernel, round 5): the ReLU derivative comes from the layer's forward activations, which the wave has just published
// to the exchange area as its own natural packs (hm_e / hm_o = this wave's and lane's pack 0, for even / odd pack index: the
// chunk swizzle of packs_to_lds) -- read back two pairs of output tiles ahead; `mask` is then unused and the forward pass builds no
// mask words (2 issue slots per dword less where a lone wave is issue-bound, 1 less here).
template <int kFwdOff, bool kHMask = false>
__device__ __forceinline__ void mlp_layer_bwd(unsigned tr_lane_addr0, unsigned tr_lane_addr1, const half8 (&in)[8], half8 (&out)[8],
                                              const uint32_t (&mask)[2], const u32x4p* hm_e = nullptr, const u32x4p* hm_o = nullptr) {
    const unsigned tr_addr0 = tr_lane_addr0 + 16u * (unsigned)kFwdOff;   // start of this layer's image (the immediates are 16-bit)
    const unsigned tr_addr1 = tr_lane_addr1 + 16u * (unsigned)kFwdOff;
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int KS = 8, D = 3, kGroups = 2 * KS;
    half8 a0[D], a1[D];
    f32x16p acc[2][2];   // [pair parity][tile in pair]
    u32x4p o32[8];
    uint32_t word[2] = {mask[0], mask[1]};
#define QR_TR_FETCH(Q, SLOT)                                                                                               \
    do {                                                                                                                   \
        constexpr int tp_ = (Q) / KS, s_ = (Q) % KS;                                                                       \
        constexpr int c0_ = 16 * (512 * (s_ >> 1) + 128 * (2 * tp_) + 16 * (s_ & 1));   /* 16-bit immediate: layer-relative */  \
        a0[SLOT] = lds_tr_pair2<c0_>(tr_addr0, tr_addr1);                                                                  \
        a1[SLOT] = lds_tr_pair2<c0_ + 16 * 128>(tr_addr0, tr_addr1);                                                       \
    } while (0)
    u32x4p hmv[4];   // pack j of the pair whose epilogue runs (steps 2 j, 2 j + 1); re-loaded in place for the second pair
    if constexpr (kHMask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) hmv[j] = ((j & 1) ? hm_o : hm_e)[j * 64];          // packs 0..3: output tiles 0, 1
    }
    QR_TR_FETCH(0, 0); QR_TR_FETCH(1, 1); QR_TR_FETCH(2, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tp = 0; tp <= 2; ++tp) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = tp * KS + s;
            if (tp < 2) {
                // reads issued after group q's: the (up to) two later groups in the ring, 4 reads each
                if (q + 2 < kGroups) lds_tr_wait<8>(a0[q % D], a1[q % D]);
                else if (q + 1 < kGroups) lds_tr_wait<4>(a0[q % D], a1[q % D]);
                else lds_tr_wait<0>(a0[q % D], a1[q % D]);
                acc[tp][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[q % D], in[s], s == 0 ? zero : acc[tp][0], 0, 0, 0);
                acc[tp][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[q % D], in[s], s == 0 ? zero : acc[tp][1], 0, 0, 0);
                switch (q + D) {   // compile-time after unrolling: the group D steps ahead goes into the slot just consumed
#define QR_TR_CASE(Q) case Q: QR_TR_FETCH(Q, (Q) % D); break;
                    QR_TR_CASE(3) QR_TR_CASE(4) QR_TR_CASE(5) QR_TR_CASE(6) QR_TR_CASE(7) QR_TR_CASE(8) QR_TR_CASE(9)
                    QR_TR_CASE(10) QR_TR_CASE(11) QR_TR_CASE(12) QR_TR_CASE(13) QR_TR_CASE(14) QR_TR_CASE(15)
#undef QR_TR_CASE
                    default: break;
                }
            }
            if (tp > 0) {
                const int p = tp - 1;
#pragma unroll
                for (int d = (16 * s) / KS; d < (16 * (s + 1)) / KS; ++d) {
                    if constexpr (kHMask) o32[4 * p + (d >> 2)][d & 3] = epilogue_dword_h(acc[p], d, hmv[d >> 2][d & 3]);
                    else o32[4 * p + (d >> 2)][d & 3] = epilogue_dword<true>(acc[p], d, word[p]);   // out[2 (2p + ti) + sh]
                }
                if constexpr (kHMask) {   // pack s / 2 of the first pair is used up: the second pair's takes its registers (8 steps ahead)
                    if (tp == 1 && (s & 1)) hmv[s >> 1] = (((s >> 1) & 1) ? hm_o : hm_e)[(4 + (s >> 1)) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef QR_TR_FETCH
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = __builtin_bit_cast(half8, o32[k]);
}

// One 128-unit layer for ONE 32-sample tile: in[KS] -> out[8].  Forward (BWD = false): out = relu(acc) packed,
// mask = (acc > 0).  Backward (BWD = true): out = acc where mask is set (the ReLU derivative of the layer being
// entered), else 0.  Two output tiles are accumulated side by side (two independent MFMA chains).
//
// The schedule is written out and pinned (sched_barrier), like policy_layer: left to itself the scheduler emitted
// "ds_read, s_waitcnt lgkmcnt, MFMA" per K-step (LDS latency exposed 148 times per wave, one wave per SIMD) and lumps of
// pack instructions behind s_nop 7-10 hazard waits.  K-step group q = (pair tp, step s):
//     2 MFMAs on ring slot q % D     |     ds_read of group q + D's operands into that slot (D = 4 K-steps ahead)
//     1/KS of the epilogue of the PREVIOUS pair (its accumulators finished a pair ago: no hazard wait)
// kSwz: the image in LDS is chunk-swizzled (ppo_grad_kernel: bits 2-3 of the 16-byte chunk index XORed with bits 5-6, so that the
// backward pass's transposed reads are bank-conflict free); chunk row r = tile * KS + s then sits at lane ^ ((h | (r & 1) << 1) << 2).
template <int KS, bool BWD, bool kSwz = false, bool kNoMask = false>
__device__ __forceinline__ void mlp_layer(const half8* __restrict__ W, int lane, const half8 (&in)[KS], half8 (&out)[8],
                                          uint32_t (&mask)[2]) {
    const int lane_e = kSwz ? lane ^ ((lane >> 5) << 2) : lane;           // chunk rows (tile * KS + s) of even / odd parity
    const int lane_o = kSwz ? lane ^ (((lane >> 5) | 2) << 2) : lane;
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int D = KS < 4 ? KS : 4;
    half8 a0[D], a1[D];
    // operands of group q: tiles 2 tp and 2 tp + 1 at K-step s
    auto fetch = [&](int q, int slot) {
        const int tp = q / KS, s = q % KS;
        a0[slot] = W[((2 * tp) * KS + s) * 64 + ((((2 * tp) * KS + s) & 1) ? lane_o : lane_e)];
        a1[slot] = W[((2 * tp + 1) * KS + s) * 64 + ((((2 * tp + 1) * KS + s) & 1) ? lane_o : lane_e)];
    };
#pragma unroll
    for (int q = 0; q < D; ++q) fetch(q, q);
    __builtin_amdgcn_sched_barrier(0);
    f32x16p acc[2][2];   // [pair parity][tile in pair]
    u32x4p o32[8];
    uint32_t word[2] = {BWD ? mask[0] : 0u, BWD ? mask[1] : 0u};
#pragma unroll
    for (int tp = 0; tp <= 2; ++tp) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int q = tp * KS + s;
            if (tp < 2) {
                acc[tp][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[q % D], in[s], s == 0 ? zero : acc[tp][0], 0, 0, 0);
                acc[tp][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[q % D], in[s], s == 0 ? zero : acc[tp][1], 0, 0, 0);
                if (q + D < 2 * KS) fetch(q + D, q % D);
            }
            if (tp > 0) {
                const int p = tp - 1;
#pragma unroll
                for (int d = (16 * s) / KS; d < (16 * (s + 1)) / KS; ++d) {
                    if constexpr (!BWD && kNoMask) {   // relu + saturation + pack only (the backward pass reads the activation back)
                        const int ti = d >> 3, sh = (d >> 2) & 1, dd = d & 3;
                        o32[4 * p + (d >> 2)][d & 3] = relu_pack2(acc[p][ti][8 * sh + 2 * dd], acc[p][ti][8 * sh + 2 * dd + 1]);
                    } else {
                        o32[4 * p + (d >> 2)][d & 3] = epilogue_dword<BWD>(acc[p], d, word[p]);   // out[2 (2p + ti) + sh]
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = __builtin_bit_cast(half8, o32[k]);
    if (!BWD) {
        mask[0] = word[0];
        mask[1] = word[1];
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

constexpr int kStashRows = 256;  // per-sample scalars of the workgroup's 4 x 64 samples, parked in LDS

// ---- gradient kernel: the exchange area ---------------------------------------------------------------------------------------
// One workgroup = 128 samples (4 tiles of 32) of ONE network per pass.  d_l and h_(l-1) go to a 64 KB exchange area in LDS as the chain
// waves' natural packs and are read back in the operand form (lane = unit, k = sample) with transposed LDS reads; only the narrow d4
// and x0 take the identity-MFMA route.  dW_l (16 tiles of 32 x 32 for a hidden layer) is split 2 x 2 over the four dW waves, K = the
// workgroup's 128 samples, accumulators live for one layer only (64 VGPRs) and leave as plain stores into partial[workgroup][slot]
// (f32 atomics were measured at ~0.3 lane-atomics per ns at any scope, tools/ubench/l2_atomics.hip; a workgroup that makes
// several passes adds to its own partial).  Per minibatch: <= 128 partials of one network each.
// LDS: 80 KB image | 64 KB exchange | 7 KB stash.
constexpr int kExHalf8 = 2 * 4 * 8 * 64;   // exchange area: [X = d | h][wave][k-step of the pack][lane] half8 (chunk-swizzled), or, for
                                           // d4^T / x0^T, [X][unit tile][k-step = 2 wave + s][lane] in the identity-MFMA form

// ---- exchange of the 128-wide matrices as NATURAL packs, read back transposed ------------------------------------------------
// A wave's pack array X[sp] (lane = sample c, half8 = 8 units of k-step sp) has the same shape as a one-row-tile weight image
// (rows = samples, k-slots = units), so the operand form (lane = unit 32 t + c', half8 = 8 samples) comes out of it with the same
// ds_read_b64_tr_b16 pair as W^T does out of the forward image -- no identity MFMAs, no f32 -> f16 conversion, no packing: the
// writer side is 8 plain 16-byte stores per matrix (the identity-MFMA transposes were measured at 3.8 k cycles per layer for the
// two matrices, a quarter of the kernel).  Region X (0 = d, 1 = h), wave w: half8 index ((4 X + w) * 8 + sp) * 64 + lane.
// Chunk (16 bytes) c5 = sample (bits 0-4), b = unit-slot parity (bit 5), a.. = k-step (bits 6-8).  A transposed read's 32-lane group
// touches the 16 chunks {k (2 bits), a, b} for fixed h' -- 64 a + 32 b apart they would share banks 4-way (256 B rows), so bits 2-3
// of the chunk index are XORed with (b, a): the group's chunks then fill one aligned 256-byte block = all 64 banks once.
__device__ __forceinline__ void packs_to_lds(const half8 (&X)[8], half8* __restrict__ Ex, int wave, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) Ex[(wave * 8 + sp) * 64 + (lane ^ ((h | ((sp & 1) << 1)) << 2))] = X[sp];
}
// lane constants of the two reads of an operand (second = halves 4..7 = 8 samples on): bit 2 of the chunk index is h' ^ b, bit 3 is
// second ^ a, so the two reads differ by more than an immediate and get one address register each
__device__ __forceinline__ unsigned ex_lane_const(int lane, int second) {
    const int a = (lane >> 4) & 1, b = lane & 1, hh = lane >> 5, k = (lane >> 2) & 3, mhi = (lane >> 1) & 1;
    return 16u * (unsigned)(64 * a + 32 * b + 4 * (hh ^ b) + 8 * (second ^ a) + k) + 8u * (unsigned)mhi;
}
// byte immediate of operand (region, wave w, unit tile offset bt from the base tile, sample half sp)
#define QR_EX_OFF(REGION, W, BT, SP) (16 * (((REGION) * 4 + (W)) * 512 + 128 * (BT) + 16 * (SP)))

// dW block of 2 x 2 tiles from natural packs: base_d / base_h = exchange address + lane constant + 2 KB x first tile.  K-steps
// (w, sp) = 8; per step four operands = 8 transposed reads, issued half a step at a time so that at most 12 are in flight
// (the counter has 4 bits): A(k) = d operands, B(k) = h operands; order A0 B0 A1 | wait, MFMAs(k), B(k+1), A(k+2).
__device__ __forceinline__ void dw_block_tr(unsigned base_d0, unsigned base_d1, unsigned base_h0, unsigned base_h1, f32x16p (&acc)[2][2]) {
    half8 a0[2], a1[2], b0[2], b1[2];
#define QR_EX_A(KQ, SLOT)                                                                           \
    do {                                                                                            \
        a0[SLOT] = lds_tr_pair2<QR_EX_OFF(0, (KQ) >> 1, 0, (KQ) & 1)>(base_d0, base_d1);            \
        a1[SLOT] = lds_tr_pair2<QR_EX_OFF(0, (KQ) >> 1, 1, (KQ) & 1)>(base_d0, base_d1);            \
    } while (0)
#define QR_EX_B(KQ, SLOT)                                                                           \
    do {                                                                                            \
        b0[SLOT] = lds_tr_pair2<QR_EX_OFF(1, (KQ) >> 1, 0, (KQ) & 1)>(base_h0, base_h1);            \
        b1[SLOT] = lds_tr_pair2<QR_EX_OFF(1, (KQ) >> 1, 1, (KQ) & 1)>(base_h0, base_h1);            \
    } while (0)
#define QR_EX_STEP(KQ, LATER)                                                                                     \
    do {                                                                                                          \
        lds_tr_wait<LATER>(a0[(KQ) & 1], a1[(KQ) & 1]);                                                           \
        lds_tr_wait<LATER>(b0[(KQ) & 1], b1[(KQ) & 1]);                                                           \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[(KQ) & 1], b0[(KQ) & 1], acc[0][0], 0, 0, 0);       \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[(KQ) & 1], b1[(KQ) & 1], acc[0][1], 0, 0, 0);       \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[(KQ) & 1], b0[(KQ) & 1], acc[1][0], 0, 0, 0);       \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[(KQ) & 1], b1[(KQ) & 1], acc[1][1], 0, 0, 0);       \
    } while (0)
    QR_EX_A(0, 0); QR_EX_B(0, 0); QR_EX_A(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(0, 4); QR_EX_B(1, 1); QR_EX_A(2, 0); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(1, 4); QR_EX_B(2, 0); QR_EX_A(3, 1); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(2, 4); QR_EX_B(3, 1); QR_EX_A(4, 0); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(3, 4); QR_EX_B(4, 0); QR_EX_A(5, 1); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(4, 4); QR_EX_B(5, 1); QR_EX_A(6, 0); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(5, 4); QR_EX_B(6, 0); QR_EX_A(7, 1); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(6, 4); QR_EX_B(7, 1); __builtin_amdgcn_sched_barrier(0);
    QR_EX_STEP(7, 0);
#undef QR_EX_STEP
#undef QR_EX_B
#undef QR_EX_A
}

// one tile: A operands = a tile in the identity-MFMA ("old") format at Eold[k-step][lane] (d4^T), B operands = unit tile of a
// natural-pack matrix in region 1 (base_h as above).  8 k-steps in two halves of 4 (8 transposed reads in flight).
template <int K0>
__device__ __forceinline__ void dw_tile_old_tr_half(const half8* __restrict__ Eold, unsigned base_h0, unsigned base_h1, int lane, f32x16p& acc) {
    half8 b0 = lds_tr_pair2<QR_EX_OFF(1, (K0 + 0) >> 1, 0, (K0 + 0) & 1)>(base_h0, base_h1);
    half8 b1 = lds_tr_pair2<QR_EX_OFF(1, (K0 + 1) >> 1, 0, (K0 + 1) & 1)>(base_h0, base_h1);
    half8 b2 = lds_tr_pair2<QR_EX_OFF(1, (K0 + 2) >> 1, 0, (K0 + 2) & 1)>(base_h0, base_h1);
    half8 b3 = lds_tr_pair2<QR_EX_OFF(1, (K0 + 3) >> 1, 0, (K0 + 3) & 1)>(base_h0, base_h1);
    const half8 x0 = Eold[(K0 + 0) * 64 + lane], x1 = Eold[(K0 + 1) * 64 + lane];
    const half8 x2 = Eold[(K0 + 2) * 64 + lane], x3 = Eold[(K0 + 3) * 64 + lane];
    lds_tr_wait<0>(b0, b1);
    lds_tr_wait<0>(b2, b3);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3, b3, acc, 0, 0, 0);
}

// NTI (1 or 2) tiles: A operands = unit tile of a natural-pack matrix in region 0 (base_d), B operands = tiles in the old format
// at Eold[(bi * 8 + k-step) * 64 + lane] (x0^T)
template <int NTI, int K0>
__device__ __forceinline__ void dw_tiles_tr_old_half(unsigned base_d0, unsigned base_d1, const half8* __restrict__ Eold, int lane, f32x16p (&acc)[2][2]) {
    half8 a0 = lds_tr_pair2<QR_EX_OFF(0, (K0 + 0) >> 1, 0, (K0 + 0) & 1)>(base_d0, base_d1);
    half8 a1 = lds_tr_pair2<QR_EX_OFF(0, (K0 + 1) >> 1, 0, (K0 + 1) & 1)>(base_d0, base_d1);
    half8 a2 = lds_tr_pair2<QR_EX_OFF(0, (K0 + 2) >> 1, 0, (K0 + 2) & 1)>(base_d0, base_d1);
    half8 a3 = lds_tr_pair2<QR_EX_OFF(0, (K0 + 3) >> 1, 0, (K0 + 3) & 1)>(base_d0, base_d1);
    const half8 y00 = Eold[(K0 + 0) * 64 + lane], y01 = Eold[(K0 + 1) * 64 + lane];
    const half8 y02 = Eold[(K0 + 2) * 64 + lane], y03 = Eold[(K0 + 3) * 64 + lane];
    lds_tr_wait<0>(a0, a1);
    lds_tr_wait<0>(a2, a3);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, y00, acc[0][0], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, y01, acc[0][0], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, y02, acc[0][0], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, y03, acc[0][0], 0, 0, 0);
    if (NTI > 1) {
        const half8 y10 = Eold[(8 + K0 + 0) * 64 + lane], y11 = Eold[(8 + K0 + 1) * 64 + lane];
        const half8 y12 = Eold[(8 + K0 + 2) * 64 + lane], y13 = Eold[(8 + K0 + 3) * 64 + lane];
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, y10, acc[0][1], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, y11, acc[0][1], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, y12, acc[0][1], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3, y13, acc[0][1], 0, 0, 0);
    }
}

// one 32 x 32 tile of dW (register r of lane (c, h) = dW[row 32 to + rho(r, h)][col 32 ti + c]) into this workgroup's partial;
// column kInDim is the constant-1 unit = the bias.  `add`: a later pass of the same workgroup (all 16 old values are loaded
// before the first add).  The row pitch is a compile-time constant, so the 16 rows are one base address + immediates.
// PT = float, or __bf16: the role-split kernel's 16-bit partials (qr_ppo::partial_bf16) -- v_cvt_pk_bf16_f32, round to nearest even,
// and a 2-byte store per element; the apply kernel widens them again before its fixed-order f32 sum.
template <int kInDim, typename PT = float>
__device__ __forceinline__ void store_dw_tile(const f32x16p& acc, PT* __restrict__ gw, PT* __restrict__ gb, int out_dim, int to, int ti,
                                              int lane, float scale, bool add) {
    const int c = lane & 31, h = lane >> 5;
    // Formed HERE: inside the pass loop of ppo_grad_kernel the row addresses are loop invariants, and hoisted out of the loop
    // they occupied ~250 registers for the whole kernel (spilled to AGPRs and scratch).  The empty asm makes the lane's column
    // opaque, so the address arithmetic stays next to the stores.
    int col = 32 * ti + c;
    asm volatile("" : "+v"(col));
    if (col > kInDim) return;
    const int row0 = 32 * to + 4 * h;
    const bool bias = col == kInDim;
    PT* base = bias ? gb + row0 : gw + row0 * kInDim + col;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] * scale;
    if (!bias) {
        if (add) {
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = row0 + rho_(r, 0) < out_dim ? (float)base[rho_(r, 0) * kInDim] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += old[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (row0 + rho_(r, 0) < out_dim) base[rho_(r, 0) * kInDim] = (PT)v[r];
    } else {
        if (add) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += row0 + rho_(r, 0) < out_dim ? (float)base[rho_(r, 0)] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (row0 + rho_(r, 0) < out_dim) base[rho_(r, 0)] = (PT)v[r];
    }
}

// The same tile in ACCUMULATOR ORDER (PpoDims::kRawSlots): quad a of lane l -> tile[(64 a + l) * 4 .. + 3], i.e. each of the four
// store instructions of a tile writes ONE contiguous, aligned run (512 B as bf16, 1 KB as f32) instead of 64 row segments of 128 B
// at odd offsets (row pitch 120 / 121 floats).  `rows` = real output rows left in this tile (quads are 4-row aligned: whole or none).
// The stores are agent-scope (sc1): written THROUGH the XCD's L2 as they are issued, so the partial of a finished layer drains to the
// fabric while the workgroup still computes the next one -- as plain stores the lines sat dirty in L2 until the end-of-kernel
// write-back (measured with tools/ppo_launch_timing.py: 6.6 us between the last workgroup's exit and the apply kernel's first
// instruction; 1.4 us with write-through stores issued per layer).
// kRows8: `rows` is a multiple of 8 (hidden layers: 32, or 24 in the last row block), so a quad is stored by both lane halves or by
// none -- a wave-uniform (scalar) branch per store instead of an EXEC-mask sequence
template <typename PT, bool kRows8 = false>
__device__ __forceinline__ void store_dw_tile_raw(const f32x16p& acc, PT* __restrict__ tile, int rows, int lane, float scale) {
    typedef PT pt4 __attribute__((ext_vector_type(4)));
    int ln = lane;
    asm volatile("" : "+v"(ln));   // (see store_dw_tile: keeps the address arithmetic next to the stores)
    const int h4 = 4 * (ln >> 5);
    pt4* out = reinterpret_cast<pt4*>(tile) + ln;
#pragma unroll
    for (int a4 = 0; a4 < 4; ++a4)
        if (kRows8 ? 8 * a4 < rows : 8 * a4 + h4 < rows) {
            pt4 v;
            v.x = (PT)(acc[4 * a4] * scale); v.y = (PT)(acc[4 * a4 + 1] * scale);
            v.z = (PT)(acc[4 * a4 + 2] * scale); v.w = (PT)(acc[4 * a4 + 3] * scale);
            if constexpr (sizeof(pt4) == 8)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + 64 * a4), __builtin_bit_cast(unsigned long long, v),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store_dwordx2 ... sc1
            else
                // s_nop 1: a store of more than 8 bytes reads its data registers AFTER issue -- two wait states before a VALU
                // instruction may overwrite them (gfx940 family).  The compiler's hazard recogniser pads its own stores and does not look
                // inside inline asm: as long as every store sat behind an EXEC-mask sequence the scalar instructions in between
                // covered it by accident; with wave-uniform branches (kRows8) the next tile's v_pk_mul_f32 followed at once and the
                // f32 partials of the hidden layers were corrupted (tests/test_gpu_ppo_kernel.py, f32 cases).
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(out + 64 * a4), "v"(v) : "memory");
        }
}

// ---- the gradient kernel: the workgroup's two jobs on different waves ---------------------------------------------------------
// One dependent chain per wave -- forward, loss, then per layer [exchange d_l / h_(l-1) through LDS, weight gradient dW_l (transposed
// reads + MFMAs + 64 stores per lane), backward to d_(l-1)] -- puts the weight-gradient half (exchange, barriers, dW, stores: 22 k of
// 44 k cycles in round 2's 4-wave kernel) on the critical path although nothing downstream needs dW.  Here a workgroup has EIGHT
// waves, two per SIMD:
//   chain waves 0-3   one 32-sample tile each: gather, forward, loss, d3, d2, d1 -- and they publish (d_l, h_(l-1)) to the exchange
//                     area as soon as d_l exists;
//   dW waves 4-7      dW_l = d_l^T h_(l-1) over the workgroup's 128 samples (2 x 2 tile blocks) and the
//                     partial stores, WHILE the chain waves are already in the next backward layer; their stores drain while they
//                     wait for the next operands.
// One exchange area (no room for two beside the 80 KB image), so per layer: chain writes -> barrier -> dW reads || chain computes the
// next delta -> barrier -> chain writes ...  Eight barriers per pass; 512 threads, 256 VGPRs per wave.  In a later pass of
// a large minibatch the chain waves' gather flies while the dW waves finish the previous pass.
// statistics of one chain wave: wave sums of the per-lane terms (policy: d log_std[4] (x 1/B), surrogate loss, approx kl, clipped
// count; value: squared error) -> wave_out; same order of additions whenever it is called
__device__ __forceinline__ void chain_stats_out(const PpoBatch& a, float (&wsum)[8], int net, bool live, int lane, int g, int et, float scale) {
    if (net == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wsum[k] = wave_sum(wsum[k]) * scale;
        wsum[4] = wave_sum(wsum[4]);
        wsum[5] = wave_sum(wsum[5]);
        wsum[6] = wave_sum(wsum[6]);
    } else {
        wsum[4] = wave_sum(wsum[4]);
    }
    if (live && lane == 0) {
        float4* wo = reinterpret_cast<float4*>(a.wave_out + (((size_t)net * a.G + g) * 2 + et) * 8);
        wo[0] = make_float4(wsum[0], wsum[1], wsum[2], wsum[3]);
        wo[1] = make_float4(wsum[4], wsum[5], wsum[6], wsum[7]);
    }
}

template <int L, typename PT>
__global__ void __launch_bounds__(512, 1) ppo_grad_kernel(PpoBatch a, PT* __restrict__ partial, int num_params) {
    using D = PpoDims<L>;
    using P = PolicyDims<L>;
    constexpr int KS1 = P::kSteps1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* W = reinterpret_cast<half8*>(smem);
    half8* E = W + D::kImage;
    const unsigned lds_base = (unsigned)(size_t)smem;
    const int net = blockIdx.y;
    const int stop_flag = *a.stop;
#if defined(QR_PHASE_TIMING) && defined(QR_TICK_PASS)
    const int pass = QR_TICK_PASS;   // (shadowed by the pass loops' own counter)
#undef QR_TICK_GATE
#define QR_TICK_GATE (pass == QR_TICK_PASS)
#endif
    PPO_TICK(a, 0);
#ifdef QR_PHASE_TIMING
    PPO_WALL(a.ticks, (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + 14);
#endif
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave8 >> 2;          // 0: chain wave, 1: dW wave (wave-uniform: every branch on it is whole-wave)
    const int wave = wave8 & 3;           // chain: the wave's 32-sample tile; dW: its 2 x 2 block of weight tiles
    const int et = wave & 1;
    const int O = net == 0 ? 4 : 1;
    PT* gn = partial + ((size_t)blockIdx.x * 2 + net) * D::kRawSlots;   // accumulator-order partial of (workgroup, net)
    const float scale = 1.0f / (float)a.B;
    const int pairs = (a.G + 1) / 2;   // passes of 2 sample groups = 128 samples
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // The pass loop is written out per role (loop unswitching by hand): inside ONE loop the dW waves' persistent accumulators would be
    // live through the chain waves' branch as well (the compiler does not know that a wave never changes its role) and spill.
    if (role == 0) {
    for (int pair = blockIdx.x, pass = 0; pair < pairs; pair += gridDim.x, ++pass) {
        // Everything derived from the lane index is formed INSIDE the pass loop, behind an opaque copy: as loop invariants the image
        // addresses, identity operands and lane constants were hoisted in front of the loop and kept alive through the whole kernel
        // (89 spilled registers and a scratch segment, i.e. the runtime's scratch set-up on every launch).
        int lane = (int)(threadIdx.x & 63), tid = (int)threadIdx.x;
        asm volatile("" : "+v"(lane), "+v"(tid));
        const int c = lane & 31, h = lane >> 5;
        float* stash = reinterpret_cast<float*>(E + kExHalf8) + wave * 32 + c;
            // =========================================== chain waves ===========================================================
            const int g = 2 * pair + (wave >> 1);
            const bool live = g < a.G;   // whole wave; a wave without samples runs on row 0 with zero deltas (it shares the barriers)
            // (Measured and dropped: under qr_ppo_epoch's device shuffle the row index is a computable bijection of the position, so
            // the kernel could form it instead of loading it -- one dependent round trip less, 3.4 k of the prologue's 7.7 k cycles
            // in a probe that skipped the index.  The 8-round Feistel + key derivation in front of the gather cost as much as the
            // trip saved: epoch graph 38.4 -> 39.4 us per update at 16 384 rows, 70 -> 75 us at 65 536.  tools/exp_patches/ppo_noidx.patch keeps the probe.)
            // a minibatch need not be a multiple of 64 rows (the reference's batch_size is 5000, R:792): positions past B in the last
            // group read the minibatch's last row and carry zero loss gradients, like the rows of a wave without samples
            const int pos = (live ? g : a.G - 1) * 64 + 32 * et + c;
            const bool row_ok = pos < a.B;
            const int b = a.idx[row_ok ? pos : a.B - 1];
            float xin[KS1][8];
            {
                const float* row = a.obs + (size_t)b * L;
#pragma unroll
                for (int s = 0; s < KS1; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = 16 * s + 8 * h + j;
                        xin[s][j] = row[k < L ? k : L - 1];
                    }
            }
            const float4 act_v = net == 0 ? reinterpret_cast<const float4*>(a.act)[b] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float old_logp_in = a.old_logp[b], adv_in = a.adv[b], ret_in = a.ret[b];
            const double acc_s1 = a.acc[0], acc_s2 = a.acc[1];
            float log_std_v[4];
            {
                const float* log_std = a.theta + net_off(L, 4).total + net_off(L, 1).total;
#pragma unroll
                for (int k = 0; k < 4; ++k) log_std_v[k] = log_std[k];
            }
            // Pass 0: the dW waves stage the operand image layer by layer and release one barrier per layer ([A] W1, [B] W2, [C] W3,
            // [S0] W4): the forward pass starts as soon as W1 is in LDS and runs while the other 72 KB arrive (staging the whole
            // image before the first layer was the long pole of the prologue: 8.2 k of 9.7 k cycles).  Later passes: [S0] only.
            __syncthreads();   // pass 0: [A] W1 staged / later passes: [S0] the dW waves have finished with the exchange area
            if (stop_flag) return;   // uniform over the grid
            PPO_TICK(a, 1);
            if (h == 0) {
                stash[0 * kStashRows] = act_v.x;
                stash[1 * kStashRows] = act_v.y;
                stash[2 * kStashRows] = act_v.z;
                stash[3 * kStashRows] = act_v.w;
                stash[4 * kStashRows] = old_logp_in;
                stash[5 * kStashRows] = adv_in;
                stash[6 * kStashRows] = ret_in;
            }
            half8 in[KS1];
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * h + j;
                    v[j] = k < L ? xin[s][j] : (k == L ? 1.0f : 0.0f);
                }
                in[s] = sat_pack(v);
            }
            const bool valid = h == 0 && live && row_ok;
            // ---- forward: the activations stay in registers until they have been published for their layer's weight gradient
            constexpr bool kHM = true;    // ReLU-derivative bits are taken from the published activations in the backward pass (see mlp_layer_bwd)
            uint32_t m1[2] = {0u, 0u}, m2[2] = {0u, 0u}, m3[2] = {0u, 0u};
            half8 h1[8], h2[8], h3[8];
            // this wave's own packs in the h region of the exchange area, as this lane wrote them (packs_to_lds): even / odd pack index
            const u32x4p* hm_e = reinterpret_cast<const u32x4p*>(E + 4 * 8 * 64 + wave * 8 * 64) + (lane ^ (h << 2));
            const u32x4p* hm_o = reinterpret_cast<const u32x4p*>(E + 4 * 8 * 64 + wave * 8 * 64) + (lane ^ ((h | 2) << 2));
            mlp_layer<KS1, false, true, kHM>(W, lane, in, h1, m1);
            PPO_TICK(a, 2);
            if (pass == 0) __syncthreads();   // [B] W2 staged
            mlp_layer<8, false, true, kHM>(W + P::kOff2, lane, h1, h2, m2);
            PPO_TICK(a, 3);
            if (pass == 0) __syncthreads();   // [C] W3 staged
            mlp_layer<8, false, true, kHM>(W + P::kOff3, lane, h2, h3, m3);
            PPO_TICK(a, 4);
            if (pass == 0) __syncthreads();   // [S0] W4 staged
            half8 w4[8], w4t[4];
#pragma unroll
            for (int s = 0; s < 8; ++s) w4[s] = W[P::kOff4 + s * 64 + (lane ^ ((h | ((s & 1) << 1)) << 2))];
            {
                const unsigned a40 = lds_base + tr_lane_out(lane, 0, true) + 16u * (unsigned)P::kOff4;
                const unsigned a41 = lds_base + tr_lane_out(lane, 1, true) + 16u * (unsigned)P::kOff4;
                w4t[0] = lds_tr_pair2<16 * 128 * 0>(a40, a41);
                w4t[1] = lds_tr_pair2<16 * 128 * 1>(a40, a41);
                w4t[2] = lds_tr_pair2<16 * 128 * 2>(a40, a41);
                w4t[3] = lds_tr_pair2<16 * 128 * 3>(a40, a41);
            }
            float out4[4];
            {
                f32x16p acc = zero;
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4[s], h3[s], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) out4[r] = acc[r];
            }
            // ---- per-sample loss gradients (x B; the 1/B of the batch means is applied when the tiles are stored)
            float wsum[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            const float* st_ = stash;
            float dout[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (net == 0) {
                const float act[4] = {st_[0 * kStashRows], st_[1 * kStashRows], st_[2 * kStashRows], st_[3 * kStashRows]};
                const float old_logp_v = st_[4 * kStashRows], adv_v = st_[5 * kStashRows];
                float z[4], inv_std[4], logp = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ls = log_std_v[k];
                    inv_std[k] = __expf(-ls);
                    z[k] = (act[k] - out4[k]) * inv_std[k];
                    logp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
                }
                const float log_ratio = valid ? logp - old_logp_v : 0.0f;
                const float ratio = __expf(log_ratio);
                const double amean = acc_s1 / a.B;
                const double avar = fmax((acc_s2 - a.B * amean * amean) / (a.B > 1 ? a.B - 1 : 1), 0.0);
                const float A = valid ? (adv_v - (float)amean) * (float)(1.0 / (sqrt(avar) + 1e-8)) : 0.0f;
                const bool flows = A >= 0.0f ? (ratio <= 1.0f + a.clip) : (ratio >= 1.0f - a.clip);
                const float gl = (flows && valid) ? -A * ratio : 0.0f;
                float dls[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dout[k] = gl * z[k] * inv_std[k];
                    dls[k] = gl * (z[k] * z[k] - 1.0f);
                }
                const float clipped_ratio = fminf(fmaxf(ratio, 1.0f - a.clip), 1.0f + a.clip);
                // per-lane terms of the minibatch statistics; their wave sums are formed AFTER d3 (below), where this wave would
                // otherwise wait for the dW waves at [S2]: nothing downstream in the chain needs them
#pragma unroll
                for (int k = 0; k < 4; ++k) wsum[k] = dls[k];
                wsum[4] = valid ? -fminf(A * ratio, A * clipped_ratio) : 0.0f;
                wsum[5] = valid ? (ratio - 1.0f) - log_ratio : 0.0f;
                wsum[6] = valid && fabsf(ratio - 1.0f) > a.clip ? 1.0f : 0.0f;
            } else {
                const float err = valid ? out4[0] - st_[6 * kStashRows] : 0.0f;
                dout[0] = a.vf_coef * 2.0f * err;
                wsum[4] = err * err;
            }
            PPO_TICK(a, 5);
            // ---- layer 4 operands: d4 (k-slot (h, j) = output unit 8 h + j) transposed by the identity MFMA, h3 as natural packs
            half8 d4;
            {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (j < 4 && valid) ? dout[j < 4 ? j : 0] : 0.0f;
                d4 = sat_pack(v);
                half8 id;
#pragma unroll
                for (int j = 0; j < 8; ++j) id[j] = (8 * h + j == c) ? (_Float16)1.0f : (_Float16)0.0f;
                const f32x16p acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(d4, id, zero, 0, 0, 0);
                E[(0 * 8 + 2 * wave) * 64 + lane] = plain_pack(acc, 0);
                E[(0 * 8 + 2 * wave + 1) * 64 + lane] = plain_pack(acc, 1);
            }
            packs_to_lds(h3, E + 4 * 8 * 64, wave, lane);
            __syncthreads();   // [S1] layer-4 operands published
            PPO_TICK(a, 6);
            // d3 = (W4^T d4) * relu'(z3) -- registers and the image only, while the dW waves form dW4
            half8 dA[8], dB[8];
            lds_tr_wait<0>(w4t[0], w4t[1]);
            lds_tr_wait<0>(w4t[2], w4t[3]);
            u32x4p hm3[2][4];   // h3 (published before [S1]) read back half a layer at a time: 16 registers, not 32
            if constexpr (kHM) {
#pragma unroll
                for (int k = 0; k < 4; ++k) hm3[0][k] = ((k & 1) ? hm_o : hm_e)[k * 64];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x16p acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4t[t], d4, zero, 0, 0, 0);
                if constexpr (kHM) {
                    if (t == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) hm3[1][k] = ((k & 1) ? hm_o : hm_e)[(4 + k) * 64];
                    }
                    dA[2 * t] = mask_pack_h(acc, 0, hm3[t >> 1][2 * (t & 1)]);
                    dA[2 * t + 1] = mask_pack_h(acc, 1, hm3[t >> 1][2 * (t & 1) + 1]);
                } else {
                    dA[2 * t] = mask_pack(acc, m3[t >> 1], t, 0);
                    dA[2 * t + 1] = mask_pack(acc, m3[t >> 1], t, 1);
                }
            }
            chain_stats_out(a, wsum, net, live, lane, g, et, scale);
            PPO_TICK(a, 7);
            __syncthreads();   // [S2] the dW waves are done reading the layer-4 operands
            packs_to_lds(dA, E, wave, lane);
            packs_to_lds(h2, E + 4 * 8 * 64, wave, lane);
            __syncthreads();   // [S3] layer-3 operands published
            PPO_TICK(a, 8);
            mlp_layer_bwd<P::kOff3, kHM>(lds_base + tr_lane_hidden(lane, 0, true), lds_base + tr_lane_hidden(lane, 1, true), dA, dB, m2, hm_e, hm_o);   // d2
            PPO_TICK(a, 9);
            __syncthreads();   // [S4]
            packs_to_lds(dB, E, wave, lane);
            packs_to_lds(h1, E + 4 * 8 * 64, wave, lane);
            __syncthreads();   // [S5] layer-2 operands published
            PPO_TICK(a, 10);
            mlp_layer_bwd<P::kOff2, kHM>(lds_base + tr_lane_hidden(lane, 0, true), lds_base + tr_lane_hidden(lane, 1, true), dB, dA, m1, hm_e, hm_o);   // d1
            PPO_TICK(a, 11);
            __syncthreads();   // [S6]
            // ---- layer 1 operands: d1 as natural packs, x0 transposed by the identity MFMA (column unit = input index)
            packs_to_lds(dA, E, wave, lane);
#pragma unroll
            for (int ut = 0; ut < D::kIT; ++ut) {
                f32x16p acc = zero;
#pragma unroll
                for (int s = 0; s < KS1; ++s) {
                    half8 id;
#pragma unroll
                    for (int j = 0; j < 8; ++j) id[j] = (16 * s + 8 * h + j == 32 * ut + c) ? (_Float16)1.0f : (_Float16)0.0f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(in[s], id, acc, 0, 0, 0);
                }
                E[((4 + ut) * 8 + 2 * wave) * 64 + lane] = plain_pack(acc, 0);
                E[((4 + ut) * 8 + 2 * wave + 1) * 64 + lane] = plain_pack(acc, 1);
            }
            __syncthreads();   // [S7] layer-1 operands published
            PPO_TICK(a, 12);
            // (a later pass: this wave's next gather is issued while the dW waves finish; [S0] keeps the exchange area and the stash)
        }
    } else {
        // The operand image (80 KB) is staged by THESE four waves -- they have nothing else to do until the first operands are
        // published and nothing else live in their registers -- while the chain waves' gather is in flight.  ALL loads are issued
        // up front in layer order (one burst of kImgLoads x 16 bytes per thread; loads return in order), then each layer is written
        // to LDS and released with its own barrier, so that the chain waves' forward pass overlaps the rest of the staging.
        // Every workgroup walks a layer's chunks in a different rotation: all workgroups of a network read the SAME 80 KB, and in
        // the same order they would all be queueing on one L2 channel at a time.
        {
            const f32x4p* src = reinterpret_cast<const f32x4p*>(a.images + (size_t)net * D::kImage);
            f32x4p* dst = reinterpret_cast<f32x4p*>(W);
            const int t256 = (int)(threadIdx.x & 255);
            constexpr int n1 = P::kOff2 / 256, n2 = (P::kOff3 - P::kOff2) / 256, n3 = (P::kOff4 - P::kOff3) / 256, n4 = (D::kImage - P::kOff4) / 256;
            static_assert(P::kOff2 % 256 == 0 && P::kOff3 % 256 == 0 && P::kOff4 % 256 == 0 && D::kImage % 256 == 0, "image layers are whole 256-chunk blocks");
            const int rot = (int)blockIdx.x;
            f32x4p i1[n1], i2[n2], i3[n3], i4[n4];
#pragma unroll
            for (int q = 0; q < n1; ++q) i1[q] = src[((q + rot) % n1) * 256 + t256];
#pragma unroll
            for (int q = 0; q < n2; ++q) i2[q] = src[P::kOff2 + ((q + rot) % n2) * 256 + t256];
#pragma unroll
            for (int q = 0; q < n3; ++q) i3[q] = src[P::kOff3 + ((q + rot) % n3) * 256 + t256];
#pragma unroll
            for (int q = 0; q < n4; ++q) i4[q] = src[P::kOff4 + ((q + rot) % n4) * 256 + t256];
            auto put = [&](int i, const f32x4p& v) { dst[i ^ (((i >> 5) & 3) << 2)] = v; };   // chunk swizzle: conflict-free transposed reads
#pragma unroll
            for (int q = 0; q < n1; ++q) put(((q + rot) % n1) * 256 + t256, i1[q]);
            __syncthreads();   // [A] W1 staged
            if (stop_flag) return;   // uniform over the grid (the chain waves leave at the same barrier)
#pragma unroll
            for (int q = 0; q < n2; ++q) put(P::kOff2 + ((q + rot) % n2) * 256 + t256, i2[q]);
            __syncthreads();   // [B] W2 staged
#pragma unroll
            for (int q = 0; q < n3; ++q) put(P::kOff3 + ((q + rot) % n3) * 256 + t256, i3[q]);
            __syncthreads();   // [C] W3 staged
#pragma unroll
            for (int q = 0; q < n4; ++q) put(P::kOff4 + ((q + rot) % n4) * 256 + t256, i4[q]);
        }
        PPO_TICK(a, 1);   // (profiling build) image staged: compare with the chain waves' release from [S0]
        // weight-gradient accumulators, live across the passes (initialised AFTER the staging burst: its 80 registers are free again)
        f32x16p dw4 = zero, dw3[2][2] = {{zero, zero}, {zero, zero}}, dw2[2][2] = {{zero, zero}, {zero, zero}}, dw1[2][2] = {{zero, zero}, {zero, zero}};
    for (int pair = blockIdx.x, pass = 0; pair < pairs; pair += gridDim.x, ++pass) {
        // Everything derived from the lane index is formed INSIDE the pass loop, behind an opaque copy: as loop invariants the image
        // addresses, identity operands and lane constants were hoisted in front of the loop and kept alive through the whole kernel
        // (89 spilled registers and a scratch segment, i.e. the runtime's scratch set-up on every launch).
        int lane = (int)(threadIdx.x & 63), tid = (int)threadIdx.x;
        asm volatile("" : "+v"(lane), "+v"(tid));
        const unsigned ex_lane0 = lds_base + 16u * (unsigned)D::kImage + ex_lane_const(lane, 0);   // exchange area + lane
        const unsigned ex_lane1 = lds_base + 16u * (unsigned)D::kImage + ex_lane_const(lane, 1);   // constants (two reads)
            // ============================================= dW waves ============================================================
            __syncthreads();   // [S0]
            if (stop_flag) return;
            // The weight-gradient accumulators of ALL four layers live in this wave's registers across the passes of a large
            // minibatch (16 + 64 + 64 + 16 kIT VGPRs): a later pass accumulates on the matrix core instead of reading its partial
            // back and adding (round 2: a read-modify-write of 126 KB per workgroup and pass), and the partial leaves ONCE, in the
            // last pass: each layer as soon as it is complete, in accumulator order (16 stores of one contiguous run per lane and
            // hidden layer -- as 64 row-segment stores per lane they queued for ~9 k cycles per layer and had to be deferred to the
            // end), written through L2 (store_dw_tile_raw) so that they drain under the remaining layers.
            const bool last = pair + (int)gridDim.x >= pairs;
            const int to0 = 2 * (wave >> 1), ti0 = 2 * (wave & 1);
            __syncthreads();   // [S1]
            PPO_TICK(a, 6);
            dw_tile_old_tr_half<0>(E, ex_lane0 + 2048u * (unsigned)wave, ex_lane1 + 2048u * (unsigned)wave, lane, dw4);   // tile (0, wave)
            dw_tile_old_tr_half<4>(E, ex_lane0 + 2048u * (unsigned)wave, ex_lane1 + 2048u * (unsigned)wave, lane, dw4);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();   // [S2] operands consumed (the MFMAs have them in registers): the chain may overwrite them
            PPO_TICK(a, 7);
            __syncthreads();   // [S3]
            PPO_TICK(a, 8);
            dw_block_tr(ex_lane0 + 2048u * (unsigned)to0, ex_lane1 + 2048u * (unsigned)to0, ex_lane0 + 2048u * (unsigned)ti0, ex_lane1 + 2048u * (unsigned)ti0, dw3);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();   // [S4]
            if (last) store_dw_tile_raw<PT>(dw4, gn + (D::kRawT4 + wave) * 1024, O, lane, scale);   // 1 tile: cheap
            if (last) {   // dW3 is complete: its partial drains while dW2 / dW1 are formed
#pragma unroll
                for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                    for (int bi = 0; bi < 2; ++bi) store_dw_tile_raw<PT, true>(dw3[bt][bi], gn + (D::kRawT3 + 4 * (to0 + bt) + ti0 + bi) * 1024, kH - 32 * (to0 + bt), lane, scale);
            }
            PPO_TICK(a, 9);
            __syncthreads();   // [S5]
            PPO_TICK(a, 10);
            dw_block_tr(ex_lane0 + 2048u * (unsigned)to0, ex_lane1 + 2048u * (unsigned)to0, ex_lane0 + 2048u * (unsigned)ti0, ex_lane1 + 2048u * (unsigned)ti0, dw2);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();   // [S6]
            PPO_TICK(a, 11);
            __syncthreads();   // [S7] (arrive BEFORE the layer-2 stores: the chain waves are released to their next gather at once)
            PPO_TICK(a, 12);
            if (last) {   // likewise dW2 (the chain waves are past their last barrier of this pass)
#pragma unroll
                for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                    for (int bi = 0; bi < 2; ++bi) store_dw_tile_raw<PT, true>(dw2[bt][bi], gn + (D::kRawT2 + 4 * (to0 + bt) + ti0 + bi) * 1024, kH - 32 * (to0 + bt), lane, scale);
            }
            dw_tiles_tr_old_half<D::kIT, 0>(ex_lane0 + 2048u * (unsigned)wave, ex_lane1 + 2048u * (unsigned)wave, E + 4 * 8 * 64, lane, dw1);   // tiles (wave, 0..kIT-1)
            dw_tiles_tr_old_half<D::kIT, 4>(ex_lane0 + 2048u * (unsigned)wave, ex_lane1 + 2048u * (unsigned)wave, E + 4 * 8 * 64, lane, dw1);
            __builtin_amdgcn_sched_barrier(0);
            if (last) {
#pragma unroll
                for (int bi = 0; bi < D::kIT; ++bi) store_dw_tile_raw<PT, true>(dw1[0][bi], gn + (D::kIT * wave + bi) * 1024, kH - 32 * wave, lane, scale);
            }
            PPO_TICK(a, 13);
#ifdef QR_PHASE_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PPO_WALL(a.ticks, (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + 15);
#endif
            // ([S0] of the next pass separates these reads from the chain waves' next writes)
        }
    }
}

#if defined(QR_PHASE_TIMING) && defined(QR_TICK_PASS)
#undef QR_TICK_GATE
#define QR_TICK_GATE true
#endif
// ---- gradient reduction, global norm, clip, Adam, operand re-pack: ONE kernel ----------------------------------------------
struct ApplyArgs {
    float *theta, *m, *v;     // parameters and Adam moments (flat, n floats)
    const float* ext_grad;    // data-parallel path: [n + 4] externally averaged gradient + minibatch statistics; else nullptr
    float* grad_out;          // optional [n + 4]: the reduced gradient + minibatch statistics (qr_ppo_grad); else nullptr
    const float* partial;     // [chunks][slots] the gradient kernel's per-workgroup partials (or [n] for an external gradient)
    int partial_bf16;         // 1: the partials are __bf16 (same indexing, half the bytes)
    int partial_raw;          // > 0: accumulator-order partials of ppo_grad_kernel, [chunks][partial_raw] with partial_raw = 2 x
                              // PpoDims::kRawSlots; thread -> (slot, parameter) by DenseMap
    int owner_from;           // first thread index of the log_std entries (the workgroups from there on also sum the per-wave sums)
    int chunks, n;
    const float* wave_out;    // per-wave sums of the gradient kernel's chain waves, Gw = waves per net
    int Gw;
    float ent_coef;
    float* stats;             // optional [4], accumulated
    PpoCtrl* ctrl;
    half8* images;            // [2][kImage] operand images (re-packed after the step)
    float max_norm, lr, beta1, beta2, eps, bc1, bc2_sqrt;
    float kl_limit;           // 1.5 * target_kl * B (threshold on the minibatch's KL SUM); <= 0: no early stop
    int take_step;            // 0: reduce only (qr_ppo_grad)
    int device_step;          // 1: bias corrections from PpoCtrl::adam_t + 1 (computed here) instead of bc1 / bc2_sqrt
    int device_lr;            // 1: learning rate = PpoCtrl::lr
#ifdef QR_PHASE_TIMING
    unsigned long long* aticks;   // [workgroups][8] wall-clock stamps (profiling build only)
#endif
};

// gradient element i and, in the block(s) that own the log_std entries, the per-wave sums of the chain waves:
// red[0..3] = d loss / d log_std[k] (x 1/B), red[4..7] = sum surrogate, sum squared value error, sum approx kl, clipped count
constexpr int kApplyThreads = 256;   // 247 workgroups: the 8 MB of chunk partials are pulled by (almost) every CU
                                     // (62 x 1024 threads took 13.5 us for this kernel, bound by 62 CUs' load issue)

// i = the thread's parameter (n: none), pj = its column in the partial table (natural order: i; accumulator order: the slot),
// pair = accumulator-order launch and this thread is one of the slot threads (see the pair scheme below)
__device__ __forceinline__ float reduce_grad_element(const ApplyArgs& a, int i, int pj, bool pair, float* red /* shared [8] */) {
    const int n = a.n;
    if (a.ext_grad) {
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 4) red[4 + threadIdx.x] = a.ext_grad[n + threadIdx.x];
        __syncthreads();
        return i < n ? a.ext_grad[i] : 0.0f;
    }
    // The block(s) holding the log_std entries (the last one or two) also sum the chain waves' per-wave sums.  All of their loads are
    // issued up front, together with the chunk partials below: one memory round trip for the whole prologue (a wave that walked
    // its 512 rows in a loop paid eight of them, and the grid barrier waits for exactly these blocks).
    //   policy waves (net 0): slots 0..3 d log_std / B, 4 surrogate, 5 approx kl, 6 clipped;  value waves (net 1): slot 4 squared error
    // (accumulator-order launch: a.owner_from = 2 x DenseMap::kDenseThreads, the first log_std thread; else n - 4)
    const bool owner = ((int)blockIdx.x + 1) * kApplyThreads > a.owner_from;
    constexpr int kRowsPerThread = 4;   // up to 1024 waves per net (a 32 768-row minibatch); larger ones take the loop below
    float4 lo[kRowsPerThread], hi[kRowsPerThread];
    float vs[kRowsPerThread];
    if (owner) {
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) {
            const int q = r * kApplyThreads + (int)threadIdx.x;
            const int qc = q < a.Gw ? q : 0;
            lo[r] = *reinterpret_cast<const float4*>(a.wave_out + (size_t)qc * 8);
            hi[r] = *reinterpret_cast<const float4*>(a.wave_out + (size_t)qc * 8 + 4);
            vs[r] = a.wave_out[((size_t)a.Gw + qc) * 8 + 4];
        }
    }
    float g = 0.0f;
    // ALL chunk partials of this element are requested before the first add (up to 128 x 4 bytes per thread in flight: the kernel
    // has the registers, one wave per SIMD).  Four batches of 32 were four dependent round trips through the fabric for data the
    // previous kernel had just written (32 MB per minibatch: this read is most of the kernel's time).
    constexpr int kMaxChunks = 128;
    float gs[kMaxChunks];
    // Accumulator-order partials: the threads of slots (k, k + 1) of one lane's quad -- neighbours t, t ^ 1 -- share the work the other
    // way round: BOTH read the pair of slots (one dword of two bf16, or a dwordx2), thread t & 1 = 0 for chunks 0..63 and thread
    // t & 1 = 1 for chunks 64..127.  64 loads per thread is what a wave can have in flight (vmcnt), so the whole table (16 MB) is
    // requested in ONE round trip; 128 two-byte loads per thread went out in two rounds (measured: 5.4 -> 3 us for this phase).
    // Each thread sums its 64 chunks of both slots (fixed tree), then the two exchange the half they do not own.
    if (pair) {
        const size_t pitch = (size_t)a.partial_raw;
        const int qh = pj & 1, pb2 = pj & ~1;
        if (a.partial_bf16) {
            const unsigned int* pw = reinterpret_cast<const unsigned int*>(a.partial);   // two bf16 per dword
#pragma unroll
            for (int q = 0; q < kMaxChunks / 2; ++q) {
                const int qq = 64 * qh + q;
                const unsigned int w = pw[((size_t)(qq < a.chunks ? qq : 0) * pitch + pb2) >> 1];
                gs[q] = __uint_as_float(w << 16);
                gs[q + 64] = __uint_as_float(w & 0xFFFF0000u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < kMaxChunks / 2; ++q) {
                const int qq = 64 * qh + q;
                const float2 w = *reinterpret_cast<const float2*>(a.partial + (size_t)(qq < a.chunks ? qq : 0) * pitch + pb2);
                gs[q] = w.x;
                gs[q + 64] = w.y;
            }
        }
    } else if (i < n - 4) {  // natural order (split / 4-wave forms): the partials of one parameter, one per sample chunk
        const size_t pitch = (size_t)n;
#pragma unroll
        for (int q = 0; q < kMaxChunks; ++q) gs[q] = a.partial[(size_t)(q < a.chunks ? q : 0) * pitch + pj];  // unconditional loads
    }
    if (owner) {
        float t[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < kRowsPerThread; ++r) {
            const bool on = r * kApplyThreads + (int)threadIdx.x < a.Gw;
            t[0] += on ? lo[r].x : 0.0f; t[1] += on ? lo[r].y : 0.0f; t[2] += on ? lo[r].z : 0.0f; t[3] += on ? lo[r].w : 0.0f;
            t[4] += on ? hi[r].x : 0.0f; t[6] += on ? hi[r].y : 0.0f; t[7] += on ? hi[r].z : 0.0f;
            t[5] += on ? vs[r] : 0.0f;
        }
        for (int q = kRowsPerThread * kApplyThreads + (int)threadIdx.x; q < a.Gw; q += kApplyThreads) {
            const float4 l4 = *reinterpret_cast<const float4*>(a.wave_out + (size_t)q * 8);
            const float4 h4 = *reinterpret_cast<const float4*>(a.wave_out + (size_t)q * 8 + 4);
            t[0] += l4.x; t[1] += l4.y; t[2] += l4.z; t[3] += l4.w;
            t[4] += h4.x; t[6] += h4.y; t[7] += h4.z;
            t[5] += a.wave_out[((size_t)a.Gw + q) * 8 + 4];
        }
        __shared__ float part[8][kApplyThreads / 64];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float w = wave_sum(t[k]);
            if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = w;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            float w = 0.0f;
#pragma unroll
            for (int q = 0; q < kApplyThreads / 64; ++q) w += part[threadIdx.x][q];
            red[threadIdx.x] = w;
        }
        __syncthreads();
    }
    if (pair) {
        // gs[0..63] = slot k even, gs[64..127] = slot k + 1, this thread's 64 chunks each: two fixed-shape trees (chunks past the
        // count hold zeros), then lower half + upper half -- the order depends on nothing but the chunk index: bitwise reproducible
        const int qh = pj & 1;
#pragma unroll
        for (int q = 0; q < kMaxChunks / 2; ++q) {
            const bool on = 64 * qh + q < a.chunks;
            gs[q] = on ? gs[q] : 0.0f;
            gs[q + 64] = on ? gs[q + 64] : 0.0f;
        }
#pragma unroll
        for (int w = kMaxChunks / 4; w >= 1; w >>= 1)
#pragma unroll
            for (int q = 0; q < w; ++q) { gs[q] += gs[q + w]; gs[64 + q] += gs[64 + q + w]; }
        const float send = qh == 0 ? gs[64] : gs[0];          // the half sum of the slot the neighbour owns
        const float recv = __shfl_xor(send, 1);
        g = qh == 0 ? gs[0] + recv : recv + gs[64];           // chunks 0..63 + chunks 64..127
        if (i >= n - 4) g = 0.0f;                             // (the unused rows of the value net's output quad)
    } else if (i < n - 4) {
        // fixed-shape tree over the 128 slots (slots past the chunk count hold zeros): the summation order depends on nothing but
        // the chunk index -> bitwise reproducible
#pragma unroll
        for (int q = 0; q < kMaxChunks; ++q) gs[q] = q < a.chunks ? gs[q] : 0.0f;
#pragma unroll
        for (int w = kMaxChunks / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int q = 0; q < w; ++q) gs[q] += gs[q + w];
        g = gs[0];
    } else if (i < n) {  // log_std; entropy = sum(log_std) + const
        g = red[i - (n - 4)] - a.ent_coef;
    }
    return g;
}

// f16 operand-image positions of parameter `local` (index inside one net's block of the flat vector): the inverse of
// ppo_pack_kernel's gather.  Every weight sits once in the image.
template <int L>
__device__ __forceinline__ void pack_scatter(_Float16* __restrict__ img, int O, int local, float val) {
    using P = PolicyDims<L>;
    const NetOff o = net_off(L, O);
    const _Float16 hv = (_Float16)val;
    // (row, hidden k-slot) of a 128-wide layer image starting at half8 offset `off`: see ppo_pack_kernel
    auto hidden_pos = [](int off, int row, int hid, bool single_tile) -> size_t {
        const int t = single_tile ? 0 : (row >> 5), c = row & 31, q = hid >> 5, u = hid & 31;
        const int h = (u >> 2) & 1, r = (u & 3) + 4 * (u >> 3);      // u = rho(r, h)
        const int sp = 2 * q + (r >> 3), j = r & 7;
        return ((size_t)(off + t * 512 + sp * 64 + 32 * h + c)) * 8 + j;
    };
    auto layer1_pos = [](int row, int k) -> size_t {
        const int t = row >> 5, c = row & 31, sidx = k >> 4, h = (k >> 3) & 1, j = k & 7;
        return ((size_t)((t * P::kSteps1 + sidx) * 64 + 32 * h + c)) * 8 + j;
    };
    if (local < o.b1) {                       // w1[row][k]
        img[layer1_pos(local / L, local % L)] = hv;
    } else if (local < o.w2) {                // b1[row] = input L (the constant 1)
        img[layer1_pos(local - o.b1, L)] = hv;
    } else if (local < o.b2) {                // w2[out][in]
        const int out = (local - o.w2) / kH, in = (local - o.w2) % kH;
        img[hidden_pos(P::kOff2, out, in, false)] = hv;
    } else if (local < o.w3) {
        img[hidden_pos(P::kOff2, local - o.b2, kPolBiasUnit, false)] = hv;
    } else if (local < o.b3) {
        const int out = (local - o.w3) / kH, in = (local - o.w3) % kH;
        img[hidden_pos(P::kOff3, out, in, false)] = hv;
    } else if (local < o.w4) {
        img[hidden_pos(P::kOff3, local - o.b3, kPolBiasUnit, false)] = hv;
    } else if (local < o.b4) {                // w4[oo][i]: output tile rows 0..O-1
        const int oo = (local - o.w4) / kH, i = (local - o.w4) % kH;
        img[hidden_pos(P::kOff4, oo, i, true)] = hv;
    } else {                                  // b4[oo]
        img[hidden_pos(P::kOff4, local - o.b4, kPolBiasUnit, true)] = hv;
    }
}

template <int L>
__global__ void __launch_bounds__(kApplyThreads, 2) ppo_apply_kernel(ApplyArgs a) {   // <= 256 VGPRs: two workgroups per CU are resident
    PpoCtrl* c = a.ctrl;
    const int stop_flag = c->stop;       // set by an EARLIER launch: uniform over the grid; tested after the reduction below
    const unsigned int gen = c->gen;
    const int adam_t = c->adam_t;        // like `gen`: read by everybody before anybody arrives at the barrier, bumped by the master after it
    const float lr_dev = c->lr;
    const int n = a.n;
#ifdef QR_PHASE_TIMING
    PPO_WALL(a.aticks, (size_t)blockIdx.x * 8 + 0);
#endif
    const int j = blockIdx.x * kApplyThreads + threadIdx.x;
    // the thread's parameter: natural order -> j itself; accumulator-order partials -> DenseMap (the unused rows of the value net's
    // output quad and the tail of the last workgroup own nothing: i = n), the four log_std entries ride behind the two nets
    int i = j;
    int pj = j;
    bool slot_thread = false;
    if (a.partial_raw > 0 && !a.ext_grad) {
        using M = DenseMap<L>;
        constexpr int S = PpoDims<L>::kRawSlots, T = M::kDenseThreads;
        const int n4 = net_off(L, 4).total;
        int p = -1;
        pj = 0;
        if (j < T) { M::map(j, 4, pj, p); i = p < 0 ? n : p; }
        else if (j < 2 * T) { M::map(j - T, 1, pj, p); pj += S; i = p < 0 ? n : n4 + p; }
        else i = j - 2 * T < 4 ? n - 4 + (j - 2 * T) : n;
        slot_thread = j < 2 * T;
    }
    const bool last_block = blockIdx.x == gridDim.x - 1;
    __shared__ float red[8];
    // Adam state and parameter of this element: loaded NOW, together with the chunk partials, so that their round trip is over
    // before the grid barrier releases (they do not depend on the norm): measured -0.15 us
    const bool owns = a.take_step && i < n;
    const float m_in = owns ? a.m[i] : 0.0f, v_in = owns ? a.v[i] : 0.0f, th_in = owns ? a.theta[i] : 0.0f;
    const float g = reduce_grad_element(a, i, pj, slot_thread, red);
#ifdef QR_PHASE_TIMING
    PPO_WALL(a.aticks, (size_t)blockIdx.x * 8 + 1);
#endif
    if (a.grad_out) {
        if (i < n) a.grad_out[i] = g;
        if (last_block && threadIdx.x < 4) a.grad_out[n + threadIdx.x] = red[4 + threadIdx.x];
    }
    if (!a.take_step) {
        if (last_block && threadIdx.x < 4 && a.stats) a.stats[threadIdx.x] += red[4 + threadIdx.x];
        return;
    }
    if (stop_flag) return;  // SB3: no update after the early stop
    // ---- squared norm across the grid: every workgroup of this launch is resident (247 x 256 threads on 256 CUs).
    // Arrive: one 64-bit store per workgroup = generation | f32 sum.  The LAST workgroup is the master (it owns the minibatch
    // statistics, so it can take the target-KL decision itself): its first wave polls all arrival words (one coalesced load per 64
    // workgroups), adds the sums in a fixed order and releases `go` = generation | verdict bits | f32 total.  Everybody else
    // polls that one word.  RELAXED polling (an agent-scope ACQUIRE load invalidates the XCD's L2 on every iteration and slowed
    // the workgroups still summing partials); nothing but the polled word itself is consumed, so no fence is needed.  Bounded.
    double sq = (double)g * g, unused = 0.0;
    block_sum2_f64<kApplyThreads>(sq, unused);
#ifdef QR_PHASE_TIMING
    PPO_WALL(a.aticks, (size_t)blockIdx.x * 8 + 2);
#endif
    const unsigned int want = (gen + 1u) & kGoGenMask;
    if (threadIdx.x == 0) {
        const unsigned long long word = ((unsigned long long)want << 32) | (unsigned long long)__float_as_uint((float)sq);
        __hip_atomic_store(&c->arrive[blockIdx.x], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last_block && threadIdx.x < 64) {
        bool all = false;
        double total = 0.0;
        for (int spins = 0; !all && spins < (1 << 20); ++spins) {
            bool mine = true;
            total = 0.0;
            for (int q = threadIdx.x; q < (int)gridDim.x; q += 64) {
                const unsigned long long w = __hip_atomic_load(&c->arrive[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine = mine && (unsigned int)(w >> 32) == want;
                total += (double)__uint_as_float((unsigned int)w);
            }
            all = __ballot(mine) == ~0ull;
            if (!all) __builtin_amdgcn_s_sleep(1);
        }
        total = wave_sum_f64(total);   // lanes in a fixed order: one value, whatever the arrival order was
        if (threadIdx.x == 0) {
            if (!all) atomicAdd(&c->barrier_timeouts, 1);
            const float nsq = (float)total;
            const bool stop_m = a.kl_limit > 0.0f && red[6] > a.kl_limit;   // SB3: checked BEFORE the optimiser step of this minibatch
            const bool finite_m = nsq <= 3.0e38f;                            // false for inf and NaN
            c->gen = gen + 1u;   // every workgroup has read `gen` before it arrived; visible to the next launch
            // a barrier that timed out (never observed) releases with the non-finite verdict: NOBODY steps on a norm that lacks
            // some workgroups' sums (the launch is counted in barrier_timeouts and in skipped_nonfinite)
            const unsigned int hi = want | (stop_m ? kGoStop : 0u) | ((finite_m && all) ? 0u : kGoNonFinite);
            __hip_atomic_store(&c->go, ((unsigned long long)hi << 32) | (unsigned long long)__float_as_uint(nsq), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __shared__ unsigned long long go_s;
    if (threadIdx.x == 0) {
        unsigned long long w = 0;
        int spins = 0;
        while ((((unsigned int)((w = __hip_atomic_load(&c->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32)) & kGoGenMask) != want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 20)) {  // never observed
                atomicAdd(&c->barrier_timeouts, 1);
                w = ((unsigned long long)(want | kGoNonFinite) << 32);   // take no step on a broken barrier
                break;
            }
        }
        go_s = w;
    }
    __syncthreads();
#ifdef QR_PHASE_TIMING
    PPO_WALL(a.aticks, (size_t)blockIdx.x * 8 + 3);
#endif
    const unsigned long long gw = go_s;
    const float norm_sq = __uint_as_float((unsigned int)gw);
    const bool stop_now = ((unsigned int)(gw >> 32) & kGoStop) != 0u;
    const bool finite = ((unsigned int)(gw >> 32) & kGoNonFinite) == 0u;
    if (!stop_now && finite && i < n) {
        const float norm = sqrtf(norm_sq);
        const float clip = fminf(1.0f, a.max_norm / (norm + 1e-6f));  // torch.nn.utils.clip_grad_norm_
        const float gc = g * clip;
        const float mi = a.beta1 * m_in + (1.0f - a.beta1) * gc;
        const float vi = a.beta2 * v_in + (1.0f - a.beta2) * gc * gc;
        a.m[i] = mi;
        a.v[i] = vi;
        // bias corrections: host-computed for a caller-counted step, or from the device's own count of steps really taken (the
        // same float expressions as adam_constants())
        const float bc1 = a.device_step ? 1.0f - powf(a.beta1, (float)(adam_t + 1)) : a.bc1;
        const float bc2_sqrt = a.device_step ? sqrtf(1.0f - powf(a.beta2, (float)(adam_t + 1))) : a.bc2_sqrt;
        const float lr = a.device_lr ? lr_dev : a.lr;
        const float th = th_in - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + a.eps);  // torch.optim.Adam
        a.theta[i] = th;
        // operand images of the next minibatch: this parameter's f16 copies (log_std has none)
        const int n4 = net_off(L, 4).total, n1 = net_off(L, 1).total;
        if (i < n4) pack_scatter<L>(reinterpret_cast<_Float16*>(a.images), 4, i, th);
        else if (i < n4 + n1) pack_scatter<L>(reinterpret_cast<_Float16*>(a.images + PpoDims<L>::kImage), 1, i - n4, th);
    }
    if (last_block && threadIdx.x == 0) {
        if (a.stats)
            for (int k = 0; k < 4; ++k) a.stats[k] += red[4 + k];
        if (stop_now) c->stop = 1;
        else if (finite) { c->applied += 1; c->adam_t = adam_t + 1; }
        else c->skipped_nonfinite += 1;
    }
#ifdef QR_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PPO_WALL(a.aticks, (size_t)blockIdx.x * 8 + 4);
#endif
}

// ---- GAE(lambda) and episode statistics over a rollout buffer [T][N] (what SB3's RolloutBuffer.compute_returns_and_advantage
// and VecMonitor do on the host): one lane = one env.  done is 0 / 1 as float.  term_val (optional) = V(terminal observation)
// at the steps that ended by the time limit, 0 elsewhere: SB3's collect_rollouts adds gamma * V(terminal_obs) to the reward
// of a truncated step before the buffer sees it (the episode still ends there: no bootstrap through the reset).
__global__ void __launch_bounds__(256) ppo_gae_kernel(int T, int N, const float* __restrict__ rew, const float* __restrict__ done,
                                                      const float* __restrict__ val, const float* __restrict__ last_val,
                                                      const float* __restrict__ term_val, float gamma,
                                                      float lam, float* __restrict__ adv, float* __restrict__ ret,
                                                      float* __restrict__ ep_ret, float* __restrict__ ep_len,
                                                      float* __restrict__ ep_gates, float* __restrict__ fin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float f0 = 0.0f, f1 = 0.0f, f2 = 0.0f, f3 = 0.0f;
    if (i < N) {
        float last = 0.0f, next_val = last_val[i];
        for (int t = T - 1; t >= 0; --t) {
            const size_t e = (size_t)t * N + i;
            const float nonterminal = 1.0f - done[e], v = val[e];
            const float r = term_val ? fmaf(gamma, term_val[e], rew[e]) : rew[e];
            const float delta = r + gamma * next_val * nonterminal - v;
            last = delta + gamma * lam * nonterminal * last;
            adv[e] = last;
            ret[e] = last + v;
            next_val = v;
        }
        if (ep_ret) {
            float er = ep_ret[i], el = ep_len[i], eg = ep_gates[i];
            for (int t = 0; t < T; ++t) {
                const size_t e = (size_t)t * N + i;
                const float r = rew[e], d = done[e];
                er += r;
                el += 1.0f;
                eg += r > 5.0f ? 1.0f : 0.0f;  // gate reward 10 - 10 * d2g (R:537)
                f0 += er * d; f1 += el * d; f2 += eg * d; f3 += d;
                const float keep = 1.0f - d;
                er *= keep; el *= keep; eg *= keep;
            }
            ep_ret[i] = er; ep_len[i] = el; ep_gates[i] = eg;
        }
    }
    if (fin) {  // finished-episode sums: wave reduction, one atomic per wave and statistic
        f0 = wave_sum(f0); f1 = wave_sum(f1); f2 = wave_sum(f2); f3 = wave_sum(f3);
        if ((threadIdx.x & 63) == 0 && f3 > 0.0f) {
            unsafeAtomicAdd(fin + 0, f0); unsafeAtomicAdd(fin + 1, f1); unsafeAtomicAdd(fin + 2, f2); unsafeAtomicAdd(fin + 3, f3);
        }
    }
}

}  // namespace qr

// ------------------------------------------------------------------------------------------------------------------------
struct qr_ppo {
    int L = 0, device = 0, max_B = 0, num_params = 0;
    int image_half8 = 0, raw_slots = 0;
    static constexpr int kFusedChunks = 128;          // workgroups (= partials) per network of the gradient kernel
    bool partial_bf16 = true;                         // QR_PPO_PARTIAL_F32: the role-split kernel's partials as f32 (twice the bytes)
    bool epoch_graph = true;                          // QR_PPO_NO_EPOCH_GRAPH: qr_ppo_epoch enqueues its launches on the stream
    qr::half8* d_images = nullptr;
    float* d_partial = nullptr;  // [kFusedChunks][raw slots] partial weight gradients of the gradient kernel's workgroups
    float* d_wave = nullptr;     // [2][2 x max groups][8] per-wave sums of the chain waves
#ifdef QR_PHASE_TIMING
    unsigned long long* ticks = nullptr;
    unsigned long long* apply_ticks = nullptr;
#endif
    qr::PpoCtrl* d_ctrl = nullptr;
    // advantage sums per minibatch: entries 0..kMaxEpochMinibatches-1 are filled for a whole epoch by qr_ppo_epoch_begin,
    // the last entry serves a minibatch that was not announced that way
    static constexpr int kMaxEpochMinibatches = 4096;
    double* d_mbstats = nullptr;   // [kMaxEpochMinibatches + 1][2]
    const int32_t* epoch_idx = nullptr;
    int epoch_B = 0, epoch_count = 0, epoch_cursor = 0;
    float target_kl = 0.0f;        // <= 0: no early stop
    const float* packed_theta = nullptr;   // parameter vector the operand images were last built from (pack / apply)
    // qr_ppo_epoch: one epoch's launches (advantage statistics + num_minibatches x (gradient, apply)) captured once into a hipGraph
    // and replayed while the arguments stay the same -- everything that changes between epochs lives in device memory (the
    // permutation's CONTENT, the Adam step count, the learning rate, the stop flag)
    struct EpochGraph {
        hipGraphExec_t exec = nullptr;
        const void *theta = nullptr, *m = nullptr, *v = nullptr, *obs = nullptr, *act = nullptr, *old_logp = nullptr, *adv = nullptr,
                   *ret = nullptr, *perm = nullptr, *stats = nullptr;
        int B = 0, M = 0, E = 0, shuffle = 0;
        unsigned long long seed = 0;
        float clip = 0, vf_coef = 0, ent_coef = 0, max_grad_norm = 0, beta1 = 0, beta2 = 0, eps = 0, target_kl = 0;
    } eg;
    hipStream_t capture_stream = nullptr;
    void* f32_scratch = nullptr;           // activations / deltas of the f32-class gradient path (quadrace_ppo_f32.hip), allocated on first use
    size_t f32_scratch_bytes = 0;
    void* f32_graphs = nullptr;            // its cached graphs (one per distinct argument set), released by qr::ppo_f32_release_graphs
    unsigned long long shuffle_seed = 0;   // key of the on-device epoch permutations (qr_ppo_shuffle_state)
};

namespace qr {
int set_last_error(int code, const std::string& msg);  // quadrace_abi.hip
hipError_t launch_policy(int L, const half8* w, int n, const float* obs, float* mean, hipStream_t st);  // quadrace_policy.hip
const half8* ppo_policy_image(const qr_ppo* p) { return p ? p->d_images : nullptr; }
// what quadrace_ppo_f32.hip needs of a handle
int ppo_handle_info(const qr_ppo* p, int* L, int* device, int* max_B, int* num_params) {
    if (!p) return QR_E_INVALID;
    *L = p->L; *device = p->device; *max_B = p->max_B; *num_params = p->num_params;
    return QR_OK;
}
void** ppo_f32_scratch_slot(qr_ppo* p) { return &p->f32_scratch; }
size_t* ppo_f32_scratch_bytes(qr_ppo* p) { return &p->f32_scratch_bytes; }
void** ppo_f32_graphs_slot(qr_ppo* p) { return &p->f32_graphs; }
hipStream_t* ppo_capture_stream_slot(qr_ppo* p) { return &p->capture_stream; }
bool ppo_uses_graphs(const qr_ppo* p) { return p->epoch_graph; }
void ppo_f32_release_graphs(void* cache);   // quadrace_ppo_f32.hip
}  // namespace qr

namespace {

int ppofail(int code, const std::string& m) { return qr::set_last_error(code, m); }

#define PPO_HIP(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return ppofail(QR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

template <int L>
struct PpoOps {
    using D = qr::PpoDims<L>;
    static int pack(qr_ppo* p, const float* theta, hipStream_t st) {
        hipLaunchKernelGGL(qr::ppo_pack_kernel<L>, dim3((D::kImage + 255) / 256, 2), dim3(256), 0, st, theta, p->d_images);
        PPO_HIP(hipGetLastError());
        return QR_OK;
    }
    // advantage statistics of this minibatch: the epoch table when the rows were announced by qr_ppo_epoch_begin, else one launch
    static int adv_stats(qr_ppo* p, qr::PpoBatch& b, hipStream_t st) {
        // the table is consumed strictly in order (entry k serves the call whose rows are idx_all + k * B); anything else
        // disarms it, so a recycled buffer address can never pick up the sums of an older permutation
        if (p->epoch_idx && b.B == p->epoch_B && p->epoch_cursor < p->epoch_count &&
            b.idx == p->epoch_idx + (size_t)p->epoch_cursor * b.B) {
            b.acc = p->d_mbstats + 2 * p->epoch_cursor;
            if (++p->epoch_cursor == p->epoch_count) p->epoch_idx = nullptr;
            return QR_OK;
        }
        p->epoch_idx = nullptr;
        double* slot = p->d_mbstats + 2 * qr_ppo::kMaxEpochMinibatches;
        hipLaunchKernelGGL(qr::ppo_zero_table_kernel, dim3(1), dim3(256), 0, st, slot, 2);
        hipLaunchKernelGGL(qr::ppo_adv_stats_kernel, dim3((b.B + 1023) / 1024, 1), dim3(1024), 0, st, b.adv, b.idx, b.B, slot,
                           (unsigned long long*)nullptr);
        b.acc = slot;
        return QR_OK;
    }
    static constexpr size_t kLdsFused = ((size_t)D::kImage + qr::kExHalf8) * 16 + 7 * qr::kStashRows * sizeof(float);
    // dynamic-LDS limit of the gradient kernel on the CURRENT device (idempotent; not a stream operation, so it also runs
    // before a graph capture instead of inside it)
    static int configure(qr_ppo* p) {
        static unsigned long long configured_f = 0, configured_fb = 0;   // per device ordinal
        if (p->partial_bf16) {
            PPO_HIP(qr::ensure_dynamic_lds(reinterpret_cast<const void*>(qr::ppo_grad_kernel<L, __bf16>), kLdsFused, configured_fb));
        } else {
            PPO_HIP(qr::ensure_dynamic_lds(reinterpret_cast<const void*>(qr::ppo_grad_kernel<L, float>), kLdsFused, configured_f));
        }
        return QR_OK;
    }
    // one kernel: forward, backward and the weight gradients of 128 samples per workgroup and pass; partial gradients of every workgroup
    // in d_partial, per-wave sums in d_wave; returns the number of partials per network
    static int grad(qr_ppo* p, qr::PpoBatch b, hipStream_t st, int* chunks_out) {
        constexpr size_t lds_f = kLdsFused;
        if (int rc = configure(p)) return rc;
        if (int rc = adv_stats(p, b, st)) return rc;
        const int pairs = (b.G + 1) / 2;
        const int wgs = pairs < qr_ppo::kFusedChunks ? pairs : qr_ppo::kFusedChunks;
        if (p->partial_bf16)
            hipLaunchKernelGGL((qr::ppo_grad_kernel<L, __bf16>), dim3(wgs, 2), dim3(512), lds_f, st, b,
                               reinterpret_cast<__bf16*>(p->d_partial), p->num_params);
        else
            hipLaunchKernelGGL((qr::ppo_grad_kernel<L, float>), dim3(wgs, 2), dim3(512), lds_f, st, b, p->d_partial, p->num_params);
        PPO_HIP(hipGetLastError());
        *chunks_out = wgs;
        return QR_OK;
    }
    static int apply(qr_ppo* p, qr::ApplyArgs a, hipStream_t st) {
        a.ctrl = p->d_ctrl;
        a.images = p->d_images;
        a.n = p->num_params;
        a.partial = p->d_partial;
        // the role-split gradient kernel leaves its partials in accumulator order: one thread per slot (+ one workgroup for log_std)
        const bool raw = !a.ext_grad;
        a.partial_bf16 = (raw && p->partial_bf16) ? 1 : 0;
        a.partial_raw = raw ? 2 * D::kRawSlots : 0;
        a.wave_out = p->d_wave;
#ifdef QR_PHASE_TIMING
        a.aticks = p->apply_ticks;
#endif
        const int threads = raw ? 2 * qr::DenseMap<L>::kDenseThreads + 4 : p->num_params;
        a.owner_from = threads - 4;
        static_assert((2 * qr::DenseMap<L>::kDenseThreads + 4 + qr::kApplyThreads - 1) / qr::kApplyThreads <= 512,
                      "grid barrier: arrive[512], two resident workgroups per CU");
        hipLaunchKernelGGL(qr::ppo_apply_kernel<L>, dim3((threads + qr::kApplyThreads - 1) / qr::kApplyThreads),
                           dim3(qr::kApplyThreads), 0, st, a);
        PPO_HIP(hipGetLastError());
        return QR_OK;
    }
};

template <typename F>
int dispatch_L(int L, F&& f) {
    switch (L) {
        case 13: return f(std::integral_constant<int, 13>());
        case 17: return f(std::integral_constant<int, 17>());
        case 21: return f(std::integral_constant<int, 21>());
        case 25: return f(std::integral_constant<int, 25>());
        case 29: return f(std::integral_constant<int, 29>());
        case 20: return f(std::integral_constant<int, 20>());
        case 24: return f(std::integral_constant<int, 24>());
        case 28: return f(std::integral_constant<int, 28>());
        case 32: return f(std::integral_constant<int, 32>());
        case 36: return f(std::integral_constant<int, 36>());
        default: return ppofail(QR_E_INVALID, "obs_len must be an observation length of the race envs");
    }
}

int fill_batch(qr_ppo* p, qr::PpoBatch& b, const float* theta, const float* obs, const float* act, const float* old_logp,
               const float* adv, const float* ret, const int32_t* idx, int32_t B, float clip, float vf_coef, float ent_coef,
               float* stats) {
    if (!p || !theta || !obs || !act || !old_logp || !adv || !ret || !idx) return ppofail(QR_E_INVALID, "qr_ppo: null argument");
    // (the gradient kernel masks the tail of a last, partial group of 64 rows)
    if (B < 64 || B > p->max_B)
        return ppofail(QR_E_INVALID, "qr_ppo: minibatch size must be >= 64 and <= max_minibatch");
    b.obs = obs; b.act = act; b.old_logp = old_logp; b.adv = adv; b.ret = ret; b.idx = idx;
    b.B = B; b.G = (B + 63) / 64;
    b.clip = clip; b.vf_coef = vf_coef; b.ent_coef = ent_coef;
    b.acc = nullptr;
    b.theta = theta;
    b.images = p->d_images;
    b.wave_out = p->d_wave;
    b.stats = stats;
    b.stop = &p->d_ctrl->stop;
#ifdef QR_PHASE_TIMING
    b.ticks = p->ticks;
#endif
    return QR_OK;
}

// adam_step >= 1: the caller counts the steps (bias corrections computed here); adam_step == 0: the device's own count of steps
// really taken (PpoCtrl::adam_t).  device_lr: the learning rate is PpoCtrl::lr (qr_ppo_epoch writes it; library-internal mode --
// the extern "C" entry points take lr >= 0 and reject anything else).
void adam_constants(qr::ApplyArgs& a, float max_grad_norm, float lr, bool device_lr, float beta1, float beta2, float eps, int adam_step) {
    a.max_norm = max_grad_norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.device_step = adam_step == 0;
    a.device_lr = device_lr ? 1 : 0;
    const int t = adam_step > 0 ? adam_step : 1;
    a.bc1 = 1.0f - powf(beta1, (float)t);
    a.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)t));
}

}  // namespace

extern "C" {

int qr_ppo_create(int32_t obs_len, int32_t device, int32_t max_minibatch, qr_ppo** out) {
    return qr_ppo_create_ex(obs_len, device, max_minibatch, 0, out);
}

int qr_ppo_create_ex(int32_t obs_len, int32_t device, int32_t max_minibatch, int32_t flags, qr_ppo** out) {
    if (!out) return ppofail(QR_E_INVALID, "qr_ppo_create: null output");
    if (flags < 0 || (flags & ~(QR_PPO_PARTIAL_F32 | QR_PPO_NO_EPOCH_GRAPH)))
        return ppofail(QR_E_INVALID, "qr_ppo_create_ex: unknown flag");
    *out = nullptr;
    if (max_minibatch < 64 || max_minibatch % 64 != 0) return ppofail(QR_E_INVALID, "qr_ppo_create: max_minibatch must be a multiple of 64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ppofail(QR_E_NO_DEVICE, "qr_ppo_create: no HIP device visible (no CPU fallback)");
    if (device < 0 || device >= ndev) return ppofail(QR_E_INVALID, "qr_ppo_create: bad device ordinal");
    qr_ppo* p = new qr_ppo();
    p->L = obs_len;
    p->device = device;
    p->max_B = max_minibatch;
    const int rc = dispatch_L(obs_len, [&](auto Lc) {
        using D = qr::PpoDims<decltype(Lc)::value>;
        p->image_half8 = D::kImage;
        p->raw_slots = 2 * D::kRawSlots;
        return (int)QR_OK;
    });
    if (rc != QR_OK) { delete p; return rc; }
    p->num_params = qr::ppo_num_params(obs_len);
    // which forms of the kernels this handle uses: explicit flags (the library reads no environment variable)
    p->partial_bf16 = !(flags & QR_PPO_PARTIAL_F32);
    p->epoch_graph = !(flags & QR_PPO_NO_EPOCH_GRAPH);
    PPO_HIP(hipSetDevice(device));
    const size_t mbbytes = (size_t)(qr_ppo::kMaxEpochMinibatches + 1) * 2 * sizeof(double);
    hipError_t e = hipMalloc((void**)&p->d_images, (size_t)2 * p->image_half8 * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_partial, (size_t)qr_ppo::kFusedChunks * (p->raw_slots > p->num_params ? p->raw_slots : p->num_params) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_wave, (size_t)2 * 2 * (max_minibatch / 64) * 8 * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_ctrl, sizeof(qr::PpoCtrl));
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_mbstats, mbbytes);
    if (e == hipSuccess) e = hipMemset(p->d_ctrl, 0, sizeof(qr::PpoCtrl));
    if (e == hipSuccess) e = hipMemset(p->d_mbstats, 0, mbbytes);
    if (e != hipSuccess) {
        qr_ppo_destroy(p);
        return ppofail(QR_E_HIP, std::string("qr_ppo_create: ") + hipGetErrorString(e));
    }
    *out = p;
    return QR_OK;
}

int qr_ppo_destroy(qr_ppo* p) {
    if (!p) return QR_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(p->d_images);
    (void)hipFree(p->d_partial);
    (void)hipFree(p->d_wave);
    (void)hipFree(p->d_ctrl);
    (void)hipFree(p->d_mbstats);
    qr::ppo_f32_release_graphs(p->f32_graphs);
    if (p->f32_scratch) (void)hipFree(p->f32_scratch);
    if (p->eg.exec) (void)hipGraphExecDestroy(p->eg.exec);
    if (p->capture_stream) (void)hipStreamDestroy(p->capture_stream);
    delete p;
    return QR_OK;
}

#ifdef QR_PHASE_TIMING
__attribute__((visibility("default"))) int qr_ppo_debug_set_ticks(qr_ppo* p, unsigned long long* ticks_dev) { p->ticks = ticks_dev; return QR_OK; }
__attribute__((visibility("default"))) int qr_ppo_debug_set_apply_ticks(qr_ppo* p, unsigned long long* ticks_dev) { p->apply_ticks = ticks_dev; return QR_OK; }
#endif

int qr_ppo_num_params(const qr_ppo* p) { return p ? p->num_params : ppofail(QR_E_INVALID, "qr_ppo_num_params: null handle"); }

int qr_ppo_pack(qr_ppo* p, const float* theta_dev, void* stream) {
    if (!p || !theta_dev) return ppofail(QR_E_INVALID, "qr_ppo_pack: null argument");
    PPO_HIP(hipSetDevice(p->device));
    p->packed_theta = theta_dev;
    return dispatch_L(p->L, [&](auto Lc) { return PpoOps<decltype(Lc)::value>::pack(p, theta_dev, (hipStream_t)stream); });
}

int qr_ppo_control(qr_ppo* p, float target_kl, int32_t clear, void* stream) {
    if (!p) return ppofail(QR_E_INVALID, "qr_ppo_control: null handle");
    PPO_HIP(hipSetDevice(p->device));
    p->target_kl = target_kl;
    if (clear) {  // stop flag and counters (the barrier counters that follow them in PpoCtrl keep running)
        PPO_HIP(hipMemsetAsync(&p->d_ctrl->stop, 0, 4 * sizeof(int), (hipStream_t)stream));
    }
    return QR_OK;
}

int qr_ppo_status(qr_ppo* p, int32_t* out4, void* stream) {
    if (!p || !out4) return ppofail(QR_E_INVALID, "qr_ppo_status: null argument");
    PPO_HIP(hipSetDevice(p->device));
    PPO_HIP(hipMemcpyAsync(out4, &p->d_ctrl->stop, 4 * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    PPO_HIP(hipStreamSynchronize((hipStream_t)stream));
    return QR_OK;
}

static int epoch_begin_impl(qr_ppo* p, const float* adv_dev, const int32_t* idx_dev, int32_t B, int32_t num_minibatches, void* stream,
                            unsigned long long* bump);
int qr_ppo_epoch_begin(qr_ppo* p, const float* adv_dev, const int32_t* idx_dev, int32_t B, int32_t num_minibatches, void* stream) {
    return epoch_begin_impl(p, adv_dev, idx_dev, B, num_minibatches, stream, nullptr);
}
static int epoch_begin_impl(qr_ppo* p, const float* adv_dev, const int32_t* idx_dev, int32_t B, int32_t num_minibatches, void* stream,
                            unsigned long long* bump) {
    if (!p || !adv_dev || !idx_dev) return ppofail(QR_E_INVALID, "qr_ppo_epoch_begin: null argument");
    if (B < 64 || B > p->max_B || num_minibatches < 1 || num_minibatches > qr_ppo::kMaxEpochMinibatches)
        return ppofail(QR_E_INVALID, "qr_ppo_epoch_begin: bad minibatch size / count");
    PPO_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(qr::ppo_zero_table_kernel, dim3((2 * num_minibatches + 255) / 256), dim3(256), 0, st, p->d_mbstats, 2 * num_minibatches);
    hipLaunchKernelGGL(qr::ppo_adv_stats_kernel, dim3((B + 1023) / 1024, num_minibatches), dim3(1024), 0, st, adv_dev, idx_dev, B,
                       p->d_mbstats, bump);
    PPO_HIP(hipGetLastError());
    p->epoch_idx = idx_dev;
    p->epoch_B = B;
    p->epoch_count = num_minibatches;
    p->epoch_cursor = 0;
    return QR_OK;
}

int qr_ppo_grad(qr_ppo* p, const float* theta_dev, const float* obs_dev, const float* act_dev, const float* old_logp_dev,
                const float* adv_dev, const float* ret_dev, const int32_t* idx_dev, int32_t B, float clip, float vf_coef,
                float ent_coef, float* grad_out_dev, float* stats_dev, void* stream) {
    qr::PpoBatch b;
    if (int rc = fill_batch(p, b, theta_dev, obs_dev, act_dev, old_logp_dev, adv_dev, ret_dev, idx_dev, B, clip, vf_coef, ent_coef, stats_dev))
        return rc;
    if (!grad_out_dev) return ppofail(QR_E_INVALID, "qr_ppo_grad: null grad_out");
    PPO_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    return dispatch_L(p->L, [&](auto Lc) {
        constexpr int L = decltype(Lc)::value;
        // the operand images are current when they were built from this vector by qr_ppo_pack or kept in step with it by
        // qr_ppo_minibatch / qr_ppo_apply (callers that edit theta themselves call qr_ppo_pack, as documented)
        if (p->packed_theta != theta_dev) {
            if (int r = PpoOps<L>::pack(p, theta_dev, st)) return r;
            p->packed_theta = theta_dev;
        }
        int chunks = 0;
        if (int r = PpoOps<L>::grad(p, b, st, &chunks)) return r;
        qr::ApplyArgs a{};
        a.grad_out = grad_out_dev;
        a.chunks = chunks;
        a.Gw = 2 * b.G;
        a.ent_coef = ent_coef;
        a.stats = stats_dev;
        a.take_step = 0;
        return PpoOps<L>::apply(p, a, st);
    });
}

static int ppo_minibatch_impl(qr_ppo* p, float* theta_dev, float* adam_m_dev, float* adam_v_dev, const float* obs_dev, const float* act_dev,
                              const float* old_logp_dev, const float* adv_dev, const float* ret_dev, const int32_t* idx_dev, int32_t B,
                              float clip, float vf_coef, float ent_coef, float max_grad_norm, float lr, bool device_lr, float beta1,
                              float beta2, float eps, int32_t adam_step, float* stats_dev, void* stream) {
    qr::PpoBatch b;
    if (int rc = fill_batch(p, b, theta_dev, obs_dev, act_dev, old_logp_dev, adv_dev, ret_dev, idx_dev, B, clip, vf_coef, ent_coef, stats_dev))
        return rc;
    if (!adam_m_dev || !adam_v_dev || adam_step < 0) return ppofail(QR_E_INVALID, "qr_ppo_minibatch: bad Adam state");
    PPO_HIP(hipSetDevice(p->device));
    p->packed_theta = theta_dev;   // the apply kernel keeps the images in step with this vector
    hipStream_t st = (hipStream_t)stream;
    return dispatch_L(p->L, [&](auto Lc) {
        constexpr int L = decltype(Lc)::value;
        int chunks = 0;
        if (int r = PpoOps<L>::grad(p, b, st, &chunks)) return r;  // images were packed by the previous call (or qr_ppo_pack)
        qr::ApplyArgs a{};
        a.theta = theta_dev; a.m = adam_m_dev; a.v = adam_v_dev;
        a.chunks = chunks;
        a.Gw = 2 * b.G;
        a.ent_coef = ent_coef;
        a.stats = stats_dev;
        a.kl_limit = p->target_kl > 0.0f ? 1.5f * p->target_kl * (float)B : 0.0f;
        a.take_step = 1;
        adam_constants(a, max_grad_norm, lr, device_lr, beta1, beta2, eps, adam_step);
        return PpoOps<L>::apply(p, a, st);
    });
}

int qr_ppo_minibatch(qr_ppo* p, float* theta_dev, float* adam_m_dev, float* adam_v_dev, const float* obs_dev, const float* act_dev,
                     const float* old_logp_dev, const float* adv_dev, const float* ret_dev, const int32_t* idx_dev, int32_t B,
                     float clip, float vf_coef, float ent_coef, float max_grad_norm, float lr, float beta1, float beta2, float eps,
                     int32_t adam_step, float* stats_dev, void* stream) {
    if (!(lr >= 0.0f)) return ppofail(QR_E_INVALID, "qr_ppo_minibatch: the learning rate must be >= 0");
    return ppo_minibatch_impl(p, theta_dev, adam_m_dev, adam_v_dev, obs_dev, act_dev, old_logp_dev, adv_dev, ret_dev, idx_dev, B, clip,
                              vf_coef, ent_coef, max_grad_norm, lr, false, beta1, beta2, eps, adam_step, stats_dev, stream);
}

// Data-parallel training: every rank calls qr_ppo_grad on its shard of the minibatch, the caller averages the [n + 4] vector
// (gradient + minibatch statistics: one all-reduce over RCCL), then every rank applies the same update with qr_ppo_apply --
// including the same target-KL decision, because the KL sum travels with the gradient.
int qr_ppo_apply(qr_ppo* p, float* theta_dev, float* adam_m_dev, float* adam_v_dev, float* grad_dev, int32_t B, float max_grad_norm,
                 float lr, float beta1, float beta2, float eps, int32_t adam_step, float* stats_dev, void* stream) {
    if (!p || !theta_dev || !adam_m_dev || !adam_v_dev || !grad_dev || adam_step < 0 || B < 1 || !(lr >= 0.0f))
        return ppofail(QR_E_INVALID, "qr_ppo_apply: bad argument (null pointer, adam_step < 0, B < 1 or a learning rate below 0)");
    PPO_HIP(hipSetDevice(p->device));
    p->packed_theta = theta_dev;
    hipStream_t st = (hipStream_t)stream;
    return dispatch_L(p->L, [&](auto Lc) {
        constexpr int L = decltype(Lc)::value;
        qr::ApplyArgs a{};
        a.theta = theta_dev; a.m = adam_m_dev; a.v = adam_v_dev;
        a.ext_grad = grad_dev;
        a.stats = stats_dev;
        a.kl_limit = p->target_kl > 0.0f ? 1.5f * p->target_kl * (float)B : 0.0f;
        a.take_step = 1;
        adam_constants(a, max_grad_norm, lr, false, beta1, beta2, eps, adam_step);
        return PpoOps<L>::apply(p, a, st);
    });
}

// Device-resident optimiser step count (see PpoCtrl::adam_t): read it for a checkpoint, set it when one is loaded.
int qr_ppo_adam_step(qr_ppo* p, int32_t* value, int32_t set, void* stream) {
    if (!p || !value) return ppofail(QR_E_INVALID, "qr_ppo_adam_step: null argument");
    if (set && *value < 0) return ppofail(QR_E_INVALID, "qr_ppo_adam_step: negative step count");
    PPO_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    if (set) PPO_HIP(hipMemcpyAsync(&p->d_ctrl->adam_t, value, sizeof(int32_t), hipMemcpyHostToDevice, st));
    else PPO_HIP(hipMemcpyAsync(value, &p->d_ctrl->adam_t, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PPO_HIP(hipStreamSynchronize(st));
    return QR_OK;
}

// On-device epoch permutations (qr_ppo_epoch with device_shuffle): state[0] = seed, state[1] = epochs shuffled so far.
// set == 0 reads both (for a checkpoint), set != 0 writes both.  Blocks.
int qr_ppo_shuffle_state(qr_ppo* p, uint64_t* state2, int32_t set, void* stream) {
    if (!p || !state2) return ppofail(QR_E_INVALID, "qr_ppo_shuffle_state: null argument");
    PPO_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    if (set) {
        p->shuffle_seed = state2[0];
        PPO_HIP(hipMemcpyAsync(&p->d_ctrl->shuffle_count, &state2[1], sizeof(uint64_t), hipMemcpyHostToDevice, st));
    } else {
        state2[0] = p->shuffle_seed;
        PPO_HIP(hipMemcpyAsync(&state2[1], &p->d_ctrl->shuffle_count, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    PPO_HIP(hipStreamSynchronize(st));
    return QR_OK;
}

// One whole epoch: the advantage statistics of every minibatch, then num_minibatches x (gradient kernel, apply kernel) on the rows
// perm[k B .. (k + 1) B), k = 0 .. num_minibatches - 1 -- enqueued as ONE replayed hipGraph.  Dependent kernel nodes of a graph start
// ~1 us sooner after each other than dependent stream launches (tools/ubench/launch_floor.hip), and there are 2 x 128 of them per
// epoch at the config-5 shape; the host issues one graph launch instead of 257 kernel launches.  Replays are valid because nothing a
// node's arguments name changes: the permutation is rewritten IN PLACE by the caller, the Adam step count and the learning rate
// are read from device memory, the early-stop flag is sticky on the device.  The graph is re-captured when any argument changes.
int qr_ppo_epoch(qr_ppo* p, float* theta_dev, float* adam_m_dev, float* adam_v_dev, const float* obs_dev, const float* act_dev,
                 const float* old_logp_dev, const float* adv_dev, const float* ret_dev, int32_t* perm_dev, int32_t B,
                 int32_t num_minibatches, int32_t num_epochs, int32_t device_shuffle, float clip, float vf_coef, float ent_coef,
                 float max_grad_norm, float lr, float beta1, float beta2, float eps, float* stats_dev, void* stream) {
    if (!p || !theta_dev || !adam_m_dev || !adam_v_dev || !obs_dev || !act_dev || !old_logp_dev || !adv_dev || !ret_dev || !perm_dev)
        return ppofail(QR_E_INVALID, "qr_ppo_epoch: null argument");
    if (B < 64 || B > p->max_B || num_minibatches < 1 || num_minibatches > qr_ppo::kMaxEpochMinibatches || !(lr >= 0.0f))
        return ppofail(QR_E_INVALID, "qr_ppo_epoch: bad minibatch size / count / learning rate");
    if (num_epochs < 1 || num_epochs > 64 || (num_epochs > 1 && !device_shuffle))
        return ppofail(QR_E_INVALID, "qr_ppo_epoch: num_epochs > 1 needs device_shuffle (the caller cannot rewrite the permutation in between)");
    PPO_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool caller_captures = st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    // the learning rate travels through device memory (it may follow a schedule; the epoch graph's nodes read PpoCtrl::lr).  Written by
    // a one-thread kernel: when the CALLER captures `st`, this node carries the value of the capturing call (lr is frozen into that
    // graph, like every other scalar argument of the call).
    hipLaunchKernelGGL(qr::ppo_set_lr_kernel, dim3(1), dim3(1), 0, st, &p->d_ctrl->lr, lr);
    PPO_HIP(hipGetLastError());
    p->packed_theta = theta_dev;
    // per-device kernel attributes are set outside the capture (not a stream operation)
    if (int rc = dispatch_L(p->L, [&](auto Lc) { return PpoOps<decltype(Lc)::value>::configure(p); })) return rc;
    const unsigned int rows = (unsigned int)B * (unsigned int)num_minibatches;
    auto enqueue = [&](hipStream_t s) -> int {
        for (int e = 0; e < num_epochs; ++e) {
            if (device_shuffle) {   // perm = keyed bijection of [0, rows) for (seed, device-resident epoch count)
                hipLaunchKernelGGL(qr::ppo_shuffle_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, perm_dev, rows, p->shuffle_seed,
                                   &p->d_ctrl->shuffle_count);
                PPO_HIP(hipGetLastError());
            }
            if (int rc = epoch_begin_impl(p, adv_dev, perm_dev, B, num_minibatches, s, device_shuffle ? &p->d_ctrl->shuffle_count : nullptr))
                return rc;
            for (int k = 0; k < num_minibatches; ++k)
                if (int rc = ppo_minibatch_impl(p, theta_dev, adam_m_dev, adam_v_dev, obs_dev, act_dev, old_logp_dev, adv_dev, ret_dev,
                                                perm_dev + (size_t)k * B, B, clip, vf_coef, ent_coef, max_grad_norm, 0.0f, true, beta1,
                                                beta2, eps, 0, stats_dev, s))
                    return rc;
        }
        return QR_OK;
    };
    if (caller_captures || !p->epoch_graph) return enqueue(st);   // a graph launch cannot be captured: plain nodes into the caller's graph
    qr_ppo::EpochGraph& g = p->eg;
    const bool hit = g.exec && g.theta == theta_dev && g.m == adam_m_dev && g.v == adam_v_dev && g.obs == obs_dev && g.act == act_dev &&
                     g.old_logp == old_logp_dev && g.adv == adv_dev && g.ret == ret_dev && g.perm == perm_dev && g.stats == stats_dev &&
                     g.B == B && g.M == num_minibatches && g.E == num_epochs && g.shuffle == device_shuffle && g.seed == p->shuffle_seed &&
                     g.clip == clip && g.vf_coef == vf_coef && g.ent_coef == ent_coef &&
                     g.max_grad_norm == max_grad_norm && g.beta1 == beta1 && g.beta2 == beta2 && g.eps == eps && g.target_kl == p->target_kl;
    if (!hit) {
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
        if (!p->capture_stream) PPO_HIP(hipStreamCreateWithFlags(&p->capture_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        PPO_HIP(hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeRelaxed));
        const int rc = enqueue(p->capture_stream);
        const hipError_t end = hipStreamEndCapture(p->capture_stream, &graph);
        if (rc != QR_OK || end != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc != QR_OK ? rc : ppofail(QR_E_HIP, std::string("qr_ppo_epoch: graph capture failed: ") + hipGetErrorString(end));
        }
        const hipError_t inst = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (inst != hipSuccess) {
            g.exec = nullptr;
            return ppofail(QR_E_HIP, std::string("qr_ppo_epoch: hipGraphInstantiate: ") + hipGetErrorString(inst));
        }
        g.theta = theta_dev; g.m = adam_m_dev; g.v = adam_v_dev; g.obs = obs_dev; g.act = act_dev; g.old_logp = old_logp_dev;
        g.adv = adv_dev; g.ret = ret_dev; g.perm = perm_dev; g.stats = stats_dev; g.B = B; g.M = num_minibatches;
        g.E = num_epochs; g.shuffle = device_shuffle; g.seed = p->shuffle_seed;
        g.clip = clip; g.vf_coef = vf_coef; g.ent_coef = ent_coef; g.max_grad_norm = max_grad_norm; g.beta1 = beta1; g.beta2 = beta2;
        g.eps = eps; g.target_kl = p->target_kl;
    }
    // the capture consumed the epoch table's host-side cursor; a replay does not run that host code, so leave it disarmed
    p->epoch_idx = nullptr;
    PPO_HIP(hipGraphLaunch(g.exec, st));
    return QR_OK;
}

// Forward pass of one of the two networks with the operand images of the last pack (0 = policy means, 1 = value in column 0):
// out_dev [n][4].  Same kernel as qr_policy_forward.
int qr_ppo_forward(qr_ppo* p, int32_t net, int32_t n, const float* obs_dev, float* out_dev, void* stream) {
    if (!p || !obs_dev || !out_dev || n < 1 || net < 0 || net > 1) return ppofail(QR_E_INVALID, "qr_ppo_forward: bad argument");
    PPO_HIP(hipSetDevice(p->device));
    const hipError_t e = qr::launch_policy(p->L, p->d_images + (size_t)net * p->image_half8, n, obs_dev, out_dev, (hipStream_t)stream);
    if (e != hipSuccess) return ppofail(QR_E_HIP, std::string("qr_ppo_forward: ") + hipGetErrorString(e));
    return QR_OK;
}

// GAE(lambda) advantages / returns of a rollout [T][N] and (optionally) episode statistics; see ppo_gae_kernel.
int qr_ppo_gae(qr_ppo* p, int32_t T, int32_t N, const float* rew_dev, const float* done_dev, const float* val_dev,
               const float* last_val_dev, const float* term_val_dev, float gamma, float lam, float* adv_out_dev, float* ret_out_dev,
               float* ep_ret_dev, float* ep_len_dev, float* ep_gates_dev, float* fin_dev, void* stream) {
    if (!p || !rew_dev || !done_dev || !val_dev || !last_val_dev || !adv_out_dev || !ret_out_dev || T < 1 || N < 1)
        return ppofail(QR_E_INVALID, "qr_ppo_gae: bad argument");
    if (ep_ret_dev && (!ep_len_dev || !ep_gates_dev)) return ppofail(QR_E_INVALID, "qr_ppo_gae: episode state needs all three arrays");
    PPO_HIP(hipSetDevice(p->device));
    hipLaunchKernelGGL(qr::ppo_gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, N, rew_dev, done_dev, val_dev,
                       last_val_dev, term_val_dev, gamma, lam, adv_out_dev, ret_out_dev, ep_ret_dev, ep_len_dev, ep_gates_dev, fin_dev);
    PPO_HIP(hipGetLastError());
    return QR_OK;
}

}  // extern "C"
