// quadrace_policy.hpp -- the reference's policy network on the gfx950 matrix cores.
//
// The policy the reference trains and deploys is an MLP  obs[L] -> 120 -> 120 -> 120 -> 4  with ReLU
// (SB3 `MlpPolicy`, net_arch pi=[120,120,120], R:783; generated C twin c_code/neural_network.c:397-430 `nn_forward`).
// At 65 536 envs it is ~32 k MAC per env-step -- 20x the arithmetic of the env step itself -- and it is a real
// GEMM chain, so it runs on MFMA: per wave (64 envs)  H^T[128 x 64] = W[128 x K] * X^T[K x 64]  with
// v_mfma_f32_32x32x16_f16 (f16 operands, f32 accumulate; 160 MFMAs per step).
//
// Layout trick that keeps the whole chain in registers (operand layouts verified on MI355X, tools/ubench/mfma_layout.hip):
//   A (weights, from LDS):  lane l holds  W[row = 32t + (l&31)][k-slot (h = l>>5, j = 0..7)]
//   B (activations):        lane l holds  act[k-slot (h, j)] of env (32*et + (l&31))
//   D (result):             lane l, reg r holds row 32t + rho(r, h), rho(r,h) = (r&3) + 8*(r>>2) + 4h, of env 32*et + (l&31)
// A k-slot is just a name for a hidden unit, so the NEXT layer's K-step (t, s) uses k-slot (h, j) := hidden unit
// 32t + rho(8s + j, h): its B operand is then exactly registers 8s..8s+7 of accumulator tile t (ReLU + f16 pack, all
// lane-local), and the host packs the weight columns in the same order.  Biases ride along as one more input: hidden
// unit 120 of every layer is wired to the constant 1 (weights row 120 = e_bias), and column `bias index` of each
// weight matrix holds the bias vector.  The first layer's B operand comes from the lane-per-env observation registers
// via v_permlane32_swap (one swap yields both env tiles); the last layer's 4 outputs return to lane = env the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qr {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel, and one process may drive handles on
// several GPUs (include/quadrace.h): each launch site keeps a mask of the device ordinals it has configured.
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, unsigned long long& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((done_mask >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    return e;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16p __attribute__((ext_vector_type(16)));
typedef unsigned u32x2p __attribute__((ext_vector_type(2)));

constexpr int kPolHidden = 120;
constexpr int kPolHiddenPad = 128;
constexpr int kPolBiasUnit = 120;  // hidden unit wired to the constant 1

template <int L>
struct PolicyDims {
    static constexpr int kIn = L + 1;                          // + constant-1 input
    static constexpr int kSteps1 = (kIn + 15) / 16;            // K-steps of layer 1
    static constexpr int kHalf8PerLayer1 = 4 * kSteps1 * 64;   // half8 elements (tiles * ksteps * lanes)
    static constexpr int kHalf8Hidden = 4 * 8 * 64;
    static constexpr int kHalf8Out = 1 * 8 * 64;
    static constexpr int kOff2 = kHalf8PerLayer1;
    static constexpr int kOff3 = kOff2 + kHalf8Hidden;
    static constexpr int kOff4 = kOff3 + kHalf8Hidden;
    static constexpr int kTotalHalf8 = kOff4 + kHalf8Out;      // * 16 bytes
};

__device__ __forceinline__ void swap32(float a, float b, float& lo, float& hi) {
    const u32x2p r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo = __uint_as_float(r.x);
    hi = __uint_as_float(r.y);
}

// relu + f16 pack of accumulator registers 8s..8s+7 -> B operand of the next layer's K-step
// (convert first, then one packed f16 max per two values: v_cvt_pk_f16_f32 + v_pk_max_f16 = 1 instruction per
// element instead of 2.5 for an f32 fmaxf, which also canonicalises its input)
__device__ __forceinline__ half8 relu_pack(const f32x16p& acc, int s) {
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)acc[8 * s + j];
    const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const _Float16 m = (_Float16)65504.0f;  // saturate instead of overflowing to inf (inf * 0 = NaN downstream)
    const half8 top = {m, m, m, m, m, m, m, m};
    return __builtin_elementwise_min(__builtin_elementwise_max(b, zero), top);
}

// f32 -> f16 with saturation to the finite range; NaN becomes 0 (maxnum/minnum drop the NaN operand)
__device__ __forceinline__ half8 sat_pack(const float* v) {
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)v[j];
    const _Float16 m = (_Float16)65504.0f;
    const half8 top = {m, m, m, m, m, m, m, m};
    const half8 bot = {-m, -m, -m, -m, -m, -m, -m, -m};
    return __builtin_elementwise_min(__builtin_elementwise_max(b, bot), top);
}

// relu + saturation + f16 pack of two accumulator values (v_cvt_pk_f16_f32, v_pk_max_f16 against 0, v_pk_min_f16 against
// 65504 -- the instructions relu_pack() compiles to).  The max / min pair is inline asm so that the compiler keeps these
// units where the source puts them: between the MFMAs of the next output tile (see policy_layer).
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t relu_pack2(float lo, float hi) {
    // The conversion stays compiler-visible: it READS matrix-core results, and the wait states between an MFMA and a
    // VALU read of its destination are inserted by the compiler's hazard recogniser, which does not look inside inline asm.
    typedef _Float16 half2p __attribute__((ext_vector_type(2)));
    const half2p c = {(_Float16)lo, (_Float16)hi};
    uint32_t r = __builtin_bit_cast(uint32_t, c);
    asm("v_pk_max_f16 %0, %0, 0\n\tv_pk_min_f16 %0, %0, %1" : "+v"(r) : "v"(0x7BFF7BFFu));   // 65504 in both halves
    return r;
}

// One layer with KS K-steps per 32-row output tile: in[et][KS] -> out[et][8].  W = this layer's A operands in LDS.
//
// A wave issues in order and an MFMA issued while the matrix core is busy stalls the wave, so everything else must sit
// BETWEEN the MFMAs in program order to overlap with them.  The schedule is therefore written out and pinned with
// sched_barrier (left to itself the scheduler emitted "ds_read, s_waitcnt lgkmcnt(0), MFMA, MFMA" per K-step -- the LDS
// latency exposed 32 times per layer -- and one lump of 48 pack instructions per tile behind the MFMAs):
//   K-step g of output tile t:   ds_read of the operand 4 K-steps ahead (ring of 4)
//                                2 MFMAs (env tiles 0 and 1) on tile t's operand g
//                                1/KS of the ReLU + f16 pack of tile t-1 (its accumulators finished a tile ago)
//                                `filler(slot)`: a slice of INDEPENDENT caller work (the closed-loop kernel draws its
//                                action noise here) -- a K-step leaves ~25 of its 32 VALU issue slots unused
struct NoFiller {
    __device__ __forceinline__ void operator()(int) const {}
};
// kChainOut (third hidden layer): the output layer's image follows this layer's in LDS, so the ring simply keeps fetching, and
// the output layer's K-steps 0..5 -- which only need this layer's tiles 0..2 -- are issued in the groups that pack tile 3
// (they have no MFMAs of their own); K-steps 6 and 7 follow.  accO[et] = the output tile (rows 0..3 = action means).
template <int KS, class Filler = NoFiller, bool kChainOut = false>
__device__ __forceinline__ void policy_layer(const half8* __restrict__ W, int lane, const half8 (&in)[2][KS],
                                             half8 (&out)[2][8], Filler&& filler = Filler(), f32x16p* accO = nullptr) {
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    f32x16p acc[2][2];  // [tile parity][env tile]
    constexpr int kTotal = 4 * KS + (kChainOut ? 8 : 0);   // image elements (K-step operands) this call consumes
    constexpr int D = KS < 4 ? KS : 4;   // A operands are fetched D K-steps (>= 256 cycles of MFMA time) ahead, into a ring of
    half8 a[D];                          // D registers-quads (a whole-tile double buffer cost 64 VGPRs and pushed the closed-loop
    u32x4p o32[2][8];                    // kernel into AGPR copies); o32 = the packed outputs, dword by dword
#pragma unroll
    for (int q = 0; q < D; ++q) a[q] = W[q * 64 + lane];   // group q = t KS + g reads element (t KS + g) of the layer image
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t <= 4; ++t) {
#pragma unroll
        for (int g = 0; g < KS; ++g) {
            const int q = t * KS + g;
            if (t < 4) {
                acc[t & 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % D], in[0][g], g == 0 ? zero : acc[t & 1][0], 0, 0, 0);
                acc[t & 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % D], in[1][g], g == 0 ? zero : acc[t & 1][1], 0, 0, 0);
                if (q + D < kTotal) a[q % D] = W[(q + D) * 64 + lane];
            }
            if (kChainOut && t == 4 && g < 6) {
                accO[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % D], __builtin_bit_cast(half8, o32[0][g]), g == 0 ? zero : accO[0], 0, 0, 0);
                accO[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % D], __builtin_bit_cast(half8, o32[1][g]), g == 0 ? zero : accO[1], 0, 0, 0);
                if (q + D < kTotal) a[q % D] = W[(q + D) * 64 + lane];
            }
            if (t > 0) {  // dwords [16 g / KS, 16 (g + 1) / KS) of the previous tile: d = 8 et + 4 s + dd
                const int p = (t - 1) & 1;
#pragma unroll
                for (int d = (16 * g) / KS; d < (16 * (g + 1)) / KS; ++d) {
                    const int et = d >> 3, sh = (d >> 2) & 1, dd = d & 3;
                    o32[et][2 * (t - 1) + sh][dd] = relu_pack2(acc[p][et][8 * sh + 2 * dd], acc[p][et][8 * sh + 2 * dd + 1]);
                }
            }
            if (t < 4) filler(t * KS + g);   // slots 0 .. 4 KS - 1, one per MFMA pair
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int k = 0; k < 8; ++k) out[et][k] = __builtin_bit_cast(half8, o32[et][k]);
    if (kChainOut) {
#pragma unroll
        for (int g = 6; g < 8; ++g) {
            const int q = 4 * KS + g;
            accO[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % D], out[0][g], accO[0], 0, 0, 0);
            accO[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % D], out[1][g], accO[1], 0, 0, 0);
        }
    }
}

// Full policy forward for the wave's 64 envs.  o[L] = this lane's observation (lane = env); mean[4] = action means
// of this lane's env.  Wlds = packed f16 weights in LDS (PolicyDims<L> layout).  Must be called by all 64 lanes.
// `filler2(slot)` / `filler3(slot)`, slot = 0..31, are called once per K-step of the second / third hidden layer (see
// policy_layer): independent caller work that runs in the MFMA shadow.
template <int L, class Filler2 = NoFiller, class Filler3 = NoFiller>
__device__ __forceinline__ void policy_forward(const half8* __restrict__ Wlds, int lane, const float* o, float mean[4],
                                               Filler2&& filler2 = Filler2(), Filler3&& filler3 = Filler3()) {
    using D = PolicyDims<L>;
    // ---- layer 1 B operands: input k = 16s + 8h + j; lanes 0..31 supply h = 0, lanes 32..63 h = 1 (of env l-32)
    half8 in1[2][D::kSteps1];
#pragma unroll
    for (int s = 0; s < D::kSteps1; ++s) {
        float t0[8], t1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
            const float x0 = (k0 < L) ? o[k0 < L ? k0 : 0] : (k0 == L ? 1.0f : 0.0f);
            const float x1 = (k1 < L) ? o[k1 < L ? k1 : 0] : (k1 == L ? 1.0f : 0.0f);
            swap32(x0, x1, t0[j], t1[j]);
        }
        in1[0][s] = sat_pack(t0);  // observations can be large (rates up to 1000 rad/s) or NaN: keep f16 finite
        in1[1][s] = sat_pack(t1);
    }
    half8 h1[2][8], h2[2][8];
    policy_layer<D::kSteps1>(Wlds, lane, in1, h1);
    policy_layer<8>(Wlds + D::kOff2, lane, h1, h2, filler2);
    // third hidden layer + output layer (one 32-row tile, rows 0..3 = action means) as one pipelined sequence
    static_assert(D::kOff4 == D::kOff3 + 4 * 8 * 64, "the output image must follow the third layer's");
    f32x16p accO[2];
    policy_layer<8, Filler3, true>(Wlds + D::kOff3, lane, h2, h1, static_cast<Filler3&&>(filler3), accO);
    const f32x16p acc0 = accO[0], acc1 = accO[1];
    // rows 0..3 live in registers 0..3 of lanes 0..31 (h = 0) of each env tile: bring tile 1 to lanes 32..63
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float lo, hi;
        swap32(acc0[r], acc1[r], lo, hi);
        mean[r] = lo;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Reference-precision ("f32-class") forward of the same network (round 6).  The reference evaluates the policy
// in float32 (SB3 / torch, R:783-795; generated C twin c_code/neural_network.c:397-430); policy_kernel above rounds every operand to
// one f16 (max |d mean| 7e-4 against nn_forward).  Here BOTH operands of every layer are split into two f16 pieces, exactly like
// layer 1 of the residual MLPs (quadrace_device.hpp residual_mlp):
//     x = X0 + X1,  X0 = f16(x), X1 = f16(x - X0)          w = W0 + W1 (host side, once)
//     w x  ~  W0 X0 + W1 X0 + W0 X1                         (the dropped W1 X1 is <= 2^-22 |w x|; f32 accumulation on the matrix core)
// -- three matrix instructions per K-step instead of one, same operand layouts, same "accumulator registers are the next layer's k-slots"
// trick, so the chain still never leaves the registers.  The low-piece image W1 is read from global memory (80 KB, L2-resident, shared by
// every workgroup); W0 is staged in LDS like policy_kernel's image.  Not hand-scheduled: this is the accuracy path (evaluation,
// precision="f32" collection), the f16 kernel stays the throughput path.
// |x| beyond the f16 range: X0 saturates at +-65504 and X1 carries the rest (up to 131 008: no observation or activation gets there).
__device__ __forceinline__ void split_pack(const float* v, half8& p0, half8& p1) {
    float r[8];
    p0 = sat_pack(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j] - (float)p0[j];   // exact in f32 (NaN -> p0 = 0, r = NaN -> sat_pack gives 0)
    p1 = sat_pack(r);
}

template <int KS>
__device__ __forceinline__ void policy_layer_f32class(const half8* __restrict__ W0, const half8* __restrict__ W1g, int lane,
                                                      const half8 (&in0)[2][KS], const half8 (&in1)[2][KS], half8 (&out0)[2][8],
                                                      half8 (&out1)[2][8]) {
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16p acc[2] = {zero, zero};
#pragma unroll
        for (int g = 0; g < KS; ++g) {
            const half8 a0 = W0[(t * KS + g) * 64 + lane], a1 = W1g[(t * KS + g) * 64 + lane];
#pragma unroll
            for (int et = 0; et < 2; ++et) {   // small terms first
                acc[et] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, in0[et][g], acc[et], 0, 0, 0);
                acc[et] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, in1[et][g], acc[et], 0, 0, 0);
                acc[et] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, in0[et][g], acc[et], 0, 0, 0);
            }
        }
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(acc[et][8 * sh + j], 0.0f);   // ReLU in f32 (torch semantics; the bias unit's 1 stays 1)
                split_pack(v, out0[et][2 * t + sh], out1[et][2 * t + sh]);
            }
    }
}

// Full f32-class forward for the wave's 64 envs: W0 = image of the high pieces (LDS), W1g = image of the low pieces (global memory).
template <int L>
__device__ __forceinline__ void policy_forward_f32class(const half8* __restrict__ W0, const half8* __restrict__ W1g, int lane, const float* o,
                                                        float mean[4]) {
    using D = PolicyDims<L>;
    half8 in0[2][D::kSteps1], in1[2][D::kSteps1];
#pragma unroll
    for (int s = 0; s < D::kSteps1; ++s) {   // layer-1 B operands as in policy_forward(), both pieces
        float t0[8], t1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
            const float x0 = (k0 < L) ? o[k0 < L ? k0 : 0] : (k0 == L ? 1.0f : 0.0f);
            const float x1 = (k1 < L) ? o[k1 < L ? k1 : 0] : (k1 == L ? 1.0f : 0.0f);
            swap32(x0, x1, t0[j], t1[j]);
        }
        split_pack(t0, in0[0][s], in1[0][s]);
        split_pack(t1, in0[1][s], in1[1][s]);
    }
    half8 h0[2][8], h1[2][8], g0[2][8], g1[2][8];
    policy_layer_f32class<D::kSteps1>(W0, W1g, lane, in0, in1, h0, h1);
    policy_layer_f32class<8>(W0 + D::kOff2, W1g + D::kOff2, lane, h0, h1, g0, g1);
    policy_layer_f32class<8>(W0 + D::kOff3, W1g + D::kOff3, lane, g0, g1, h0, h1);
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    f32x16p accO[2] = {zero, zero};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const half8 a0 = W0[D::kOff4 + g * 64 + lane], a1 = W1g[D::kOff4 + g * 64 + lane];
#pragma unroll
        for (int et = 0; et < 2; ++et) {
            accO[et] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, h0[et][g], accO[et], 0, 0, 0);
            accO[et] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, h1[et][g], accO[et], 0, 0, 0);
            accO[et] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, h0[et][g], accO[et], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // rows 0..3 live in registers 0..3 of lanes 0..31 of each env tile: tile 1 to lanes 32..63
        float lo, hi;
        swap32(accO[0][r], accO[1][r], lo, hi);
        mean[r] = lo;
    }
}

// Arguments of the closed-loop rollout (policy + sampling inside the env rollout kernel)
struct PolicyArgs {
    const half8* weights;   // packed f16 image (PolicyDims<L> layout)
    const half8* weights_lo;  // image of the low f16 pieces (f32-class forward); used when `f32class` is set
    int f32class;             // 1: reference-precision forward (policy_forward_f32class) instead of the f16-operand one
    float std[4];           // exp(log_std)
    float logp_const;       // -sum(log_std) - 2*log(2*pi)
    uint32_t seed_lo, seed_hi;  // Philox key of the action noise
    uint32_t step_lo, step_hi;  // global step counter of this call's first step (noise stream position)
    int deterministic;          // 1: action = mean (no noise)
};

}  // namespace qr
