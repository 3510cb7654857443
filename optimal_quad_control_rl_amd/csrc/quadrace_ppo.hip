// quadrace_ppo.hip -- PPO minibatch update on the gfx950 matrix cores (BASELINE config 5).
//
// The reference trains with SB3's PPO (R:783-795): two separate ReLU MLPs  obs[L] -> 120 -> 120 -> 120 -> {4 | 1}
// (policy mean / value), a state-independent log-std, clipped surrogate + vf_coef * MSE value loss, per-minibatch
// advantage normalisation, global grad-norm clipping, Adam.  With the collect phase fused into one kernel
// (qr_rollout_policy) torch's minibatch update was > 95 % of training time: ~100 small launches around
// 16 k x 120 x 120 GEMMs.  Here one minibatch is six launches:
//
//   adv_stats   mean / unbiased std of the minibatch's advantages (SB3 normalises per minibatch)
//   phase A     per wave = 64 samples of one net: forward (f16 MFMA chain of quadrace_policy.hpp), per-sample loss
//               gradients, backward through the transposed weight images -- activations h_l and deltas d_l never leave
//               registers in the "lane = sample" form.  Each of them is also emitted in the transposed operand form
//               (lane = unit, k = sample) by multiplying with an identity operand on the matrix core, and written
//               to a scratch buffer (f16, ~105 KB per 64 samples and net).
//   phase B     dW_l = d_l^T x h_(l-1): one wave per 32x32 weight tile and sample chunk (split-K over the minibatch),
//               v_mfma_f32_32x32x16_f16 with k = sample, f32 atomics into the flat gradient.  Biases ride along as the
//               constant-1 unit of every layer.
//   norm, adam  global gradient norm -> clip scale; Adam on the flat parameter vector (also clears the gradient)
//   pack        f32 parameters -> f16 forward and transposed operand images for the next minibatch
//
// Operand layouts are those of quadrace_policy.hpp (verified on MI355X with tools/ubench/mfma_layout.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <type_traits>

#include "../../include/quadrace.h"
#include "quadrace_policy.hpp"

namespace qr {

constexpr int kPpoBlock = 256;
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
    static constexpr int kOffT4 = P::kTotalHalf8;  // W4^T: [t][lane]
    static constexpr int kOffT3 = kOffT4 + 4 * 64; // W3^T: [t][sp][lane]
    static constexpr int kOffT2 = kOffT3 + 4 * 8 * 64;
    static constexpr int kImage = kOffT2 + 4 * 8 * 64;  // half8 per net (forward image + 3 transposed images)
    // transposed-operand scratch: slots of [group][kk = 2*st + s][lane] half8
    static constexpr int kSlotX0 = 0, kSlotH1 = kIT, kSlotH2 = kIT + 4, kSlotH3 = kIT + 8;
    static constexpr int kSlotD1 = kIT + 12, kSlotD2 = kIT + 16, kSlotD3 = kIT + 20, kSlotD4 = kIT + 24;
    static constexpr int kSlots = kIT + 25;
    static constexpr int kJobsPerNet = 4 * kIT + 36;  // 32x32 weight tiles: layer1 4*kIT, layers 2,3 16 each, layer4 4
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
        } else if (e < D::kOffT4) {  // output layer: rows 0..O-1 of one 32-row tile
            const int sp = (e - P::kOff4) / 64;
            const int hid = 32 * (sp >> 1) + rho_(8 * (sp & 1) + j, h);
            if (c < O) val = hid < kH ? th[o.w4 + c * kH + hid] : (hid == kPolBiasUnit ? th[o.b4 + c] : 0.0f);
        } else if (e < D::kOffT3) {  // W4^T: row = hidden unit i, k-slot (h, j) = output unit 8h + j
            const int t = (e - D::kOffT4) / 64;
            const int i = 32 * t + c, oo = 8 * h + j;
            if (i < kH && oo < O) val = th[o.w4 + oo * kH + i];
        } else {  // W3^T then W2^T: row = input unit i of that layer, k-slots = its output units (accumulator-row order)
            const bool second = e >= D::kOffT2;
            const int e2 = e - (second ? D::kOffT2 : D::kOffT3);
            const int t = e2 / 512, sp = (e2 / 64) % 8;
            const int i = 32 * t + c, oo = 32 * (sp >> 1) + rho_(8 * (sp & 1) + j, h);
            if (i < kH && oo < kH) val = th[(second ? o.w2 : o.w3) + oo * kH + i];
        }
        v[j] = (_Float16)val;
    }
    images[(size_t)net * D::kImage + e] = v;
}

// ---- adv_stats: mean and 1 / (unbiased std + 1e-8) of adv[idx[0..B)] ---------------------------------------------------
__global__ void __launch_bounds__(1024) ppo_adv_stats_kernel(const float* __restrict__ adv, const int* __restrict__ idx, int B,
                                                             float* __restrict__ out) {
    __shared__ double s1[1024], s2[1024];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < B; i += 1024) {
        const double x = adv[idx[i]];
        a += x;
        b += x * x;
    }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s1[threadIdx.x] += s1[threadIdx.x + w];
            s2[threadIdx.x] += s2[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = s1[0] / B;
        double var = (s2[0] - B * mean * mean) / (B > 1 ? B - 1 : 1);
        if (var < 0.0) var = 0.0;
        out[0] = (float)mean;
        out[1] = (float)(1.0 / (sqrt(var) + 1e-8));
    }
}

// ---- phase A ------------------------------------------------------------------------------------------------------
struct PpoBatch {
    const float* obs;       // [rows][L]
    const float* act;       // [rows][4]
    const float* old_logp;  // [rows]
    const float* adv;       // [rows]
    const float* ret;       // [rows]
    const int* idx;         // [B] rows of this minibatch
    int B, G;               // G = B / 64 groups
    float clip, vf_coef, ent_coef;
    const float* adv_stats;  // [mean, rstd]
    const float* theta;      // flat parameters (log_std is read from here)
    const half8* images;     // [2][kImage]
    half8* tbuf;             // [2][kSlots][G][4][64]
    float* grad;             // flat gradient (atomics)
    float* stats;            // [0] sum pg loss, [1] sum value loss, [2] sum approx kl, [3] clipped count (atomics)
};

__device__ __forceinline__ half8 plain_pack(const f32x16p& acc, int s) {
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)acc[8 * s + j];
    return b;
}
// mask word layout: m[t >> 1] bit 16 * (t & 1) + r  <->  accumulator register r of output tile t
__device__ __forceinline__ half8 mask_pack(const f32x16p& acc, uint32_t word, int t, int s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ((word >> (16 * (t & 1) + 8 * s + j)) & 1u) ? acc[8 * s + j] : 0.0f;
    return sat_pack(v);
}
__device__ __forceinline__ uint32_t relu_bits(const f32x16p& acc, int t) {
    uint32_t w = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) w |= (acc[r] > 0.0f ? 1u : 0u) << (16 * (t & 1) + r);
    return w;
}

// One 128-unit layer: in[et][KS] -> out[et][8].  Forward (BWD = false): out = relu(acc) packed, mask = (acc > 0).
// Backward (BWD = true): out = acc where mask is set (the ReLU derivative of the layer being entered), else 0.
template <int KS, bool BWD>
__device__ __forceinline__ void mlp_layer(const half8* __restrict__ W, int lane, const half8 (&in)[2][KS], half8 (&out)[2][8],
                                          uint32_t (&mask)[2][2]) {
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (!BWD) mask[0][0] = mask[0][1] = mask[1][0] = mask[1][1] = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16p acc0 = zero, acc1 = zero;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const half8 a = W[(t * KS + s) * 64 + lane];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[1][s], acc1, 0, 0, 0);
        }
        if (BWD) {
            out[0][2 * t] = mask_pack(acc0, mask[0][t >> 1], t, 0);
            out[0][2 * t + 1] = mask_pack(acc0, mask[0][t >> 1], t, 1);
            out[1][2 * t] = mask_pack(acc1, mask[1][t >> 1], t, 0);
            out[1][2 * t + 1] = mask_pack(acc1, mask[1][t >> 1], t, 1);
        } else {
            mask[0][t >> 1] |= relu_bits(acc0, t);
            mask[1][t >> 1] |= relu_bits(acc1, t);
            out[0][2 * t] = relu_pack(acc0, 0);
            out[0][2 * t + 1] = relu_pack(acc0, 1);
            out[1][2 * t] = relu_pack(acc1, 0);
            out[1][2 * t + 1] = relu_pack(acc1, 1);
        }
    }
}

// Transposed operand form of a 128-unit matrix held as packs X[st][K-step]: multiply with the identity on the matrix
// core.  D = X[st] (rows = samples, k = units of tile ut) x Id (k -> column unit)  =>  lane = unit, registers = samples.
__device__ __forceinline__ void tstore_hidden(const half8 (&X)[2][8], half8* __restrict__ dst, size_t slot_stride, int lane) {
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const int c = lane & 31, h = lane >> 5;
    half8 id[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) id[s][j] = (rho_(8 * s + j, h) == c) ? (_Float16)1.0f : (_Float16)0.0f;
#pragma unroll
    for (int ut = 0; ut < 4; ++ut)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16p acc = zero;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[st][2 * ut], id[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[st][2 * ut + 1], id[1], acc, 0, 0, 0);
            dst[ut * slot_stride + (2 * st) * 64] = plain_pack(acc, 0);
            dst[ut * slot_stride + (2 * st + 1) * 64] = plain_pack(acc, 1);
        }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int L>
__global__ void __launch_bounds__(kPpoBlock, 1) ppo_phase_a_kernel(PpoBatch a) {
    using D = PpoDims<L>;
    using P = PolicyDims<L>;
    constexpr int KS1 = P::kSteps1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* W = reinterpret_cast<half8*>(smem);
    const int net = blockIdx.y;
    {
        const float4* src = reinterpret_cast<const float4*>(a.images + (size_t)net * D::kImage);
        float4* dst = reinterpret_cast<float4*>(W);
        for (int i = threadIdx.x; i < D::kImage; i += kPpoBlock) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int g = blockIdx.x * (kPpoBlock / 64) + (threadIdx.x >> 6);
    if (g >= a.G) return;  // whole wave
    const int b = a.idx[g * 64 + lane];
    const f32x16p zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const size_t slot_stride = (size_t)a.G * 256;
    half8* tb = a.tbuf + ((size_t)net * D::kSlots * a.G + g) * 256 + lane;  // slot 0, this group, kk = 0

    // ---- layer-1 operand (input k = 16 s + 8 h + j; input L = constant 1), as in policy_forward
    half8 in1[2][KS1];
    {
        float o[L];
        const float* row = a.obs + (size_t)b * L;
#pragma unroll
        for (int k = 0; k < L; ++k) o[k] = row[k];
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            float t0[8], t1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k0 = 16 * s + j, k1 = 16 * s + 8 + j;
                const float x0 = (k0 < L) ? o[k0 < L ? k0 : 0] : (k0 == L ? 1.0f : 0.0f);
                const float x1 = (k1 < L) ? o[k1 < L ? k1 : 0] : (k1 == L ? 1.0f : 0.0f);
                swap32(x0, x1, t0[j], t1[j]);
            }
            in1[0][s] = sat_pack(t0);
            in1[1][s] = sat_pack(t1);
        }
    }
    // transposed inputs: column unit = input index
#pragma unroll
    for (int ut = 0; ut < D::kIT; ++ut)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f32x16p acc = zero;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                half8 id;
#pragma unroll
                for (int j = 0; j < 8; ++j) id[j] = (16 * s + 8 * h + j == 32 * ut + c) ? (_Float16)1.0f : (_Float16)0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(in1[st][s], id, acc, 0, 0, 0);
            }
            tb[(D::kSlotX0 + ut) * slot_stride + (2 * st) * 64] = plain_pack(acc, 0);
            tb[(D::kSlotX0 + ut) * slot_stride + (2 * st + 1) * 64] = plain_pack(acc, 1);
        }

    // ---- forward
    uint32_t m1[2][2], m2[2][2], m3[2][2];
    half8 x[2][8], y[2][8];
    mlp_layer<KS1, false>(W, lane, in1, x, m1);
    tstore_hidden(x, tb + D::kSlotH1 * slot_stride, slot_stride, lane);
    mlp_layer<8, false>(W + P::kOff2, lane, x, y, m2);
    tstore_hidden(y, tb + D::kSlotH2 * slot_stride, slot_stride, lane);
    mlp_layer<8, false>(W + P::kOff3, lane, y, x, m3);  // x = h3
    tstore_hidden(x, tb + D::kSlotH3 * slot_stride, slot_stride, lane);
    float out4[4];
    {
        f32x16p acc0 = zero, acc1 = zero;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const half8 w = W[P::kOff4 + s * 64 + lane];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x[1][s], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float lo, hi;
            swap32(acc0[r], acc1[r], lo, hi);
            out4[r] = lo;  // rows 0..3 of this lane's sample
        }
    }

    // ---- per-sample loss gradients (x B; the 1/B of the batch means is applied in phase B)
    float dout[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (net == 0) {
        const float* log_std = a.theta + net_off(L, 4).total + net_off(L, 1).total;
        const float4 av = reinterpret_cast<const float4*>(a.act)[b];
        const float act[4] = {av.x, av.y, av.z, av.w};
        float z[4], inv_std[4], logp = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ls = log_std[k];
            inv_std[k] = __expf(-ls);
            z[k] = (act[k] - out4[k]) * inv_std[k];
            logp += -0.5f * z[k] * z[k] - ls - 0.9189385332046727f;
        }
        const float log_ratio = logp - a.old_logp[b];
        const float ratio = __expf(log_ratio);
        const float A = (a.adv[b] - a.adv_stats[0]) * a.adv_stats[1];
        const bool flows = A >= 0.0f ? (ratio <= 1.0f + a.clip) : (ratio >= 1.0f - a.clip);
        const float gl = flows ? -A * ratio : 0.0f;  // d loss / d logp
        float dls[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dout[k] = gl * z[k] * inv_std[k];
            dls[k] = gl * (z[k] * z[k] - 1.0f);
        }
        const float scale = 1.0f / (float)a.B;
        float* gls = a.grad + net_off(L, 4).total + net_off(L, 1).total;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sum = wave_sum(dls[k]);
            if (lane == 0) unsafeAtomicAdd(gls + k, sum * scale - (g == 0 ? a.ent_coef : 0.0f));  // entropy = sum(log_std) + const
        }
        if (a.stats) {
            const float clipped_ratio = fminf(fmaxf(ratio, 1.0f - a.clip), 1.0f + a.clip);
            const float pg = wave_sum(-fminf(A * ratio, A * clipped_ratio));
            const float kl = wave_sum((ratio - 1.0f) - log_ratio);
            const float cf = wave_sum(fabsf(ratio - 1.0f) > a.clip ? 1.0f : 0.0f);
            if (lane == 0) {
                unsafeAtomicAdd(a.stats + 0, pg);
                unsafeAtomicAdd(a.stats + 2, kl);
                unsafeAtomicAdd(a.stats + 3, cf);
            }
        }
    } else {
        const float err = out4[0] - a.ret[b];
        dout[0] = a.vf_coef * 2.0f * err;  // vf_coef * d mse / d v  (x B)
        if (a.stats) {
            const float vl = wave_sum(err * err);
            if (lane == 0) unsafeAtomicAdd(a.stats + 1, vl);
        }
    }

    // ---- output deltas as a B operand (k-slot (h, j) = output unit 8 h + j) and in transposed form
    half8 d4[2];
    {
        float t0[8], t1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) swap32(j < 4 ? dout[j < 4 ? j : 0] : 0.0f, 0.0f, t0[j], t1[j]);
        d4[0] = sat_pack(t0);
        d4[1] = sat_pack(t1);
        half8 id;
#pragma unroll
        for (int j = 0; j < 8; ++j) id[j] = (8 * h + j == c) ? (_Float16)1.0f : (_Float16)0.0f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const f32x16p acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(d4[st], id, zero, 0, 0, 0);
            tb[D::kSlotD4 * slot_stride + (2 * st) * 64] = plain_pack(acc, 0);
            tb[D::kSlotD4 * slot_stride + (2 * st + 1) * 64] = plain_pack(acc, 1);
        }
    }
    // ---- backward: d3 = (W4^T d4) * relu'(z3);  d2 = (W3^T d3) * relu'(z2);  d1 = (W2^T d2) * relu'(z1)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const half8 w = W[D::kOffT4 + t * 64 + lane];
        const f32x16p acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, d4[0], zero, 0, 0, 0);
        const f32x16p acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, d4[1], zero, 0, 0, 0);
        x[0][2 * t] = mask_pack(acc0, m3[0][t >> 1], t, 0);
        x[0][2 * t + 1] = mask_pack(acc0, m3[0][t >> 1], t, 1);
        x[1][2 * t] = mask_pack(acc1, m3[1][t >> 1], t, 0);
        x[1][2 * t + 1] = mask_pack(acc1, m3[1][t >> 1], t, 1);
    }
    tstore_hidden(x, tb + D::kSlotD3 * slot_stride, slot_stride, lane);
    mlp_layer<8, true>(W + D::kOffT3, lane, x, y, m2);
    tstore_hidden(y, tb + D::kSlotD2 * slot_stride, slot_stride, lane);
    mlp_layer<8, true>(W + D::kOffT2, lane, y, x, m1);
    tstore_hidden(x, tb + D::kSlotD1 * slot_stride, slot_stride, lane);
}

// ---- phase B: weight gradients -------------------------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(64) ppo_phase_b_kernel(const half8* __restrict__ tbuf, float* __restrict__ grad, int G,
                                                         int groups_per_chunk, float scale) {
    using D = PpoDims<L>;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int net = blockIdx.x / D::kJobsPerNet;
    int j = blockIdx.x % D::kJobsPerNet;
    int layer, to, ti;
    if (j < 4 * D::kIT) { layer = 1; to = j / D::kIT; ti = j % D::kIT; }
    else if (j < 4 * D::kIT + 16) { j -= 4 * D::kIT; layer = 2; to = j >> 2; ti = j & 3; }
    else if (j < 4 * D::kIT + 32) { j -= 4 * D::kIT + 16; layer = 3; to = j >> 2; ti = j & 3; }
    else { layer = 4; to = 0; ti = j - (4 * D::kIT + 32); }
    const int slot_a = layer == 1 ? D::kSlotD1 + to : (layer == 2 ? D::kSlotD2 + to : (layer == 3 ? D::kSlotD3 + to : D::kSlotD4));
    const int slot_b = layer == 1 ? D::kSlotX0 + ti : (layer == 2 ? D::kSlotH1 + ti : (layer == 3 ? D::kSlotH2 + ti : D::kSlotH3 + ti));
    const half8* A = tbuf + ((size_t)net * D::kSlots + slot_a) * G * 256 + lane;
    const half8* Bm = tbuf + ((size_t)net * D::kSlots + slot_b) * G * 256 + lane;
    const int g0 = blockIdx.y * groups_per_chunk;
    const int g1 = min(G, g0 + groups_per_chunk);
    f32x16p acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int g = g0; g < g1; ++g) {
        half8 av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            av[kk] = A[((size_t)g * 4 + kk) * 64];
            bv[kk] = Bm[((size_t)g * 4 + kk) * 64];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[kk], bv[kk], acc, 0, 0, 0);
    }
    // D: register r of lane (c, h) = dW[row = 32 to + rho(r, h)][col = 32 ti + c]
    const int O = net == 0 ? 4 : 1;
    const NetOff o = net_off(L, O);
    float* gn = grad + (net == 0 ? 0 : net_off(L, 4).total);
    const int col = 32 * ti + c;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * to + rho_(r, h);
        const float v = acc[r] * scale;
        if (layer == 1) {
            if (row < kH) {
                if (col < L) unsafeAtomicAdd(gn + o.w1 + row * L + col, v);
                else if (col == L) unsafeAtomicAdd(gn + o.b1 + row, v);
            }
        } else if (layer == 4) {
            if (row < O) {
                if (col < kH) unsafeAtomicAdd(gn + o.w4 + row * kH + col, v);
                else if (col == kPolBiasUnit) unsafeAtomicAdd(gn + o.b4 + row, v);
            }
        } else {
            const int ow = layer == 2 ? o.w2 : o.w3, ob = layer == 2 ? o.b2 : o.b3;
            if (row < kH) {
                if (col < kH) unsafeAtomicAdd(gn + ow + row * kH + col, v);
                else if (col == kPolBiasUnit) unsafeAtomicAdd(gn + ob + row, v);
            }
        }
    }
}

// ---- global gradient norm -> clip scale; Adam ----------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) ppo_norm_kernel(const float* __restrict__ grad, int n, float max_norm,
                                                        float* __restrict__ out /* [norm, scale] */) {
    __shared__ double s[1024];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) a += (double)grad[i] * grad[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s[0]);
        out[0] = norm;
        out[1] = fminf(1.0f, max_norm / (norm + 1e-6f));  // torch.nn.utils.clip_grad_norm_
    }
}

__global__ void __launch_bounds__(256) ppo_adam_kernel(float* __restrict__ theta, float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ grad, int n, const float* __restrict__ clip,
                                                       float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = grad[i] * clip[1];
    if (!(fabsf(g) <= 3.0e38f)) g = 0.0f;  // a non-finite gradient never reaches the parameters
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    theta[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);  // torch.optim.Adam
    grad[i] = 0.0f;
}

}  // namespace qr

// ------------------------------------------------------------------------------------------------------------------------
struct qr_ppo {
    int L = 0, device = 0, max_B = 0, num_params = 0;
    int image_half8 = 0, slots = 0, jobs_per_net = 0;
    qr::half8* d_images = nullptr;
    qr::half8* d_tbuf = nullptr;
    float* d_grad = nullptr;
    float* d_scalars = nullptr;  // [0..1] adv mean / rstd, [2..3] grad norm / clip scale
};

namespace qr {
int set_last_error(int code, const std::string& msg);  // quadrace_abi.hip
const half8* ppo_policy_image(const qr_ppo* p) { return p ? p->d_images : nullptr; }
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
    static int grad(qr_ppo* p, qr::PpoBatch b, hipStream_t st) {
        const size_t lds = (size_t)D::kImage * 16;
        static bool configured = false;
        if (!configured) {
            PPO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(qr::ppo_phase_a_kernel<L>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = true;
        }
        hipLaunchKernelGGL(qr::ppo_adv_stats_kernel, dim3(1), dim3(1024), 0, st, b.adv, b.idx, b.B, p->d_scalars);
        hipLaunchKernelGGL(qr::ppo_phase_a_kernel<L>, dim3((b.G + 3) / 4, 2), dim3(qr::kPpoBlock), lds, st, b);
        // split-K over the minibatch: enough waves to fill the chip (2 * kJobsPerNet tiles x chunks)
        int chunks = 2048 / (2 * D::kJobsPerNet);
        if (chunks > b.G) chunks = b.G;
        if (chunks < 1) chunks = 1;
        const int per = (b.G + chunks - 1) / chunks;
        chunks = (b.G + per - 1) / per;
        hipLaunchKernelGGL(qr::ppo_phase_b_kernel<L>, dim3(2 * D::kJobsPerNet, chunks), dim3(64), 0, st, p->d_tbuf, p->d_grad, b.G,
                           per, 1.0f / (float)b.B);
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
    if (B < 64 || B % 64 != 0 || B > p->max_B) return ppofail(QR_E_INVALID, "qr_ppo: minibatch size must be a multiple of 64 within max_minibatch");
    b.obs = obs; b.act = act; b.old_logp = old_logp; b.adv = adv; b.ret = ret; b.idx = idx;
    b.B = B; b.G = B / 64;
    b.clip = clip; b.vf_coef = vf_coef; b.ent_coef = ent_coef;
    b.adv_stats = p->d_scalars;
    b.theta = theta;
    b.images = p->d_images;
    b.tbuf = p->d_tbuf;
    b.grad = p->d_grad;
    b.stats = stats;
    return QR_OK;
}

}  // namespace

extern "C" {

int qr_ppo_create(int32_t obs_len, int32_t device, int32_t max_minibatch, qr_ppo** out) {
    if (!out) return ppofail(QR_E_INVALID, "qr_ppo_create: null output");
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
        p->slots = D::kSlots;
        p->jobs_per_net = D::kJobsPerNet;
        return (int)QR_OK;
    });
    if (rc != QR_OK) { delete p; return rc; }
    p->num_params = qr::ppo_num_params(obs_len);
    PPO_HIP(hipSetDevice(device));
    const size_t tbytes = (size_t)2 * p->slots * (max_minibatch / 64) * 256 * 16;
    hipError_t e = hipMalloc((void**)&p->d_images, (size_t)2 * p->image_half8 * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_tbuf, tbytes);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_grad, (size_t)p->num_params * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_scalars, 16 * 4);
    if (e == hipSuccess) e = hipMemset(p->d_grad, 0, (size_t)p->num_params * 4);
    if (e == hipSuccess) e = hipMemset(p->d_scalars, 0, 16 * 4);
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
    (void)hipFree(p->d_tbuf);
    (void)hipFree(p->d_grad);
    (void)hipFree(p->d_scalars);
    delete p;
    return QR_OK;
}

int qr_ppo_num_params(const qr_ppo* p) { return p ? p->num_params : ppofail(QR_E_INVALID, "qr_ppo_num_params: null handle"); }

int qr_ppo_pack(qr_ppo* p, const float* theta_dev, void* stream) {
    if (!p || !theta_dev) return ppofail(QR_E_INVALID, "qr_ppo_pack: null argument");
    PPO_HIP(hipSetDevice(p->device));
    return dispatch_L(p->L, [&](auto Lc) { return PpoOps<decltype(Lc)::value>::pack(p, theta_dev, (hipStream_t)stream); });
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
    PPO_HIP(hipMemsetAsync(p->d_grad, 0, (size_t)p->num_params * 4, st));
    if (int rc = dispatch_L(p->L, [&](auto Lc) {
            constexpr int L = decltype(Lc)::value;
            if (int r = PpoOps<L>::pack(p, theta_dev, st)) return r;
            return PpoOps<L>::grad(p, b, st);
        }))
        return rc;
    PPO_HIP(hipMemcpyAsync(grad_out_dev, p->d_grad, (size_t)p->num_params * 4, hipMemcpyDeviceToDevice, st));
    PPO_HIP(hipMemsetAsync(p->d_grad, 0, (size_t)p->num_params * 4, st));
    return QR_OK;
}

int qr_ppo_minibatch(qr_ppo* p, float* theta_dev, float* adam_m_dev, float* adam_v_dev, const float* obs_dev, const float* act_dev,
                     const float* old_logp_dev, const float* adv_dev, const float* ret_dev, const int32_t* idx_dev, int32_t B,
                     float clip, float vf_coef, float ent_coef, float max_grad_norm, float lr, float beta1, float beta2, float eps,
                     int32_t adam_step, float* stats_dev, void* stream) {
    qr::PpoBatch b;
    if (int rc = fill_batch(p, b, theta_dev, obs_dev, act_dev, old_logp_dev, adv_dev, ret_dev, idx_dev, B, clip, vf_coef, ent_coef, stats_dev))
        return rc;
    if (!adam_m_dev || !adam_v_dev || adam_step < 1) return ppofail(QR_E_INVALID, "qr_ppo_minibatch: bad Adam state");
    PPO_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const float bc1 = 1.0f - powf(beta1, (float)adam_step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)adam_step));
    return dispatch_L(p->L, [&](auto Lc) {
        constexpr int L = decltype(Lc)::value;
        if (int r = PpoOps<L>::grad(p, b, st)) return r;  // images were packed by the previous call (or qr_ppo_pack)
        hipLaunchKernelGGL(qr::ppo_norm_kernel, dim3(1), dim3(1024), 0, st, p->d_grad, p->num_params, max_grad_norm, p->d_scalars + 2);
        hipLaunchKernelGGL(qr::ppo_adam_kernel, dim3((p->num_params + 255) / 256), dim3(256), 0, st, theta_dev, adam_m_dev, adam_v_dev,
                           p->d_grad, p->num_params, p->d_scalars + 2, lr, beta1, beta2, eps, bc1, bc2s);
        PPO_HIP(hipGetLastError());
        return PpoOps<L>::pack(p, theta_dev, st);
    });
}

}  // extern "C"
