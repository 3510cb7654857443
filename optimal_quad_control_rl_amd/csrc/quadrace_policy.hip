// quadrace_policy.hip -- policy-network kernels (see quadrace_policy.hpp) and their host side.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/quadrace.h"
#include "quadrace_policy.hpp"

namespace qr {

constexpr int kPolBlock = 256;

// cooperative copy of the packed f16 weights (16-byte elements) into LDS
__device__ __forceinline__ void stage_policy(const half8* __restrict__ src, half8* dst, int count) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < count; i += kPolBlock) d4[i] = s4[i];
}

// obs [n][L] row-major -> mean actions [n][4]
template <int L>
__global__ void __launch_bounds__(kPolBlock, 1)
policy_kernel(const half8* __restrict__ weights, int n, const float* __restrict__ obs, float4* __restrict__ mean_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* W = reinterpret_cast<half8*>(smem);
    const int i = blockIdx.x * kPolBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int ii = i < n ? i : 0;  // ragged-tail lanes shadow env 0 (MFMA / swaps are wave-wide)
    float o[L];
    const float* row = obs + (size_t)ii * L;
    if constexpr (L % 4 == 0) {
#pragma unroll
        for (int k = 0; k < L / 4; ++k) {
            const float4 v = reinterpret_cast<const float4*>(row)[k];
            o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < L; ++k) o[k] = row[k];
    }
    stage_policy(weights, W, PolicyDims<L>::kTotalHalf8);
    __syncthreads();
    float mean[4];
    policy_forward<L>(W, lane, o, mean);
    if (i < n) mean_out[i] = make_float4(mean[0], mean[1], mean[2], mean[3]);
}

template <int L>
hipError_t launch_policy_L(const half8* w, int n, const float* obs, float* mean, hipStream_t st) {
    const size_t lds = (size_t)PolicyDims<L>::kTotalHalf8 * 16;
    static unsigned long long configured = 0;   // per device ordinal
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(policy_kernel<L>), lds, configured)) return e;
    hipLaunchKernelGGL(policy_kernel<L>, dim3((n + kPolBlock - 1) / kPolBlock), dim3(kPolBlock), lds, st, w, n, obs,
                       reinterpret_cast<float4*>(mean));
    return hipGetLastError();
}

hipError_t launch_policy(int L, const half8* w, int n, const float* obs, float* mean, hipStream_t st) {
    switch (L) {  // every observation length the two env variants can produce (gates_ahead 0..4)
        case 13: return launch_policy_L<13>(w, n, obs, mean, st);
        case 17: return launch_policy_L<17>(w, n, obs, mean, st);
        case 21: return launch_policy_L<21>(w, n, obs, mean, st);
        case 25: return launch_policy_L<25>(w, n, obs, mean, st);
        case 29: return launch_policy_L<29>(w, n, obs, mean, st);
        case 20: return launch_policy_L<20>(w, n, obs, mean, st);
        case 24: return launch_policy_L<24>(w, n, obs, mean, st);
        case 28: return launch_policy_L<28>(w, n, obs, mean, st);
        case 32: return launch_policy_L<32>(w, n, obs, mean, st);
        case 36: return launch_policy_L<36>(w, n, obs, mean, st);
        default: return hipErrorInvalidValue;
    }
}

template <int L>
__global__ void __launch_bounds__(kPolBlock, 1)
policy_f32class_kernel(const half8* __restrict__ w0, const half8* __restrict__ w1, int n, const float* __restrict__ obs,
                       float4* __restrict__ mean_out) {
    using D = PolicyDims<L>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* W = reinterpret_cast<half8*>(smem);
    const int i = blockIdx.x * kPolBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int ii = i < n ? i : 0;  // ragged-tail lanes shadow env 0 (MFMA / swaps are wave-wide)
    float o[L];
    const float* row = obs + (size_t)ii * L;
#pragma unroll
    for (int k = 0; k < L; ++k) o[k] = row[k];
    stage_policy(w0, W, D::kTotalHalf8);
    __syncthreads();
    float mean[4];
    policy_forward_f32class<L>(W, w1, lane, o, mean);
    if (i < n) mean_out[i] = make_float4(mean[0], mean[1], mean[2], mean[3]);
}

template <int L>
hipError_t launch_policy_f32class_L(const half8* w0, const half8* w1, int n, const float* obs, float* mean, hipStream_t st) {
    const size_t lds = (size_t)PolicyDims<L>::kTotalHalf8 * 16;
    static unsigned long long configured = 0;   // per device ordinal
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(policy_f32class_kernel<L>), lds, configured)) return e;
    hipLaunchKernelGGL(policy_f32class_kernel<L>, dim3((n + kPolBlock - 1) / kPolBlock), dim3(kPolBlock), lds, st, w0, w1, n, obs,
                       reinterpret_cast<float4*>(mean));
    return hipGetLastError();
}

hipError_t launch_policy_f32class(int L, const half8* w0, const half8* w1, int n, const float* obs, float* mean, hipStream_t st) {
    switch (L) {
        case 13: return launch_policy_f32class_L<13>(w0, w1, n, obs, mean, st);
        case 17: return launch_policy_f32class_L<17>(w0, w1, n, obs, mean, st);
        case 21: return launch_policy_f32class_L<21>(w0, w1, n, obs, mean, st);
        case 25: return launch_policy_f32class_L<25>(w0, w1, n, obs, mean, st);
        case 29: return launch_policy_f32class_L<29>(w0, w1, n, obs, mean, st);
        case 20: return launch_policy_f32class_L<20>(w0, w1, n, obs, mean, st);
        case 24: return launch_policy_f32class_L<24>(w0, w1, n, obs, mean, st);
        case 28: return launch_policy_f32class_L<28>(w0, w1, n, obs, mean, st);
        case 32: return launch_policy_f32class_L<32>(w0, w1, n, obs, mean, st);
        case 36: return launch_policy_f32class_L<36>(w0, w1, n, obs, mean, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace qr

struct qr_policy {
    int L = 0, device = 0;
    int steps1 = 0;
    size_t total_half8 = 0;
    qr::half8* d_weights = nullptr;      // W0 = f16(w): the image of the f16-operand kernels
    qr::half8* d_weights_lo = nullptr;   // W1 = f16(w - W0): the low pieces, same layout (f32-class forward only)
    bool has_weights = false;
};

namespace qr {
int set_last_error(int code, const std::string& msg);  // quadrace_abi.hip
}
namespace {
int pfail(int code, const std::string& m) { return qr::set_last_error(code, m); }
inline int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
}  // namespace

namespace qr {  // accessors for the closed-loop rollout entry point in quadrace_abi.hip
const half8* policy_weights(const qr_policy* p) { return (p && p->has_weights) ? p->d_weights : nullptr; }
const half8* policy_weights_lo(const qr_policy* p) { return (p && p->has_weights) ? p->d_weights_lo : nullptr; }
int policy_obs_len(const qr_policy* p) { return p ? p->L : -1; }
int policy_device(const qr_policy* p) { return p ? p->device : -1; }
}  // namespace qr

extern "C" {

const char* qr_policy_last_error(void) { return qr_last_error(); }  // same thread-local message as the env calls

int qr_policy_create(int32_t obs_len, int32_t device, qr_policy** out) {
    if (!out) return pfail(QR_E_INVALID, "qr_policy_create: null output");
    *out = nullptr;
    static const int ok[] = {13, 17, 21, 25, 29, 20, 24, 28, 32, 36};
    bool found = false;
    for (int v : ok) found |= (v == obs_len);
    if (!found) return pfail(QR_E_INVALID, "qr_policy_create: obs_len must be an observation length of the race envs");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return pfail(QR_E_NO_DEVICE, "qr_policy_create: no HIP device visible (no CPU fallback)");
    if (device < 0 || device >= ndev) return pfail(QR_E_INVALID, "qr_policy_create: bad device ordinal");
    if (hipSetDevice(device) != hipSuccess) return pfail(QR_E_HIP, "qr_policy_create: hipSetDevice failed");
    qr_policy* p = new qr_policy();
    p->L = obs_len;
    p->device = device;
    p->steps1 = (obs_len + 1 + 15) / 16;
    p->total_half8 = (size_t)4 * p->steps1 * 64 + 2 * 4 * 8 * 64 + 8 * 64;
    if (hipMalloc((void**)&p->d_weights, p->total_half8 * 16) != hipSuccess ||
        hipMalloc((void**)&p->d_weights_lo, p->total_half8 * 16) != hipSuccess) {
        if (p->d_weights) (void)hipFree(p->d_weights);
        delete p;
        return pfail(QR_E_HIP, "qr_policy_create: hipMalloc failed");
    }
    *out = p;
    return QR_OK;
}

int qr_policy_destroy(qr_policy* p) {
    if (!p) return QR_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    if (p->d_weights) (void)hipFree(p->d_weights);
    if (p->d_weights_lo) (void)hipFree(p->d_weights_lo);
    delete p;
    return QR_OK;
}

// torch.nn.Linear layout: w[out][in], b[out]
int qr_policy_set_weights(qr_policy* p, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* w3, const float* b3, const float* w4, const float* b4) {
    if (!p || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4)
        return pfail(QR_E_INVALID, "qr_policy_set_weights: null argument");
    const int L = p->L, H = qr::kPolHidden, HP = qr::kPolHiddenPad, BU = qr::kPolBiasUnit;
    std::vector<__half> img(p->total_half8 * 8, __float2half(0.0f)), img_lo(p->total_half8 * 8, __float2half(0.0f));
    // every packed value as two f16 pieces: W0 = f16(w) (the f16-operand image), W1 = f16(w - W0) (exact difference in f32; the image
    // of the low pieces for the f32-class forward).  A weight beyond the f16 range saturates in W0 and W1 carries the rest.
    auto put = [&](size_t idx, float w) {
        float c = w;
        if (c > 65504.0f) c = 65504.0f;
        if (c < -65504.0f) c = -65504.0f;
        const __half h0 = __float2half(c);
        img[idx] = __float2half(w);   // (unchanged behaviour of the f16 image: inf beyond the range, as before)
        float r = w - __half2float(h0);
        if (r > 65504.0f) r = 65504.0f;
        if (r < -65504.0f) r = -65504.0f;
        img_lo[idx] = (w == w) ? __float2half(r) : __float2half(0.0f);
    };
    // padded weight accessors incl. the bias column and the constant-1 unit
    auto W1 = [&](int row, int k) -> float {  // row < 128, k < 16*steps1 ; input k == L is the constant 1
        if (row < H) return k < L ? w1[row * L + k] : (k == L ? b1[row] : 0.0f);
        return (row == BU && k == L) ? 1.0f : 0.0f;
    };
    auto WH = [&](const float* w, const float* b, int row, int hid) -> float {  // hidden layers: in = hidden units
        if (row < H) return hid < H ? w[row * H + hid] : (hid == BU ? b[row] : 0.0f);
        return (row == BU && hid == BU) ? 1.0f : 0.0f;
    };
    auto W4 = [&](int row, int hid) -> float {  // 4 output rows in a 32-row tile
        if (row < 4) return hid < H ? w4[row * H + hid] : (hid == BU ? b4[row] : 0.0f);
        return 0.0f;
    };
    size_t e = 0;  // half8 element index
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < p->steps1; ++s)
            for (int l = 0; l < 64; ++l, ++e)
                for (int j = 0; j < 8; ++j) put(e * 8 + j, W1(32 * t + (l & 31), 16 * s + 8 * (l >> 5) + j));
    for (int layer = 0; layer < 2; ++layer) {
        const float* w = layer == 0 ? w2 : w3;
        const float* b = layer == 0 ? b2 : b3;
        for (int t = 0; t < 4; ++t)
            for (int sp = 0; sp < 8; ++sp)
                for (int l = 0; l < 64; ++l, ++e)
                    for (int j = 0; j < 8; ++j) {
                        const int hid = 32 * (sp >> 1) + rho(8 * (sp & 1) + j, l >> 5);
                        put(e * 8 + j, WH(w, b, 32 * t + (l & 31), hid));
                    }
    }
    for (int sp = 0; sp < 8; ++sp)
        for (int l = 0; l < 64; ++l, ++e)
            for (int j = 0; j < 8; ++j) {
                const int hid = 32 * (sp >> 1) + rho(8 * (sp & 1) + j, l >> 5);
                put(e * 8 + j, W4(l & 31, hid));
            }
    (void)HP;
    if (e != p->total_half8) return pfail(QR_E_STATE, "qr_policy_set_weights: internal packing size mismatch");
    if (hipSetDevice(p->device) != hipSuccess) return pfail(QR_E_HIP, "hipSetDevice failed");
    if (hipDeviceSynchronize() != hipSuccess) return pfail(QR_E_HIP, "hipDeviceSynchronize failed");
    if (hipMemcpy(p->d_weights, img.data(), img.size() * sizeof(__half), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(p->d_weights_lo, img_lo.data(), img_lo.size() * sizeof(__half), hipMemcpyHostToDevice) != hipSuccess)
        return pfail(QR_E_HIP, "qr_policy_set_weights: upload failed");
    p->has_weights = true;
    return QR_OK;
}

int qr_policy_forward(qr_policy* p, int32_t n, const float* obs_dev, float* mean_out_dev, void* stream) {
    if (!p || !obs_dev || !mean_out_dev || n < 1) return pfail(QR_E_INVALID, "qr_policy_forward: bad argument");
    if (!p->has_weights) return pfail(QR_E_STATE, "qr_policy_forward: qr_policy_set_weights has not been called");
    hipError_t e = qr::launch_policy(p->L, p->d_weights, n, obs_dev, mean_out_dev, (hipStream_t)stream);
    if (e != hipSuccess) return pfail(QR_E_HIP, std::string("qr_policy_forward: ") + hipGetErrorString(e));
    return QR_OK;
}

int qr_policy_forward_f32class(qr_policy* p, int32_t n, const float* obs_dev, float* mean_out_dev, void* stream) {
    if (!p || !obs_dev || !mean_out_dev || n < 1) return pfail(QR_E_INVALID, "qr_policy_forward_f32class: bad argument");
    if (!p->has_weights) return pfail(QR_E_STATE, "qr_policy_forward_f32class: qr_policy_set_weights has not been called");
    hipError_t e = qr::launch_policy_f32class(p->L, p->d_weights, p->d_weights_lo, n, obs_dev, mean_out_dev, (hipStream_t)stream);
    if (e != hipSuccess) return pfail(QR_E_HIP, std::string("qr_policy_forward_f32class: ") + hipGetErrorString(e));
    return QR_OK;
}

}  // extern "C"
