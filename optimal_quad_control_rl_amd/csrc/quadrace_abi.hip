// quadrace_abi.hip -- host side of libquadrace.so: the extern "C" boundary declared in include/quadrace.h.
// Owns the planar HBM state of one env shard, the constant tables, and enqueues the gfx950 kernels on the
// caller's stream.  There is deliberately no CPU execution path: without a gfx950 device qr_create fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/quadrace.h"
#include "quadrace_device.hpp"
#include "quadrace_policy.hpp"

namespace qr {
hipError_t launch_step(int variant, const Params& P, const float* actions, float* obs, float* rew, uint8_t* done,
                       uint8_t* trunc, hipStream_t st);
const char* rollout_kernel_name(int variant, const Params& P, int form);
hipError_t launch_rollout(int variant, const Params& P, int form, int K, const float* actions, float* obs, float* rew,
                          uint8_t* done, uint8_t* trunc, hipStream_t st);
hipError_t launch_rollout_policy(int variant, const Params& P, const PolicyArgs& A, int K, float* obs, float* act,
                                 float* logp, float* rew, uint8_t* done, uint8_t* trunc, float* last_obs,
                                 hipStream_t st);
const half8* policy_weights(const qr_policy* p);
const half8* policy_weights_lo(const qr_policy* p);
int policy_obs_len(const qr_policy* p);
int policy_device(const qr_policy* p);
hipError_t launch_reset(int variant, const Params& P, const uint8_t* mask, float* obs, hipStream_t st);
hipError_t launch_observe(int variant, const Params& P, float* obs, hipStream_t st);
hipError_t launch_clear_episode(const Params& P, hipStream_t st);
hipError_t launch_residual_probe(const Params& P, float* out, hipStream_t st);
hipError_t launch_get_state(int variant, const Params& P, float* world, float* dist, int32_t* target, int32_t* steps,
                            uint32_t* episode, hipStream_t st);
hipError_t launch_set_state(int variant, const Params& P, const float* world, const float* dist,
                            const int32_t* target, const int32_t* steps, const uint32_t* episode, hipStream_t st);
}  // namespace qr

struct qr_env {
    qr_config cfg{};
    int S = 0, L = 0;
    qr::Params P{};
    void* slab = nullptr;       // one HBM allocation holding every state plane
    float* d_tables = nullptr;  // [gate rows | fused MLP table]
    int num_gates = 0;
    int rollout_form = 0;       // QR_ROLLOUT_AUTO | _MULTI_WAVE | _GENERAL (qr_set_rollout_form)
    int term_rows = 0;          // leading dimension of the registered terminal-observation buffer (K-step calls need K <= rows)
    bool has_track = false;
    std::vector<float> gate_pos, gate_yaw, gate_pos_rel, gate_yaw_rel;
    float mlp_table[qr::kMlpTableFloats] = {};
    // host-side configuration that only reaches the device through the tables
    float start[3] = {0.0f, 0.0f, 0.0f};
    float dist_lo[6] = {}, dist_hi[6] = {};
    float dist_scale = 1.0f;  // R:358
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_valid = false;
    bool timing = true;         // qr_set_timing: bracket the K-step calls with hipEvents (two marker packets per call)
    // qr_step_launches: the K step-kernel launches of one call, captured once into a hipGraph and replayed while the
    // arguments stay the same (measured, tools/ubench/launch_floor.hip: back-to-back dependent launches cost 2.6 us
    // each on a stream and 1.5 us as consecutive kernel nodes of a graph)
    struct StepGraph {
        hipGraphExec_t exec = nullptr;
        int K = 0;
        const void *act = nullptr, *obs = nullptr, *rew = nullptr, *done = nullptr, *trunc = nullptr;
        qr::Params P{};
    } sg;
    hipStream_t capture_stream = nullptr;
};

namespace {

// observation scaling of the constant disturbances (R:414-448): if min == max the range becomes (min-1, max+1)
void update_obs_scale(qr_env* e) {
    static const int col[4] = {0, 1, 2, 5};
    for (int c = 0; c < 4; ++c) {
        float lo = e->dist_lo[col[c]], hi = e->dist_hi[col[c]];
        if (lo == hi) { lo -= 1.0f; hi += 1.0f; }
        e->P.obs_lo[c] = lo;
        e->P.obs_inv[c] = 1.0f / (hi - lo);
    }
}

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace

namespace qr {
// shared with quadrace_policy.hip: one thread-local error string behind qr_last_error()
int set_last_error(int code, const std::string& msg) { return fail(code, msg); }
}  // namespace qr

namespace {

#define QR_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(QR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int upload_tables(qr_env* e) {
    const int gate_floats = e->num_gates * qr::kGateStride;
    // device image: [MLP table | reset table | gate rows]
    std::vector<float> host(qr::kOffGatesImage + gate_floats, 0.0f);
    std::memcpy(host.data(), e->mlp_table, sizeof(e->mlp_table));
    // reset table rows (lo, hi - lo, add, mul): value = ((lo + (hi-lo)*u) + add) * mul   (R:455-489 / I:270-296)
    float* R = host.data() + qr::kOffResetImage;
    auto row = [&](int t, float lo, float hi, float add, float mul) {
        volatile float span = hi - lo;  // float32 subtraction, like the oracle / the reference's ranges
        R[4 * t + 0] = lo; R[4 * t + 1] = span; R[4 * t + 2] = add; R[4 * t + 3] = mul;
    };
    const float pi9 = 0.3490658503988659f, pi = 3.141592653589793f;
    for (int t = 0; t < 3; ++t) row(t, -0.5f, 0.5f, e->start[t], 1.0f);
    for (int t = 3; t < 6; ++t) row(t, -0.5f, 0.5f, 0.0f, 1.0f);
    row(6, -pi9, pi9, 0.0f, 1.0f);
    row(7, -pi9, pi9, 0.0f, 1.0f);
    row(8, -pi, pi, 0.0f, 1.0f);
    for (int t = 9; t < 12; ++t) row(t, -0.1f, 0.1f, 0.0f, 1.0f);
    if (e->cfg.variant == QR_VARIANT_E2E) {
        for (int t = 12; t < 16; ++t) row(t, -1.0f, 1.0f, 0.0f, 1.0f);
        for (int k = 0; k < 6; ++k) row(16 + k, e->dist_lo[k], e->dist_hi[k], 0.0f, e->dist_scale);
    } else {
        row(12, -0.1f, 0.1f, 0.0f, 1.0f);
    }
    for (int g = 0; g < e->num_gates; ++g) {
        float* grow = host.data() + qr::kOffGatesImage + g * qr::kGateStride;
        grow[0] = e->gate_pos[3 * g + 0];
        grow[1] = e->gate_pos[3 * g + 1];
        grow[2] = e->gate_pos[3 * g + 2];
        grow[3] = e->gate_yaw[g];
        grow[4] = cosf(e->gate_yaw[g]);  // the reference evaluates np.cos/np.sin on the f32 yaw every step (R:372-375,528)
        grow[5] = sinf(e->gate_yaw[g]);
        grow[8] = e->gate_pos_rel[3 * g + 0];
        grow[9] = e->gate_pos_rel[3 * g + 1];
        grow[10] = e->gate_pos_rel[3 * g + 2];
        grow[11] = e->gate_yaw_rel[g];
    }
    QR_HIP(hipDeviceSynchronize());  // configuration setters are rare: do not race kernels still reading the table
    QR_HIP(hipMemcpy(e->d_tables, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return QR_OK;
}

// A host may hold handles on several GPUs in one process: every entry point that touches the device first makes the
// handle's GPU current (kernels launch on the CURRENT device; a stream of another device would be rejected).
int bind_device(const qr_env* e) {
    if (!e) return fail(QR_E_INVALID, "null env handle");
    QR_HIP(hipSetDevice(e->cfg.device));
    return QR_OK;
}

int check_ready(const qr_env* e) {
    if (int rc = bind_device(e)) return rc;
    if (!e->has_track) return fail(QR_E_STATE, "qr_set_track has not been called");
    return QR_OK;
}

// hipEvent bracket of a K-step call: skipped when switched off (qr_set_timing) or while the caller captures `st` into a graph
bool want_events(const qr_env* e, hipStream_t st) {
    if (!e->timing) return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return false;
    return true;
}

// K-step entry points write terminal observations to row [k][env]: the registered buffer must hold K rows of N envs
int check_term_rows(const qr_env* e, int K, const char* who) {
    if (e->P.term_obs && K > e->term_rows)
        return fail(QR_E_INVALID, std::string(who) + ": num_steps exceeds the rows of the registered terminal-observation buffer "
                                  "(qr_set_terminal_obs); register a [K][N][obs_len] buffer or NULL first");
    return QR_OK;
}

}  // namespace

extern "C" {

int qr_abi_version(void) { return QR_ABI_VERSION; }
const char* qr_last_error(void) { return g_err.c_str(); }

int qr_create(const qr_config* cfg, qr_env** out) {
    if (!cfg || !out) return fail(QR_E_INVALID, "qr_create: null argument");
    *out = nullptr;
    if (cfg->variant != QR_VARIANT_E2E && cfg->variant != QR_VARIANT_INDI)
        return fail(QR_E_INVALID, "qr_create: unknown variant");
    if (cfg->num_envs < 1) return fail(QR_E_INVALID, "qr_create: num_envs must be >= 1");
    if (cfg->gates_ahead < 0 || cfg->gates_ahead > QR_MAX_GATES_AHEAD)
        return fail(QR_E_INVALID, "qr_create: gates_ahead out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(QR_E_NO_DEVICE, "qr_create: no HIP device visible (libquadrace has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(QR_E_INVALID, "qr_create: bad device ordinal");
    hipDeviceProp_t prop;
    QR_HIP(hipGetDeviceProperties(&prop, cfg->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(QR_E_NO_DEVICE, std::string("qr_create: device is ") + prop.gcnArchName + ", kernels are gfx950-only");
    QR_HIP(hipSetDevice(cfg->device));

    qr_env* e = new qr_env();
    e->cfg = *cfg;
    const bool e2e = cfg->variant == QR_VARIANT_E2E;
    e->S = e2e ? 16 : 13;
    e->L = e->S + 4 * cfg->gates_ahead + (e2e ? 4 : 0);  // R:330 / I:185
    const int n = cfg->num_envs;
    const size_t ns = align_up((size_t)n, qr::kBlock);
    // slab layout (every region 256-byte aligned)
    const size_t sz_ws = align_up(sizeof(float4) * ns * (e2e ? 4 : 3), 256);
    const size_t sz_tn = e2e ? 0 : align_up(sizeof(float) * ns, 256);
    const size_t sz_dA = e2e ? align_up(sizeof(float4) * ns, 256) : 0;
    const size_t sz_dB = e2e ? align_up(sizeof(float2) * ns, 256) : 0;
    const size_t sz_ts = align_up(sizeof(int2) * ns, 256);
    const size_t total = sz_ws + sz_tn + sz_dA + sz_dB + sz_ts;
    if (hipMalloc(&e->slab, total) != hipSuccess) {
        delete e;
        return fail(QR_E_HIP, "qr_create: hipMalloc of the state slab failed");
    }
    if (hipMalloc((void**)&e->d_tables, sizeof(float) * (qr::kOffGatesImage + qr::kMaxGates * qr::kGateStride)) != hipSuccess) {
        (void)hipFree(e->slab);
        delete e;
        return fail(QR_E_HIP, "qr_create: hipMalloc of the table buffer failed");
    }
    (void)hipMemset(e->slab, 0, total);
    char* p = static_cast<char*>(e->slab);
    qr::Params& P = e->P;
    P.ws = reinterpret_cast<float4*>(p); p += sz_ws;
    P.tn = e2e ? nullptr : reinterpret_cast<float*>(p); p += sz_tn;
    P.dA = e2e ? reinterpret_cast<float4*>(p) : nullptr; p += sz_dA;
    P.dB = e2e ? reinterpret_cast<float2*>(p) : nullptr; p += sz_dB;
    P.ts = reinterpret_cast<int2*>(p);
    P.tables = e->d_tables;
    P.n = n;
    P.n_stride = (int)ns;
    P.num_gates = 0;
    P.gates_ahead = cfg->gates_ahead;
    P.max_steps = 1200;  // R:345
    P.dt = 0.01f;        // R:346
    P.flags = cfg->pause_if_collision ? qr::kFlagPauseIfCollision : 0;
    P.seed_lo = P.seed_hi = 0;
    P.gid_lo = (uint32_t)cfg->env_id_base;
    P.gid_hi = (uint32_t)(cfg->env_id_base >> 32);
    update_obs_scale(e);
    (void)hipEventCreate(&e->ev0);
    (void)hipEventCreate(&e->ev1);
    *out = e;
    return QR_OK;
}

int qr_destroy(qr_env* e) {
    if (!e) return QR_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->sg.exec) (void)hipGraphExecDestroy(e->sg.exec);
    if (e->capture_stream) (void)hipStreamDestroy(e->capture_stream);
    if (e->slab) (void)hipFree(e->slab);
    if (e->d_tables) (void)hipFree(e->d_tables);
    delete e;
    return QR_OK;
}

int qr_state_len(const qr_env* e) { return e ? e->S : QR_E_INVALID; }
int qr_obs_len(const qr_env* e) { return e ? e->L : QR_E_INVALID; }
int qr_num_envs(const qr_env* e) { return e ? e->cfg.num_envs : QR_E_INVALID; }

int qr_set_track(qr_env* e, const float* gate_pos, const float* gate_yaw, int32_t G, const float* start_pos) {
    if (!e || !gate_pos || !gate_yaw || !start_pos) return fail(QR_E_INVALID, "qr_set_track: null argument");
    if (G < 1 || G > QR_MAX_GATES) return fail(QR_E_INVALID, "qr_set_track: num_gates must be in 1..QR_MAX_GATES");
    QR_HIP(hipSetDevice(e->cfg.device));
    e->num_gates = G;
    e->gate_pos.assign(gate_pos, gate_pos + 3 * G);
    e->gate_yaw.assign(gate_yaw, gate_yaw + G);
    e->gate_pos_rel.assign(3 * G, 0.0f);
    e->gate_yaw_rel.assign(G, 0.0f);
    // gate i expressed in the frame of gate i-1, looped track (R:307-319), float32 like the reference
    for (int i = 0; i < G; ++i) {
        const int j = (i + G - 1) % G;
        const float dx = gate_pos[3 * i + 0] - gate_pos[3 * j + 0];
        const float dy = gate_pos[3 * i + 1] - gate_pos[3 * j + 1];
        const float c = cosf(gate_yaw[j]), s = sinf(gate_yaw[j]);
        volatile float cx = c * dx, sy = s * dy, sx = -s * dx, cy = c * dy;  // no host FMA contraction
        e->gate_pos_rel[3 * i + 0] = cx + sy;
        e->gate_pos_rel[3 * i + 1] = sx + cy;
        e->gate_pos_rel[3 * i + 2] = gate_pos[3 * i + 2] - gate_pos[3 * j + 2];
        e->gate_yaw_rel[i] = gate_yaw[i] - gate_yaw[j];
    }
    for (int k = 0; k < 3; ++k) e->start[k] = start_pos[k];
    e->P.num_gates = G;
    e->has_track = true;
    return upload_tables(e);
}

int qr_get_track_tables(const qr_env* e, float* gate_pos_rel, float* gate_yaw_rel) {
    if (!e) return fail(QR_E_INVALID, "null env handle");
    if (!e->has_track) return fail(QR_E_STATE, "qr_set_track has not been called");
    if (gate_pos_rel) std::memcpy(gate_pos_rel, e->gate_pos_rel.data(), sizeof(float) * 3 * e->num_gates);
    if (gate_yaw_rel) std::memcpy(gate_yaw_rel, e->gate_yaw_rel.data(), sizeof(float) * e->num_gates);
    return QR_OK;
}

int qr_set_residual(qr_env* e, const float* blob, size_t n_floats) {
    if (!e) return fail(QR_E_INVALID, "qr_set_residual: null env");
    if (e->cfg.variant != QR_VARIANT_E2E) return fail(QR_E_INVALID, "qr_set_residual: E2E variant only");
    QR_HIP(hipSetDevice(e->cfg.device));
    if (!blob) {
        e->P.flags &= ~qr::kFlagResidual;
        return QR_OK;
    }
    if (n_floats != QR_RESIDUAL_FLOATS) return fail(QR_E_INVALID, "qr_set_residual: expected 740 floats");
    // layer 1 (W1, b1 of both networks) runs on f16 matrix-core operands, each weight as two f16 pieces: a weight beyond the f16 range
    // would turn into inf - inf = NaN pieces where the reference's float32 layer gives a finite value -- refused, not approximated.
    // Layer 2 (W2, b2) is float32 arithmetic like the reference's: any finite value is accepted there.
    for (size_t k = 0; k < n_floats; ++k) {
        const bool layer1 = k < 224 + 32 || (k >= 289 && k < 289 + 320 + 32);
        if (!std::isfinite(blob[k]))
            return fail(QR_E_INVALID, "qr_set_residual: weights must be finite");
        if (layer1 && std::fabs(blob[k]) > 65504.0f)
            return fail(QR_E_INVALID, "qr_set_residual: first-layer weights and biases must be within +-65504 (the f16 range of the matrix-core operands)");
    }
    // reference order (c_code/nn_thrust.c, nn_moment.c): W1[out][in], b1, W2[out][in], b2 per network
    const float* tW1 = blob;         const float* tb1 = tW1 + 224;
    const float* tW2 = tb1 + 32;     const float* tb2 = tW2 + 32;
    const float* mW1 = tb2 + 1;      const float* mb1 = mW1 + 320;
    const float* mW2 = mb1 + 32;     const float* mb2 = mW2 + 96;
    // device image: MFMA A operands, layer-2 weights per wave half, output biases (see quadrace_device.hpp)
    float* T = e->mlp_table;
    std::memset(T, 0, sizeof(e->mlp_table));
    // layer 1: every weight and bias as two f16 pieces w = W0 + W1 (round to nearest even; w - W0 is exact in float32), laid out as
    // the A operands of the five v_mfma_f32_32x32x16_f16 instructions per env tile (quadrace_device.hpp, "Residual-MLP table")
    auto piece0 = [](float w) { return (_Float16)w; };
    auto piece1 = [](float w) { return (_Float16)(w - (float)(_Float16)w); };
    _Float16* A = reinterpret_cast<_Float16*>(T + qr::kOffTabA);
    auto slot = [&](int q, int lane, int j) -> _Float16& { return A[((size_t)q * 64 + lane) * 8 + j]; };
    for (int row = 0; row < 32; ++row) {
        const int lo = row, hi = row + 32;   // lanes holding k-slots 0..7 / 8..15 of this hidden row
        for (int k = 0; k < 7; ++k) {
            slot(0, lo, k) = piece0(tW1[row * 7 + k]);   slot(0, hi, k) = piece0(tW1[row * 7 + k]);    // X0 W0 | X1 W0
            slot(1, lo, k) = piece1(tW1[row * 7 + k]);                                                  // X0 W1 | 0
            slot(2, lo, k) = piece0(mW1[row * 10 + k]);  slot(2, hi, k) = piece0(mW1[row * 10 + k]);
            slot(3, lo, k) = piece1(mW1[row * 10 + k]);
        }
        slot(0, lo, 7) = piece0(tb1[row]);  slot(1, lo, 7) = piece1(tb1[row]);   // k-slot 7 of the low half multiplies the constant 1
        slot(2, lo, 7) = piece0(mb1[row]);  slot(3, lo, 7) = piece1(mb1[row]);
        for (int k = 0; k < 3; ++k) {        // rates p, q, r: [X0 (3), 0, X1 (3), 0] in both halves
            slot(4, lo, k) = piece0(mW1[row * 10 + 7 + k]);  slot(4, lo, 4 + k) = piece0(mW1[row * 10 + 7 + k]);   // X0 W0, X1 W0
            slot(4, hi, k) = piece1(mW1[row * 10 + 7 + k]);                                                          // X0 W1
        }
    }
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            // x 2^40 (exact): the kernels' ReLU leaves max(x, 0) * 2^-40, see relu2_scaled() in quadrace_device.hpp
            T[qr::kOffTabW2 + 64 * h + r] = tW2[row] * qr::kReluUp;
            for (int m = 0; m < 3; ++m) T[qr::kOffTabW2 + 64 * h + 16 + 16 * m + r] = mW2[m * 32 + row] * qr::kReluUp;
        }
    T[qr::kOffB2 + 0] = tb2[0];
    for (int o = 0; o < 3; ++o) T[qr::kOffB2 + 1 + o] = mb2[o];
    e->P.flags |= qr::kFlagResidual;
    if (e->has_track) return upload_tables(e);
    return QR_OK;
}

int qr_set_disturbance(qr_env* e, const float* ranges, float scale) {
    if (!e || !ranges) return fail(QR_E_INVALID, "qr_set_disturbance: null argument");
    if (e->cfg.variant != QR_VARIANT_E2E) return fail(QR_E_INVALID, "qr_set_disturbance: E2E variant only");
    for (int k = 0; k < 6; ++k) {
        e->dist_lo[k] = ranges[2 * k + 0];
        e->dist_hi[k] = ranges[2 * k + 1];
    }
    e->dist_scale = scale;
    update_obs_scale(e);
    QR_HIP(hipSetDevice(e->cfg.device));
    return upload_tables(e);  // the reset table carries the disturbance ranges
}

int qr_set_limits(qr_env* e, int32_t max_steps, float dt) {
    if (!e) return fail(QR_E_INVALID, "qr_set_limits: null env");
    e->P.max_steps = max_steps;
    e->P.dt = dt;
    return QR_OK;
}

int qr_set_pause(qr_env* e, int32_t pause) {
    if (!e) return fail(QR_E_INVALID, "qr_set_pause: null env");
    if (pause) e->P.flags |= qr::kFlagPause; else e->P.flags &= ~qr::kFlagPause;
    return QR_OK;
}

int qr_set_pause_if_collision(qr_env* e, int32_t on) {
    if (!e) return fail(QR_E_INVALID, "qr_set_pause_if_collision: null env");
    e->cfg.pause_if_collision = on ? 1 : 0;
    if (on) e->P.flags |= qr::kFlagPauseIfCollision; else e->P.flags &= ~qr::kFlagPauseIfCollision;
    return QR_OK;
}

int qr_set_terminal_obs(qr_env* e, float* term_obs_dev, int32_t rows) {
    if (!e) return fail(QR_E_INVALID, "qr_set_terminal_obs: null env");
    if (term_obs_dev && rows < 1) return fail(QR_E_INVALID, "qr_set_terminal_obs: rows must be >= 1");
    e->P.term_obs = term_obs_dev;
    e->term_rows = term_obs_dev ? rows : 0;
    return QR_OK;
}

int qr_seed(qr_env* e, uint64_t seed) {
    if (!e) return fail(QR_E_INVALID, "qr_seed: null env");
    QR_HIP(hipSetDevice(e->cfg.device));
    e->P.seed_lo = (uint32_t)seed;
    e->P.seed_hi = (uint32_t)(seed >> 32);
    QR_HIP(qr::launch_clear_episode(e->P, nullptr));
    QR_HIP(hipStreamSynchronize(nullptr));
    return QR_OK;
}

int qr_reset(qr_env* e, const uint8_t* mask_dev, float* obs_out_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;
    QR_HIP(qr::launch_reset(e->cfg.variant, e->P, mask_dev, obs_out_dev, (hipStream_t)stream));
    return QR_OK;
}

int qr_step(qr_env* e, const float* actions_dev, float* obs_out_dev, float* rew_out_dev, uint8_t* done_out_dev,
            uint8_t* trunc_out_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;
    if (!actions_dev || !obs_out_dev || !rew_out_dev || !done_out_dev)
        return fail(QR_E_INVALID, "qr_step: actions/obs/rew/done buffers are required");
    QR_HIP(qr::launch_step(e->cfg.variant, e->P, actions_dev, obs_out_dev, rew_out_dev, done_out_dev, trunc_out_dev,
                           (hipStream_t)stream));
    return QR_OK;
}

int qr_step_many(qr_env* e, int32_t K, const float* actions_dev, float* obs_out_dev, float* rew_out_dev,
                 uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;
    if (K < 1) return fail(QR_E_INVALID, "qr_step_many: num_steps must be >= 1");
    if (int rc = check_term_rows(e, K, "qr_step_many")) return rc;
    if (!actions_dev || !obs_out_dev || !rew_out_dev || !done_out_dev)
        return fail(QR_E_INVALID, "qr_step_many: actions/obs/rew/done buffers are required");
    hipStream_t st = (hipStream_t)stream;
    const bool ev = want_events(e, st);
    if (ev) QR_HIP(hipEventRecord(e->ev0, st));
    // one launch: the fused rollout kernel keeps the env state in registers across the K steps
    QR_HIP(qr::launch_rollout(e->cfg.variant, e->P, e->rollout_form, K, actions_dev, obs_out_dev, rew_out_dev, done_out_dev,
                              trunc_out_dev, st));
    if (ev) QR_HIP(hipEventRecord(e->ev1, st));
    e->timing_valid = ev;
    return QR_OK;
}

int qr_step_launches(qr_env* e, int32_t K, const float* actions_dev, float* obs_out_dev, float* rew_out_dev,
                     uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;
    if (K < 1) return fail(QR_E_INVALID, "qr_step_launches: num_steps must be >= 1");
    if (int rc = check_term_rows(e, K, "qr_step_launches")) return rc;
    if (!actions_dev || !obs_out_dev || !rew_out_dev || !done_out_dev)
        return fail(QR_E_INVALID, "qr_step_launches: actions/obs/rew/done buffers are required");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)e->cfg.num_envs;
    // a caller that is itself capturing `stream` into a graph gets plain kernel nodes (a graph launch cannot be captured)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        for (int k = 0; k < K; ++k) {
            qr::Params Pk = e->P;
            if (Pk.term_obs) Pk.term_obs += (size_t)k * n * e->L;
            QR_HIP(qr::launch_step(e->cfg.variant, Pk, actions_dev + (size_t)k * n * 4, obs_out_dev + (size_t)k * n * e->L,
                                   rew_out_dev + (size_t)k * n, done_out_dev + (size_t)k * n,
                                   trunc_out_dev ? trunc_out_dev + (size_t)k * n : nullptr, st));
        }
        e->timing_valid = false;
        return QR_OK;
    }
    qr_env::StepGraph& g = e->sg;
    const bool hit = g.exec && g.K == K && g.act == actions_dev && g.obs == obs_out_dev && g.rew == rew_out_dev &&
                     g.done == done_out_dev && g.trunc == trunc_out_dev && std::memcmp(&g.P, &e->P, sizeof(e->P)) == 0;
    if (!hit) {  // (re)capture: K kernel nodes in a chain; kernel parameters (incl. Params) are baked into the nodes
        if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
        if (!e->capture_stream) QR_HIP(hipStreamCreateWithFlags(&e->capture_stream, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        QR_HIP(hipStreamBeginCapture(e->capture_stream, hipStreamCaptureModeThreadLocal));
        hipError_t err = hipSuccess;
        for (int k = 0; k < K && err == hipSuccess; ++k) {
            qr::Params Pk = e->P;
            if (Pk.term_obs) Pk.term_obs += (size_t)k * n * e->L;   // terminal-observation rows [k][env]
            err = qr::launch_step(e->cfg.variant, Pk, actions_dev + (size_t)k * n * 4, obs_out_dev + (size_t)k * n * e->L,
                                  rew_out_dev + (size_t)k * n, done_out_dev + (size_t)k * n,
                                  trunc_out_dev ? trunc_out_dev + (size_t)k * n : nullptr, e->capture_stream);
        }
        const hipError_t end = hipStreamEndCapture(e->capture_stream, &graph);
        if (err != hipSuccess || end != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            return fail(QR_E_HIP, std::string("qr_step_launches: graph capture failed: ") +
                                      hipGetErrorString(err != hipSuccess ? err : end));
        }
        const hipError_t inst = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (inst != hipSuccess) {
            g.exec = nullptr;
            return fail(QR_E_HIP, std::string("qr_step_launches: hipGraphInstantiate: ") + hipGetErrorString(inst));
        }
        g.K = K; g.act = actions_dev; g.obs = obs_out_dev; g.rew = rew_out_dev; g.done = done_out_dev; g.trunc = trunc_out_dev;
        std::memcpy(&g.P, &e->P, sizeof(e->P));  // byte copy: compared with memcmp above
    }
    const bool ev = want_events(e, st);
    if (ev) QR_HIP(hipEventRecord(e->ev0, st));
    QR_HIP(hipGraphLaunch(g.exec, st));
    if (ev) QR_HIP(hipEventRecord(e->ev1, st));
    e->timing_valid = ev;
    return QR_OK;
}

int qr_rollout_policy(qr_env* e, qr_policy* policy, int32_t K, const float* log_std, uint64_t noise_seed,
                      uint64_t first_step, int32_t deterministic, float* obs_out_dev, float* act_out_dev,
                      float* logp_out_dev, float* rew_out_dev, uint8_t* done_out_dev, uint8_t* trunc_out_dev,
                      float* last_obs_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;
    if (K < 1 || !policy || !log_std) return fail(QR_E_INVALID, "qr_rollout_policy: bad argument");
    if (int rc = check_term_rows(e, K, "qr_rollout_policy")) return rc;
    if (!obs_out_dev || !act_out_dev || !logp_out_dev || !rew_out_dev || !done_out_dev)
        return fail(QR_E_INVALID, "qr_rollout_policy: obs/act/logp/rew/done buffers are required");
    if (e->P.flags & (qr::kFlagPause | qr::kFlagPauseIfCollision))
        return fail(QR_E_STATE, "qr_rollout_policy: pause / pause_if_collision envs are evaluation modes; use qr_step");
    const qr::half8* w = qr::policy_weights(policy);
    if (!w) return fail(QR_E_STATE, "qr_rollout_policy: the policy has no weights");
    if (qr::policy_obs_len(policy) != e->L) return fail(QR_E_INVALID, "qr_rollout_policy: policy obs_len != env obs_len");
    if (qr::policy_device(policy) != e->cfg.device) return fail(QR_E_INVALID, "qr_rollout_policy: policy on another GPU");
    if (deterministic < 0 || deterministic > (QR_ROLLOUT_DETERMINISTIC | QR_ROLLOUT_F32CLASS))
        return fail(QR_E_INVALID, "qr_rollout_policy: `deterministic` takes QR_ROLLOUT_DETERMINISTIC | QR_ROLLOUT_F32CLASS");
    qr::PolicyArgs A{};
    A.weights = w;
    A.weights_lo = qr::policy_weights_lo(policy);
    A.f32class = (deterministic & QR_ROLLOUT_F32CLASS) ? 1 : 0;
    float sum_log_std = 0.0f;
    for (int c = 0; c < 4; ++c) {
        A.std[c] = expf(log_std[c]);
        sum_log_std += log_std[c];
    }
    A.logp_const = -sum_log_std - 2.0f * 1.8378770664093453f;  // 4 * 0.5 * log(2*pi)
    // Domain separation: the reset stream is Philox(counter = (env id, episode, block), key = env seed), the action
    // noise Philox(counter = (env id, step), key = noise seed).  With equal seeds the two would share random bits
    // whenever (episode, block) == (step lo, step hi); the noise key is therefore tweaked by a fixed odd constant.
    A.seed_lo = (uint32_t)noise_seed ^ 0x9E3779B9u;
    A.seed_hi = (uint32_t)(noise_seed >> 32) ^ 0x85EBCA6Bu;
    A.step_lo = (uint32_t)first_step;
    A.step_hi = (uint32_t)(first_step >> 32);
    A.deterministic = (deterministic & QR_ROLLOUT_DETERMINISTIC) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool ev = want_events(e, st);
    if (ev) QR_HIP(hipEventRecord(e->ev0, st));
    QR_HIP(qr::launch_rollout_policy(e->cfg.variant, e->P, A, K, obs_out_dev, act_out_dev, logp_out_dev, rew_out_dev,
                                     done_out_dev, trunc_out_dev, last_obs_dev, st));
    if (ev) QR_HIP(hipEventRecord(e->ev1, st));
    e->timing_valid = ev;
    return QR_OK;
}

int qr_observe(qr_env* e, float* obs_out_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;
    if (!obs_out_dev) return fail(QR_E_INVALID, "qr_observe: null output");
    QR_HIP(qr::launch_observe(e->cfg.variant, e->P, obs_out_dev, (hipStream_t)stream));
    return QR_OK;
}

int qr_probe_residual(qr_env* e, float* out_dev, void* stream) {
    if (int rc = bind_device(e)) return rc;
    if (!out_dev) return fail(QR_E_INVALID, "qr_probe_residual: null output");
    if (e->cfg.variant != QR_VARIANT_E2E || !(e->P.flags & qr::kFlagResidual))
        return fail(QR_E_STATE, "qr_probe_residual: needs an E2E env with residual weights");
    QR_HIP(qr::launch_residual_probe(e->P, out_dev, (hipStream_t)stream));
    return QR_OK;
}

int qr_get_state(qr_env* e, float* world_dev, float* dist_dev, int32_t* target_dev, int32_t* steps_dev,
                 uint32_t* episode_dev, void* stream) {
    if (int rc = bind_device(e)) return rc;
    QR_HIP(qr::launch_get_state(e->cfg.variant, e->P, world_dev, dist_dev, target_dev, steps_dev, episode_dev,
                                (hipStream_t)stream));
    return QR_OK;
}

int qr_set_state(qr_env* e, const float* world_dev, const float* dist_dev, const int32_t* target_dev,
                 const int32_t* steps_dev, const uint32_t* episode_dev, void* stream) {
    if (int rc = check_ready(e)) return rc;  // target is reduced modulo num_gates
    QR_HIP(qr::launch_set_state(e->cfg.variant, e->P, world_dev, dist_dev, target_dev, steps_dev, episode_dev,
                                (hipStream_t)stream));
    return QR_OK;
}

const char* qr_rollout_kernel_name(const qr_env* e) {
    if (!e) return "";
    return qr::rollout_kernel_name(e->cfg.variant, e->P, e->rollout_form);
}

int qr_set_rollout_form(qr_env* e, int32_t form) {
    if (!e) return fail(QR_E_INVALID, "qr_set_rollout_form: null env");
    if (form < 0 || form > (QR_ROLLOUT_MULTI_WAVE | QR_ROLLOUT_GENERAL | QR_ROLLOUT_ONE_WAVE)) return fail(QR_E_INVALID, "qr_set_rollout_form: unknown form");
    e->rollout_form = form;
    return QR_OK;
}

int qr_set_timing(qr_env* e, int32_t on) {
    if (!e) return fail(QR_E_INVALID, "qr_set_timing: null env");
    e->timing = on != 0;
    if (!e->timing) e->timing_valid = false;
    return QR_OK;
}

int qr_last_step_many_ms(qr_env* e, float* total_ms) {
    if (!e || !total_ms) return fail(QR_E_INVALID, "qr_last_step_many_ms: null argument");
    if (int rc = bind_device(e)) return rc;
    if (!e->timing_valid) return fail(QR_E_STATE, "qr_last_step_many_ms: no qr_step_many call recorded");
    QR_HIP(hipEventSynchronize(e->ev1));
    QR_HIP(hipEventElapsedTime(total_ms, e->ev0, e->ev1));
    return QR_OK;
}

int qr_profile_steps(qr_env* e, int32_t K, const float* actions_dev, float* obs_out_dev, float* rew_out_dev,
                     uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream, float* mean_kernel_ms,
                     float* region_ms) {
    if (int rc = check_ready(e)) return rc;
    if (K < 1 || !mean_kernel_ms || !region_ms) return fail(QR_E_INVALID, "qr_profile_steps: bad argument");
    if (int rc = check_term_rows(e, K, "qr_profile_steps")) return rc;
    if (!actions_dev || !obs_out_dev || !rew_out_dev || !done_out_dev)
        return fail(QR_E_INVALID, "qr_profile_steps: actions/obs/rew/done buffers are required");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)e->cfg.num_envs;
    std::vector<hipEvent_t> ev(2 * (size_t)K);
    for (auto& x : ev) QR_HIP(hipEventCreate(&x));
    for (int k = 0; k < K; ++k) {
        QR_HIP(hipEventRecord(ev[2 * k], st));
        qr::Params Pk = e->P;
        if (Pk.term_obs) Pk.term_obs += (size_t)k * n * e->L;   // terminal-observation rows [k][env]
        QR_HIP(qr::launch_step(e->cfg.variant, Pk, actions_dev + (size_t)k * n * 4, obs_out_dev + (size_t)k * n * e->L,
                               rew_out_dev + (size_t)k * n, done_out_dev + (size_t)k * n,
                               trunc_out_dev ? trunc_out_dev + (size_t)k * n : nullptr, st));
        QR_HIP(hipEventRecord(ev[2 * k + 1], st));
    }
    QR_HIP(hipEventSynchronize(ev[2 * K - 1]));
    double sum = 0.0;
    for (int k = 0; k < K; ++k) {
        float ms = 0.0f;
        QR_HIP(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        sum += ms;
    }
    QR_HIP(hipEventElapsedTime(region_ms, ev[0], ev[2 * K - 1]));
    *mean_kernel_ms = (float)(sum / K);
    for (auto& x : ev) (void)hipEventDestroy(x);
    return QR_OK;
}

#if defined(QR_PHASE_TIMING) || defined(QR_CLOCK_PROBE)
// profiling builds only (tools/phase_timing.py, tools/clock_probe.py): device buffer [n_waves][16] of shader-clock stamps, or NULL
__attribute__((visibility("default"))) int qr_debug_set_ticks(qr_env* e, unsigned long long* ticks_dev) {
    if (!e) return QR_E_INVALID;
    e->P.ticks = ticks_dev;
    e->P.tick_on = ticks_dev != nullptr;
    return QR_OK;
}
#endif

}  // extern "C"
