/*
 * quadrace_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, float32 CPU restatement of the reference's vectorised quadrotor race
 * environment (tudelft/optimal_quad_control_RL).  It exists to CHECK the HIP product path
 * (optimal_quad_control_rl_amd/csrc) and to provide the `cpu_baseline` leg of bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline may load it; the product never does.
 *
 * Parity pin: this file is validated against golden vectors generated from the REAL reference
 * (tools/gen_golden.py imports the notebooks; fixtures under tests/golden/, tests/test_oracle_*.py)
 * and against the reference's own artefacts: the residual-MLP known answer stored at R:184+
 * (thrust 36.098232, moment [0.2847767,-0.22512697,-0.05896095]), the relative-gate tables baked into
 * c_code/nn_controller.c:40-60, and the compiled c_code/nn_thrust.c / nn_moment.c (oracle/_ref).
 *
 * Citation tags: R: = "3D quad race.ipynb", I: = "3D quad race INDI inner loop.ipynb" (raw .ipynb line
 * numbers).  Arithmetic is float32 like the reference's NumPy arrays; python-float
 * constants are rounded to float32 first (NumPy weak-scalar rule).  Compile with -ffp-contract=off.
 *
 * The in-kernel reset RNG (Philox4x32-10 keyed by (seed, global env id, episode)) is NOT part of the
 * reference (which draws from NumPy's global MT19937, R:455-487); it is this build's specification,
 * restated here so that product and oracle can be compared bit-for-bit through resets.  The reset
 * *distributions* are the reference's.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QRO_MAX_GATES 32
#define QRO_E2E 0
#define QRO_INDI 1

typedef struct qro_env {
    int variant, n, gates_ahead, num_gates, pause_if_collision, pause, max_steps, has_residual;
    int state_len, obs_len;
    float dt;
    float gate_pos[QRO_MAX_GATES][3], gate_yaw[QRO_MAX_GATES];
    float gate_pos_rel[QRO_MAX_GATES][3], gate_yaw_rel[QRO_MAX_GATES];
    float start_pos[3];
    float dist_ranges[6][2];
    float dist_scale;
    float mlp[740];
    uint64_t seed, env_id_base;
    float *world, *dist, *obs;
    float* term_obs; /* optional caller buffer [n][obs_len]: pre-reset observation of envs that finish at a step */
    int32_t *target, *steps;
    uint32_t* episode;
    int threads; /* cpu_baseline only: OpenMP threads over the (independent) envs; 1 = scalar loop */
} qro_env;

/* ------------------------------------------------------------------------------------------------
 * Equations of motion
 * ---------------------------------------------------------------------------------------------- */

/* get_body_velocity, R:103, R:155: vb = R(phi,theta,psi)^T v */
void qro_body_velocity(const float* s, float* vb) {
    const float vx = s[3], vy = s[4], vz = s[5];
    const float sph = sinf(s[6]), cph = cosf(s[6]);
    const float sth = sinf(s[7]), cth = cosf(s[7]);
    const float sps = sinf(s[8]), cps = cosf(s[8]);
    vb[0] = vx * cps * cth + vy * sps * cth - vz * sth;
    vb[1] = vx * (sph * sth * cps - sps * cph) + vy * (sph * sps * sth + cph * cps) + vz * sph * cth;
    vb[2] = vx * (sph * sps + sth * cph * cps) + vy * (-sph * cps + sps * sth * cph) + vz * cph * cth;
}

/* f_func (E2E), R:57-152; term order follows the lambdified expression.
 * s[16], u[4], d[6] = (M_ext_x, M_ext_y, M_ext_z, F_ext_x, F_ext_y, F_ext_z) -> ds[16] */
void qro_f_e2e(const float* s, const float* u, const float* d, float* ds) {
    const float vx = s[3], vy = s[4], vz = s[5];
    const float p = s[9], q = s[10], r = s[11];
    const float w1 = s[12], w2 = s[13], w3 = s[14], w4 = s[15];
    const float sph = sinf(s[6]), cph = cosf(s[6]);
    const float sth = sinf(s[7]), cth = cosf(s[7]);
    const float sps = sinf(s[8]), cps = cosf(s[8]);
    const float tth = tanf(s[7]);

    /* rotation matrix columns as they appear in the expression (R:97-100) */
    const float r01 = sph * sth * cps - sps * cph, r11 = sph * sps * sth + cph * cps;
    const float r02 = sph * sps + sth * cph * cps, r12 = -sph * cps + sps * sth * cph;

    const float S = 4000.0f * w1 + 4000.0f * w2 + 4000.0f * w3 + 4000.0f * w4 + 28000.0f; /* R:106-109 */
    const float W1 = 4000.0f * w1 + 7000.0f, W2 = 4000.0f * w2 + 7000.0f;
    const float W3 = 4000.0f * w3 + 7000.0f, W4 = 4000.0f * w4 + 7000.0f;
    const float W1s = W1 * W1, W2s = W2 * W2, W3s = W3 * W3, W4s = W4 * W4;

    const float kx = 1.07933887e-5f, ky = 9.65250793e-6f, kz = 2.7862899e-5f;
    const float kw = 4.36301076e-8f, kh = 0.0625501332f;

    /* drag forces, R:124-126 (each k multiplies the velocity component first) */
    const float Fx = d[3] + (-kx * vx * cps * cth - kx * vy * sps * cth + kx * vz * sth) * S;
    const float Fy = d[4] + (-ky * vx * r01 - ky * vy * r11 - ky * vz * sph * cth) * S;
    const float vbx = vx * cps * cth + vy * sps * cth - vz * sth;
    const float vby = vx * r01 + vy * r11 + vz * sph * cth;
    const float T = d[5] - kw * W1s - kw * W2s - kw * W3s - kw * W4s -
                    (kz * vx * r02 + kz * vy * r12 + kz * vz * cph * cth) * S - kh * (vby * vby) - kh * (vbx * vbx);

    ds[0] = vx;
    ds[1] = vy;
    ds[2] = vz;
    ds[3] = Fx * cps * cth + Fy * r01 + r02 * T;                 /* R:138 */
    ds[4] = Fx * sps * cth + Fy * r11 + r12 * T;
    ds[5] = -Fx * sth + Fy * sph * cth + T * cph * cth + 9.81f;
    ds[6] = p + q * sph * tth + r * cph * tth;                   /* R:140-142 */
    ds[7] = q * cph - r * sph;
    ds[8] = q * sph / cth + r * cph / cth;
    /* R:144-146 with constants pre-folded by sympy */
    ds[9] = 1103.7527593819f * d[0] - 0.896247240618101f * q * r - 8.79803364238411f * vx * r01 -
            8.79803364238411f * vy * r11 - 8.79803364238411f * vz * sph * cth + 1.55842505518764e-6f * W1s -
            1.55842505518764e-6f * W2s - 1.55842505518764e-6f * W3s + 1.55842505518764e-6f * W4s;
    ds[10] = 805.152979066023f * d[1] + 0.924315619967794f * p * r + 10.4077084541063f * vx * cps * cth +
             10.4077084541063f * vy * sps * cth - 10.4077084541063f * vz * sth + 9.79081191626409e-7f * W1s +
             9.79081191626409e-7f * W2s - 9.79081191626409e-7f * W3s - 9.79081191626409e-7f * W4s;
    ds[11] = 486.854917234664f * d[2] - 0.163583252190847f * p * q - 0.395780237098345f * r -
             13.3373373580007f * u[0] + 13.3373373580007f * u[1] - 13.3373373580007f * u[2] +
             13.3373373580007f * u[3] + 8.33177659850698f * w1 - 8.33177659850698f * w2 +
             8.33177659850698f * w3 - 8.33177659850698f * w4;
    ds[12] = 16.6666666666667f * u[0] - 16.6666666666667f * w1;  /* R:112-115 */
    ds[13] = 16.6666666666667f * u[1] - 16.6666666666667f * w2;
    ds[14] = 16.6666666666667f * u[2] - 16.6666666666667f * w3;
    ds[15] = 16.6666666666667f * u[3] - 16.6666666666667f * w4;
}

/* f_func (INDI), I:43-110: s[13] = (..., p,q,r, T_norm), u = (p_cmd,q_cmd,r_cmd,T_cmd) */
void qro_f_indi(const float* s, const float* u, float* ds) {
    const float vx = s[3], vy = s[4], vz = s[5];
    const float p = s[9], q = s[10], r = s[11], Tn = s[12];
    const float sph = sinf(s[6]), cph = cosf(s[6]);
    const float sth = sinf(s[7]), cth = cosf(s[7]);
    const float sps = sinf(s[8]), cps = cosf(s[8]);
    const float tth = tanf(s[7]);
    const float r01 = sph * sth * cps - sps * cph, r11 = sph * sps * sth + cph * cps;
    const float r02 = sph * sps + sth * cph * cps, r12 = -sph * cps + sps * sth * cph;
    const float kx = 0.33915248f, ky = 0.4314916f;
    /* Dx = -kx*vbx, Dy = -ky*vby expanded term-wise (I:73-74,86-87); -T = -8*Tn - 8 (I:94) */
    const float Dx = -kx * vx * cps * cth - kx * vy * sps * cth + kx * vz * sth;
    const float Dy = -ky * vx * r01 - ky * vy * r11 - ky * vz * sph * cth;
    const float mT = -8.0f * Tn - 8.0f;
    ds[0] = vx;
    ds[1] = vy;
    ds[2] = vz;
    ds[3] = Dx * cps * cth + Dy * r01 + mT * r02;
    ds[4] = Dx * sps * cth + Dy * r11 + mT * r12;
    ds[5] = -Dx * sth + Dy * sph * cth + mT * cph * cth + 9.81f;
    ds[6] = p + q * sph * tth + r * cph * tth;
    ds[7] = q * cph - r * sph;
    ds[8] = q * sph / cth + r * cph / cth;
    ds[9] = -33.3333333333333f * p + 100.0f * u[0];              /* I:101-104 */
    ds[10] = -33.3333333333333f * q + 100.0f * u[1];
    ds[11] = -33.3333333333333f * r + 66.6666666666667f * u[2];
    ds[12] = 33.3333333333333f * u[3] - 33.3333333333333f * Tn;
}

/* ------------------------------------------------------------------------------------------------
 * Residual thrust / moment MLPs (R:227-262; layout of c_code/nn_thrust.c:5-79, nn_moment.c:5-81)
 * blob: thrust W1[32][7] b1[32] W2[1][32] b2[1] | moment W1[32][10] b1[32] W2[3][32] b2[3]
 * ---------------------------------------------------------------------------------------------- */
static void mlp_forward(const float* W1, const float* b1, const float* W2, const float* b2, const float* x,
                        int n_in, int n_out, float* y) {
    float h[32];
    for (int j = 0; j < 32; ++j) {
        float acc = 0.0f;
        for (int i = 0; i < n_in; ++i) acc += W1[j * n_in + i] * x[i];
        acc += b1[j];
        h[j] = acc < 0.0f ? 0.0f : acc; /* torch.nn.ReLU (R:244-245) propagates NaN: relu(NaN) = NaN -- pinned by fixture F11 */
    }
    for (int o = 0; o < n_out; ++o) {
        float acc = 0.0f;
        for (int j = 0; j < 32; ++j) acc += W2[o * 32 + j] * h[j];
        y[o] = acc + b2[o];
    }
}

/* thrust_moment_model_world_states, R:254-262 (one row) */
void qro_residual(const float* blob, const float* s, float* thrust, float* moment) {
    float x[10], vb[3];
    qro_body_velocity(s, vb);
    x[0] = s[12]; x[1] = s[13]; x[2] = s[14]; x[3] = s[15];
    x[4] = vb[0]; x[5] = vb[1]; x[6] = vb[2];
    x[7] = s[9]; x[8] = s[10]; x[9] = s[11];
    const float* tW1 = blob;            /* 224 */
    const float* tb1 = tW1 + 224;       /* 32 */
    const float* tW2 = tb1 + 32;        /* 32 */
    const float* tb2 = tW2 + 32;        /* 1 */
    const float* mW1 = tb2 + 1;         /* 320 */
    const float* mb1 = mW1 + 320;       /* 32 */
    const float* mW2 = mb1 + 32;        /* 96 */
    const float* mb2 = mW2 + 96;        /* 3 */
    mlp_forward(tW1, tb1, tW2, tb2, x, 7, 1, thrust);
    mlp_forward(mW1, mb1, mW2, mb2, x, 10, 3, moment);
}

/* ------------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11) -- this build's reset RNG specification
 * ---------------------------------------------------------------------------------------------- */
void qro_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; } /* [0,1), 24 bits */
static float uni(float lo, float hi, float u) { return lo + (hi - lo) * u; }

/* draws[k], k = 0..23: uniform [0,1) floats of (seed, global env id, episode) */
static void reset_draws(const qro_env* e, int i, uint32_t episode, float* draws, int nblocks) {
    const uint64_t gid = e->env_id_base + (uint64_t)i;
    const uint32_t key[2] = {(uint32_t)e->seed, (uint32_t)(e->seed >> 32)};
    for (int b = 0; b < nblocks; ++b) {
        const uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), episode, (uint32_t)b};
        uint32_t o[4];
        qro_philox4x32_10(ctr, key, o);
        for (int k = 0; k < 4; ++k) draws[4 * b + k] = u01(o[k]);
    }
}

/* reset_ for one env (distributions of R:455-489, I:270-296) */
static void reset_one(qro_env* e, int i) {
    float u[24];
    const int S = e->state_len;
    float* w = e->world + (size_t)i * S;
    const uint32_t ep = e->episode[i];
    e->episode[i] = ep + 1u;
    reset_draws(e, i, ep, u, e->variant == QRO_E2E ? 6 : 4);
    const float pi9 = 0.3490658503988659f, pi = 3.141592653589793f;
    w[0] = uni(-0.5f, 0.5f, u[0]) + e->start_pos[0];
    w[1] = uni(-0.5f, 0.5f, u[1]) + e->start_pos[1];
    w[2] = uni(-0.5f, 0.5f, u[2]) + e->start_pos[2];
    w[3] = uni(-0.5f, 0.5f, u[3]);
    w[4] = uni(-0.5f, 0.5f, u[4]);
    w[5] = uni(-0.5f, 0.5f, u[5]);
    w[6] = uni(-pi9, pi9, u[6]);
    w[7] = uni(-pi9, pi9, u[7]);
    w[8] = uni(-pi, pi, u[8]);
    w[9] = uni(-0.1f, 0.1f, u[9]);
    w[10] = uni(-0.1f, 0.1f, u[10]);
    w[11] = uni(-0.1f, 0.1f, u[11]);
    if (e->variant == QRO_E2E) {
        for (int k = 0; k < 4; ++k) w[12 + k] = uni(-1.0f, 1.0f, u[12 + k]);
        float* d = e->dist + (size_t)i * 6;
        for (int k = 0; k < 6; ++k) d[k] = e->dist_scale * uni(e->dist_ranges[k][0], e->dist_ranges[k][1], u[16 + k]);
    } else {
        w[12] = uni(-0.1f, 0.1f, u[12]);
    }
    e->steps[i] = 0;
    e->target[i] = 0;
}

/* ------------------------------------------------------------------------------------------------
 * Gate-frame observation, update_states_gate: R:365-450, I:218-265
 * ---------------------------------------------------------------------------------------------- */
static float np_mod(float a, float b) { /* NumPy float remainder (sign of divisor) */
    float m = fmodf(a, b);
    if (m != 0.0f) {
        if ((b < 0.0f) != (m < 0.0f)) m += b;
    } else {
        m = copysignf(0.0f, b);
    }
    return m;
}

static void observe_one(const qro_env* e, int i, float* o) {
    const int S = e->state_len;
    const float* w = e->world + (size_t)i * S;
    const int G = e->num_gates;
    const int g = e->target[i] % G;
    const float c = cosf(e->gate_yaw[g]), s = sinf(e->gate_yaw[g]);
    const float dx = w[0] - e->gate_pos[g][0], dy = w[1] - e->gate_pos[g][1];
    const float twopi = 6.283185307179586f, pi = 3.141592653589793f;
    o[0] = dx * c + dy * s;                                   /* R:380-382 */
    o[1] = dx * -s + dy * c;
    o[2] = w[2] - e->gate_pos[g][2];
    o[3] = w[3] * c + w[4] * s;                               /* R:386-389 */
    o[4] = w[3] * -s + w[4] * c;
    o[5] = w[5];
    o[6] = w[6];
    o[7] = w[7];
    float yaw = w[8] - e->gate_yaw[g];                        /* R:392-397 */
    yaw = np_mod(yaw, twopi);
    if (yaw > pi) yaw -= twopi;
    if (yaw < -pi) yaw += twopi;
    o[8] = yaw;
    for (int k = 9; k < S; ++k) o[k] = w[k];                  /* rates + rpms (or T_norm) */
    for (int a = 0; a < e->gates_ahead; ++a) {                /* R:406-412 */
        const int idx = (e->target[i] + a + 1) % G;
        o[S + 4 * a + 0] = e->gate_pos_rel[idx][0];
        o[S + 4 * a + 1] = e->gate_pos_rel[idx][1];
        o[S + 4 * a + 2] = e->gate_pos_rel[idx][2];
        o[S + 4 * a + 3] = e->gate_yaw_rel[idx];
    }
    if (e->variant == QRO_E2E) {                              /* R:414-448 */
        const float* d = e->dist + (size_t)i * 6;
        static const int col[4] = {0, 1, 2, 5};
        for (int k = 0; k < 4; ++k) {
            float lo = e->dist_ranges[col[k]][0], hi = e->dist_ranges[col[k]][1];
            if (lo == hi) { lo -= 1.0f; hi += 1.0f; }
            o[S + 4 * e->gates_ahead + k] = 2.0f * (d[col[k]] - lo) / (hi - lo) - 1.0f;
        }
    }
}

void qro_observe(qro_env* e, float* obs_out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(e->threads) if (e->threads > 1)
#endif
    for (int i = 0; i < e->n; ++i) observe_one(e, i, e->obs + (size_t)i * e->obs_len);
    if (obs_out) memcpy(obs_out, e->obs, sizeof(float) * (size_t)e->n * e->obs_len);
}

/* ------------------------------------------------------------------------------------------------
 * Environment object
 * ---------------------------------------------------------------------------------------------- */
qro_env* qro_create(int variant, int n, int gates_ahead, int pause_if_collision, uint64_t env_id_base) {
    qro_env* e = (qro_env*)calloc(1, sizeof(qro_env));
    e->variant = variant;
    e->n = n;
    e->gates_ahead = gates_ahead;
    e->pause_if_collision = pause_if_collision;
    e->state_len = variant == QRO_E2E ? 16 : 13;
    e->obs_len = e->state_len + 4 * gates_ahead + (variant == QRO_E2E ? 4 : 0); /* R:330, I:185 */
    e->max_steps = 1200;                                                        /* R:345 */
    e->dt = 0.01f;                                                              /* R:346 */
    e->dist_scale = 1.0f;                                                       /* R:358 */
    e->env_id_base = env_id_base;
    e->threads = 1;
    e->world = (float*)calloc((size_t)n * e->state_len, sizeof(float));
    e->dist = (float*)calloc((size_t)n * 6, sizeof(float));
    e->obs = (float*)calloc((size_t)n * e->obs_len, sizeof(float));
    e->target = (int32_t*)calloc(n, sizeof(int32_t));
    e->steps = (int32_t*)calloc(n, sizeof(int32_t));
    e->episode = (uint32_t*)calloc(n, sizeof(uint32_t));
    return e;
}

void qro_destroy(qro_env* e) {
    if (!e) return;
    free(e->world); free(e->dist); free(e->obs); free(e->target); free(e->steps); free(e->episode);
    free(e);
}

/* Track precompute, R:298-319: gate i expressed in the frame of gate i-1 (looped track) */
void qro_set_track(qro_env* e, const float* gate_pos, const float* gate_yaw, int G, const float* start_pos) {
    e->num_gates = G;
    for (int i = 0; i < G; ++i) {
        for (int k = 0; k < 3; ++k) e->gate_pos[i][k] = gate_pos[3 * i + k];
        e->gate_yaw[i] = gate_yaw[i];
    }
    for (int k = 0; k < 3; ++k) e->start_pos[k] = start_pos[k];
    for (int i = 0; i < G; ++i) {
        const int j = (i + G - 1) % G;
        const float dx = e->gate_pos[i][0] - e->gate_pos[j][0];
        const float dy = e->gate_pos[i][1] - e->gate_pos[j][1];
        const float c = cosf(e->gate_yaw[j]), s = sinf(e->gate_yaw[j]);
        e->gate_pos_rel[i][0] = c * dx + s * dy;
        e->gate_pos_rel[i][1] = -s * dx + c * dy;
        e->gate_pos_rel[i][2] = e->gate_pos[i][2] - e->gate_pos[j][2];
        e->gate_yaw_rel[i] = e->gate_yaw[i] - e->gate_yaw[j];
    }
}

void qro_set_residual(qro_env* e, const float* blob) {
    e->has_residual = blob != NULL;
    if (blob) memcpy(e->mlp, blob, sizeof(e->mlp));
}
void qro_set_disturbance(qro_env* e, const float* ranges, float scale) {
    memcpy(e->dist_ranges, ranges, sizeof(e->dist_ranges));
    e->dist_scale = scale;
}
void qro_set_limits(qro_env* e, int max_steps, float dt) { e->max_steps = max_steps; e->dt = dt; }
void qro_set_pause(qro_env* e, int pause) { e->pause = pause; }
void qro_set_terminal_obs(qro_env* e, float* buf) { e->term_obs = buf; } /* twin of qr_set_terminal_obs (qr_step rows) */
int qro_set_threads(qro_env* e, int threads) {
#ifdef _OPENMP
    e->threads = threads < 1 ? 1 : threads;
#else
    (void)threads;
    e->threads = 1;
#endif
    return e->threads;
}
void qro_seed(qro_env* e, uint64_t seed) {
    e->seed = seed;
    memset(e->episode, 0, sizeof(uint32_t) * (size_t)e->n);
}

float* qro_world(qro_env* e) { return e->world; }
float* qro_dist(qro_env* e) { return e->dist; }
float* qro_obs(qro_env* e) { return e->obs; }
int32_t* qro_target(qro_env* e) { return e->target; }
int32_t* qro_steps(qro_env* e) { return e->steps; }
uint32_t* qro_episode(qro_env* e) { return e->episode; }
int qro_state_len(const qro_env* e) { return e->state_len; }
int qro_obs_len(const qro_env* e) { return e->obs_len; }
void qro_get_track_tables(const qro_env* e, float* pos_rel, float* yaw_rel) {
    for (int i = 0; i < e->num_gates; ++i) {
        for (int k = 0; k < 3; ++k) pos_rel[3 * i + k] = e->gate_pos_rel[i][k];
        yaw_rel[i] = e->gate_yaw_rel[i];
    }
}

/* reset_(mask) / reset(): R:452-496 -- recomputes the observation of ALL envs (R:492) */
void qro_reset(qro_env* e, const uint8_t* mask, float* obs_out) {
    for (int i = 0; i < e->n; ++i)
        if (!mask || mask[i]) reset_one(e, i);
    qro_observe(e, obs_out);
}

/* step_async + step_wait: R:498-595, I:301-385.  actions [N][4]. Outputs may be NULL. */
void qro_step(qro_env* e, const float* actions, float* obs_out, float* rew_out, uint8_t* done_out,
              uint8_t* trunc_out) {
    const int S = e->state_len, G = e->num_gates;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(e->threads) if (e->threads > 1)
#endif
    for (int i = 0; i < e->n; ++i) {
        float* w = e->world + (size_t)i * S;
        const float* a = actions + (size_t)i * 4;
        float ds[16], nw[16];
        if (e->variant == QRO_E2E) {
            float d[6] = {0, 0, 0, 0, 0, 0};                       /* R:502-509 */
            if (e->has_residual) {
                float thrust, moment[3];
                qro_residual(e->mlp, w, &thrust, moment);
                d[0] = moment[0]; d[1] = moment[1]; d[2] = moment[2]; d[5] = thrust;
            }
            const float* dd = e->dist + (size_t)i * 6;
            for (int k = 0; k < 6; ++k) d[k] += dd[k];
            qro_f_e2e(w, a, d, ds);
        } else {
            qro_f_indi(w, a, ds);
        }
        for (int k = 0; k < S; ++k) nw[k] = w[k] + e->dt * ds[k];   /* R:512 */
        e->steps[i] += 1;                                          /* R:514 */

        const int g = e->target[i] % G;                            /* R:518-519 */
        const float* gp = e->gate_pos[g];
        const float gyaw = e->gate_yaw[g];
        const float ox = w[0] - gp[0], oy = w[1] - gp[1], oz = w[2] - gp[2];
        const float nx = nw[0] - gp[0], ny = nw[1] - gp[1], nz = nw[2] - gp[2];
        const float d2g_old = sqrtf(ox * ox + oy * oy + oz * oz);  /* R:522-525 */
        const float d2g_new = sqrtf(nx * nx + ny * ny + nz * nz);
        const float rat_penalty = 0.0f * sqrtf(nw[9] * nw[9] + nw[10] * nw[10] + nw[11] * nw[11]);
        float reward = d2g_old - d2g_new - rat_penalty;
        const float n0 = cosf(gyaw), n1 = sinf(gyaw);              /* R:528-534 */
        const float proj_old = ox * n0 + oy * n1;
        const float proj_new = nx * n0 + ny * n1;
        const int crossed = (proj_old < 0.0f) && (proj_new > 0.0f);
        const int inside = fabsf(nx) < 0.5f && fabsf(ny) < 0.5f && fabsf(nz) < 0.5f;
        const int outside = fabsf(nx) > 0.5f || fabsf(ny) > 0.5f || fabsf(nz) > 0.5f;
        const int gate_passed = crossed && inside;
        const int gate_collision = crossed && outside;
        if (gate_passed) reward = 10.0f - 10.0f * d2g_new;         /* R:537 */
        if (gate_collision) reward = -10.0f;                       /* R:540 */
        const int ground = nw[2] > 0.0f;                           /* R:543-544 */
        if (ground) reward = -10.0f;
        const int oob = fabsf(nw[0]) > 10.0f || fabsf(nw[1]) > 10.0f || fabsf(nw[9]) > 1000.0f ||
                        fabsf(nw[10]) > 1000.0f || fabsf(nw[11]) > 1000.0f; /* R:549-550 */
        if (oob) reward = -10.0f;
        const int max_steps_reached = e->steps[i] >= e->max_steps; /* R:553 */
        if (gate_passed) e->target[i] = (e->target[i] + 1) % G;    /* R:556-557 */
        int done = max_steps_reached || ground || gate_collision || oob; /* R:566 */

        if (e->pause) {                                            /* R:570-572 */
            done = 0;
        } else if (e->pause_if_collision) {                        /* R:573-578 */
            if (!done) memcpy(w, nw, sizeof(float) * S);
        } else {                                                   /* R:581-585 */
            memcpy(w, nw, sizeof(float) * S);
            if (done && e->term_obs) observe_one(e, i, e->term_obs + (size_t)i * e->obs_len); /* true terminal observation */
            if (done) reset_one(e, i);
        }
        if (rew_out) rew_out[i] = reward;
        if (done_out) done_out[i] = (uint8_t)done;
        if (trunc_out) trunc_out[i] = (uint8_t)max_steps_reached;
    }
    if (!e->pause) qro_observe(e, NULL); /* update_states() runs in both non-pause branches */
    if (obs_out) memcpy(obs_out, e->obs, sizeof(float) * (size_t)e->n * e->obs_len);
}
