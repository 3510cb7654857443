/*
 * quadrace.h -- C ABI of libquadrace.so: the MI355X-native (gfx950) vectorised quadrotor race
 * environment that replaces the NumPy hot path of tudelft/optimal_quad_control_RL.
 *
 * What each entry point replaces (R: = "3D quad race.ipynb", I: = "3D quad race INDI inner loop.ipynb",
 * raw .ipynb line numbers as in SURVEY.md section 0):
 *
 *   qr_create / qr_set_track      Quadcopter3DGates.__init__            R:288-360   I:143-214
 *   qr_set_residual               torch.load(NNDroneModel .pt)          R:227-245   (c_code/nn_thrust.c, nn_moment.c layout)
 *   qr_set_disturbance            env.disturbance_ranges / _scale       R:355-358, R:772-781
 *   qr_set_limits                 env.max_steps / env.dt                R:345-346, I:648
 *   qr_set_pause                  env.pause                             R:360, R:570-572
 *   qr_set_pause_if_collision     env.pause_if_collision                R:293, R:573-578
 *   qr_seed                       VecEnv.seed (no-op upstream)          R:600-601
 *   qr_reset                      reset() / reset_(dones)               R:452-496   I:267-299
 *   qr_step                       step_async() + step_wait()            R:498-595   I:301-385
 *   qr_observe                    update_states_gate()                  R:365-450   I:218-265
 *   qr_get_state / qr_set_state   direct attribute access to world_states, disturbances, target_gates,
 *                                 step_counts                           R:803, R:4499-4500
 *   qr_get_track_tables           gate_pos_rel / gate_yaw_rel           R:307-319   (pinned by c_code/nn_controller.c:40-60)
 *
 * Conventions (precedent: the reference's own ctypes use, R:4395-4417 -- POINTER(c_float) arguments,
 * caller-allocated outputs):
 *   - every function returns 0 on success, <0 on error (see QR_E_*); qr_last_error() gives text
 *     (thread-local). No exception crosses this boundary.
 *   - all buffer arguments named *_dev are DEVICE pointers (HBM of the GPU the env was created on) owned
 *     by the caller (e.g. torch tensor .data_ptr()); the library never frees or retains them beyond the
 *     kernels it enqueues in that call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream). Calls enqueue work and return
 *     without synchronising; results are ordered on that stream.
 *   - a handle is bound to one GPU and is not thread-safe.  Every entry point that touches the device makes the
 *     handle's GPU the current HIP device first (and leaves it current), so one process may drive handles on
 *     several GPUs; `stream` must belong to that GPU.
 *   - there is NO CPU fallback: qr_create fails with QR_E_NO_DEVICE when no gfx950 device is visible.
 *
 * Layouts at the boundary are the reference's row-major arrays: actions [N][4], obs [N][obs_len],
 * rewards [N] f32, dones [N] u8, world [N][S] (S = 16 E2E / 13 INDI), disturbances [N][6].
 * Internally the state lives in HBM as planar float4 structure-of-arrays.
 */
#ifndef QUADRACE_H
#define QUADRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libquadrace.so is built with -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

#define QR_ABI_VERSION 3   /* additive since 3 (no signature changed): qr_rollout_kernel_name (round 4), qr_set_rollout_form (round 5) */

enum {
    QR_OK = 0,
    QR_E_INVALID = -1,   /* bad argument */
    QR_E_NO_DEVICE = -2, /* no HIP device / not gfx950 */
    QR_E_HIP = -3,       /* a HIP runtime call failed */
    QR_E_STATE = -4      /* call not valid in the current state (e.g. step before set_track) */
};

enum { QR_VARIANT_E2E = 0, QR_VARIANT_INDI = 1 };

#define QR_MAX_GATES 32
#define QR_MAX_GATES_AHEAD 4
#define QR_RESIDUAL_FLOATS 740 /* 289 thrust + 451 moment */

typedef struct qr_env qr_env;

typedef struct qr_config {
    int32_t variant;            /* QR_VARIANT_E2E | QR_VARIANT_INDI */
    int32_t num_envs;           /* N >= 1 (envs simulated by this handle / this GPU) */
    int32_t gates_ahead;        /* 0..QR_MAX_GATES_AHEAD (R:292) */
    int32_t device;             /* HIP device ordinal */
    int32_t pause_if_collision; /* R:293, R:573-578 */
    int32_t reserved0;
    uint64_t env_id_base;       /* global index of local env 0 (multi-GPU sharding: RNG stream id) */
} qr_config;

int qr_abi_version(void);
const char* qr_last_error(void);

int qr_create(const qr_config* cfg, qr_env** out);
int qr_destroy(qr_env* env);

/* sizes: S (world-state length) and obs_len = S + 4*gates_ahead (+4 disturbance terms for E2E) */
int qr_state_len(const qr_env* env);
int qr_obs_len(const qr_env* env);
int qr_num_envs(const qr_env* env);

/* Host arrays: gate_pos [G][3], gate_yaw [G], start_pos [3]. Computes the relative-gate tables. */
int qr_set_track(qr_env* env, const float* gate_pos, const float* gate_yaw, int32_t num_gates,
                 const float* start_pos);
/* Host outputs (any may be NULL): gate_pos_rel [G][3], gate_yaw_rel [G]. */
int qr_get_track_tables(const qr_env* env, float* gate_pos_rel, float* gate_yaw_rel);

/* Host array of QR_RESIDUAL_FLOATS f32 (thrust W1[32][7] b1[32] W2[1][32] b2[1]; moment W1[32][10]
 * b1[32] W2[3][32] b2[3]) or NULL to disable the residual model (BASELINE config 1). E2E only. */
int qr_set_residual(qr_env* env, const float* blob, size_t n_floats);

/* Host array ranges[6][2] = (min,max) for (Mx,My,Mz,Fx,Fy,Fz); scale = disturbance_scale. E2E only. */
int qr_set_disturbance(qr_env* env, const float* ranges, float scale);

int qr_set_limits(qr_env* env, int32_t max_steps, float dt);
int qr_set_pause(qr_env* env, int32_t pause);
/* env.pause_if_collision after construction (qr_config.pause_if_collision sets the initial value) */
int qr_set_pause_if_collision(qr_env* env, int32_t on);

/* Terminal observations (what SB3 bootstraps time-limit truncations from, `infos[i]["terminal_observation"]`, R:589-594).
 * With a device buffer registered, every env that finishes an episode at a step writes the gate-frame observation of its
 * FINAL state -- taken before the auto-reset -- to row [env] (qr_step: buffer [N][obs_len]) or row [k][env] (qr_step_many,
 * qr_step_launches, qr_rollout_policy: buffer [K][N][obs_len], k = step within the call).  Rows of envs that did not
 * finish are not touched.  NULL (default) switches it off.  `rows` is the leading dimension of the buffer ([rows][N][obs_len];
 * 1 for a plain [N][obs_len] buffer): a K-step call with K > rows fails with QR_E_INVALID instead of writing past the end.
 * (The reference itself hands SB3 the post-reset row, a consequence of filling `infos` after reset_(): see DESIGN.md.) */
int qr_set_terminal_obs(qr_env* env, float* term_obs_dev, int32_t rows);

/* Philox4x32-10 key for the in-kernel reset RNG; also zeroes the per-env episode counters. */
int qr_seed(qr_env* env, uint64_t seed);

/* reset_(mask): mask_dev = NULL resets every env (reset()). Writes the gate-frame observation of ALL
 * envs to obs_out_dev [N][obs_len] (may be NULL). */
int qr_reset(qr_env* env, const uint8_t* mask_dev, float* obs_out_dev, void* stream);

/* One env step for all N envs (one fused kernel): residual MLP -> Euler integration -> reward ->
 * gate pass / collision / ground / bounds / max-steps -> (auto-reset) -> gate-frame observation.
 * trunc_out_dev (may be NULL) receives max_steps_reached per env ("TimeLimit.truncated").
 * With pause set, obs_out_dev is left untouched (the reference does not refresh it, R:570-572). */
int qr_step(qr_env* env, const float* actions_dev, float* obs_out_dev, float* rew_out_dev,
            uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream);

/* K consecutive steps with pre-recorded actions [K][N][4]; outputs are [K][N][...] (trunc may be NULL).
 * Bit-identical to K calls of qr_step, but executed as ONE fused rollout kernel that keeps the env state in
 * registers between steps (no per-step launch, state traffic or end-of-kernel write-back). */
int qr_step_many(qr_env* env, int32_t num_steps, const float* actions_dev, float* obs_out_dev,
                 float* rew_out_dev, uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream);

/* The same K steps as K separate step kernels (the kernels K calls of qr_step enqueue, without the per-call FFI cost):
 * the calling pattern of a closed loop whose policy runs between steps.  The K launches are captured once into a
 * hipGraph (K kernel nodes in a chain) and replayed while (K, buffers, env configuration) stay the same: dependent
 * kernel nodes of a graph start ~1 us sooner after each other than dependent launches on a stream. */
int qr_step_launches(qr_env* env, int32_t num_steps, const float* actions_dev, float* obs_out_dev,
                     float* rew_out_dev, uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream);

/* update_states(): recompute the observation from the current state. */
int qr_observe(qr_env* env, float* obs_out_dev, void* stream);

/* Diagnostic: out_dev [N][7] = (vbx, vby, vbz, thrust, Mx, My, Mz) -- body velocity (get_body_velocity, R:155) and the
 * residual thrust / moment MLP outputs (thrust_moment_model_world_states, R:254-262) for the CURRENT state of every env,
 * computed by the device functions the step kernels inline.  E2E with residual weights only. */
int qr_probe_residual(qr_env* env, float* out_dev, void* stream);

/* Row-major copies of the internal state (device pointers; any may be NULL).
 * dist is ignored for INDI. target/steps are int32. episode = per-env reset counter (RNG stream position). */
int qr_get_state(qr_env* env, float* world_dev, float* dist_dev, int32_t* target_dev, int32_t* steps_dev,
                 uint32_t* episode_dev, void* stream);
int qr_set_state(qr_env* env, const float* world_dev, const float* dist_dev, const int32_t* target_dev,
                 const int32_t* steps_dev, const uint32_t* episode_dev, void* stream);

/* Timing hooks for benchmarks (both block until the work is done).
 * qr_last_step_many_ms: hipEvent time (ms) from before the first to after the last launch of the most recent
 *   qr_step_many / qr_step_launches call, on its launch stream (back-to-back kernels; rocprofv3 shows no gaps).
 * qr_profile_steps: like qr_step_many, but brackets EVERY step kernel with its own hipEvent pair on the
 *   launch stream and returns the mean single-kernel duration (ms) and the whole-region time (ms). */
int qr_last_step_many_ms(qr_env* env, float* total_ms);
/* The hipEvent bracket behind qr_last_step_many_ms costs two marker packets per K-step call; on = 0 drops it (default: on).
 * It is also skipped, whatever the setting, while the caller is capturing `stream` into a hipGraph. */
int qr_set_timing(qr_env* env, int32_t on);
/* Name of the device kernel a qr_step_many call on this handle launches NOW (it depends on the env count, the variant, the mode
 * flags and whether a terminal-observation buffer is registered), e.g. "rollout_fast_mlp_kernel" -- the symbol rocprofv3 lists as
 * qr::<name><variant, gates_ahead>.  Benchmarks print it so that profiles and counter evidence can be matched to the run. */
const char* qr_rollout_kernel_name(const qr_env* env);
/* Which family of fused kernels qr_step_many may pick from (all produce bit-identical results; this is a testing / A-B hook, there is
 * no environment variable): AUTO = by env count and mode; flags MULTI_WAVE = the forms built for more than
 * one workgroup per CU, at any env count; GENERAL = the general kernels (every mode) for every launch (the two may be or-ed); ONE_WAVE = the forms built for one
 * workgroup per CU at any env count (slower there: profiles/r05_one_wave_ab.txt). */
enum { QR_ROLLOUT_AUTO = 0, QR_ROLLOUT_MULTI_WAVE = 1, QR_ROLLOUT_GENERAL = 2, QR_ROLLOUT_ONE_WAVE = 4 /* the one-wave-per-SIMD forms at any env count */ };
int qr_set_rollout_form(qr_env* env, int32_t form);
int qr_profile_steps(qr_env* env, int32_t num_steps, const float* actions_dev, float* obs_out_dev,
                     float* rew_out_dev, uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream,
                     float* mean_kernel_ms, float* region_ms);

/* ---------------------------------------------------------------------------------------------------------------
 * Policy network on the matrix cores: the MLP the reference trains and deploys,
 * obs[obs_len] -> 120 -> 120 -> 120 -> 4 with ReLU (SB3 MlpPolicy net_arch pi=[120,120,120], R:783; generated C twin
 * c_code/neural_network.c:397-430 nn_forward).  f16 operands, f32 accumulation.  Replaces `model.predict(env.states,
 * deterministic=True)` (R:801) between env steps without leaving the GPU.
 *   weights: host float32 arrays in torch.nn.Linear layout, w[out][in], b[out].
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct qr_policy qr_policy;
int qr_policy_create(int32_t obs_len, int32_t device, qr_policy** out);
int qr_policy_destroy(qr_policy* policy);
const char* qr_policy_last_error(void);
int qr_policy_set_weights(qr_policy* policy, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* w3, const float* b3, const float* w4, const float* b4);
/* obs_dev [n][obs_len] row-major -> mean_out_dev [n][4] (action means, i.e. the deterministic action before clipping) */
int qr_policy_forward(qr_policy* policy, int32_t n, const float* obs_dev, float* mean_out_dev, void* stream);
/* The same network at the reference's precision (round 6): both operands of every layer as two f16 pieces (w x ~ W0 X0 + W1 X0 + W0 X1,
 * f32 accumulation on the matrix core: three matrix instructions per K-step) -- float32-class results (max |d mean| vs the reference's
 * generated nn_forward, c_code/neural_network.c:397-430, at the 1e-6 level instead of 7e-4) for evaluation and for precision="f32"
 * collection (its update half is qr_ppo_grad_f32class below); qr_policy_forward stays the throughput path.  Same arguments. */
int qr_policy_forward_f32class(qr_policy* policy, int32_t n, const float* obs_dev, float* mean_out_dev, void* stream);

/* Closed-loop rollout: K steps of  obs -> policy -> a ~ N(mean, exp(log_std)^2) -> env.step(clip(a, -1, 1))  in ONE
 * kernel (PPO's collect phase, R:820 -> SB3 collect_rollouts, without leaving the chip).  Row t of the outputs is
 * (obs_t the action was computed from, unclipped action_t, log-prob_t of that action, reward_t, done_t[, trunc_t]);
 * last_obs_dev [N][obs_len] (may be NULL) receives the observation after the last step (value bootstrap).
 * log_std: host float[4].  Action noise is Philox4x32-10 + Box-Muller keyed by (noise_seed, global env id,
 * first_step + t): pass the number of steps already taken as first_step.  `deterministic` is a set of flags (0 / 1 as before):
 * QR_ROLLOUT_DETERMINISTIC: action = mean; QR_ROLLOUT_F32CLASS (round 6): the policy forward inside the kernel is the reference-precision
 * one of qr_policy_forward_f32class (slower: three matrix instructions per K-step, low-piece weights read from global memory).
 * qr_last_step_many_ms() reports this launch too. */
enum { QR_ROLLOUT_DETERMINISTIC = 1, QR_ROLLOUT_F32CLASS = 2 };
int qr_rollout_policy(qr_env* env, qr_policy* policy, int32_t num_steps, const float* log_std, uint64_t noise_seed,
                      uint64_t first_step, int32_t deterministic, float* obs_out_dev, float* act_out_dev,
                      float* logp_out_dev, float* rew_out_dev, uint8_t* done_out_dev, uint8_t* trunc_out_dev,
                      float* last_obs_dev, void* stream);

/* ---- PPO minibatch update on the matrix cores (replaces SB3's PPO.train inner loop, R:783-795 / R:820) -------------
 * Networks: policy obs -> 120 -> 120 -> 120 -> 4 and value obs -> 120 -> 120 -> 120 -> 1 (ReLU), log_std[4].
 * Parameters live in ONE flat float32 device vector owned by the caller (torch Linear layouts, w[out][in]):
 *   [ pi: w1 b1 w2 b2 w3 b3 w4 b4 | vf: w1 b1 w2 b2 w3 b3 w4 b4 | log_std[4] ]        (qr_ppo_num_params floats)
 * Loss (SB3): -mean(min(A r, A clip(r, 1-eps, 1+eps))) + vf_coef * mse(v, ret) - ent_coef * mean(entropy), with the
 * advantages of the minibatch normalised to zero mean / unit (unbiased) std; gradients are clipped to max_grad_norm
 * (global L2) and applied with Adam (torch.optim.Adam semantics).  Forward / backward GEMMs use f16 operands with f32
 * accumulation; parameters, gradient accumulation and Adam are f32.
 * Rollout rows (obs [rows][obs_len], act [rows][4], old_logp / adv / ret [rows]) are device arrays; idx_dev[B] selects
 * the rows of this minibatch (64 <= B <= max_minibatch; a last, partial group of 64 rows is masked inside the gradient kernel --
 * the reference's batch_size is 5000, R:792). */
typedef struct qr_ppo qr_ppo;
int qr_ppo_create(int32_t obs_len, int32_t device, int32_t max_minibatch, qr_ppo** out);
/* The same with explicit choices (verification / A-B; qr_ppo_create = flags 0 = the measured-fastest form; the library reads no
 * environment variable): PARTIAL_F32 keeps the per-workgroup gradient partials in f32 instead of bf16 (twice the bytes through the
 * fabric; the form the kernel is verified in to f32 summation noise); NO_EPOCH_GRAPH makes qr_ppo_epoch enqueue plain launches
 * instead of replaying a captured graph.  Any other bit is QR_E_INVALID (bits 2 and 4 selected the round 1-2 forms of the gradient
 * kernel, removed in round 6). */
enum { QR_PPO_PARTIAL_F32 = 1, QR_PPO_NO_EPOCH_GRAPH = 8 };
int qr_ppo_create_ex(int32_t obs_len, int32_t device, int32_t max_minibatch, int32_t flags, qr_ppo** out);
int qr_ppo_destroy(qr_ppo* ppo);
int qr_ppo_num_params(const qr_ppo* ppo);
/* builds the f16 operand images from the parameters: call once before the first qr_ppo_minibatch and after any
 * change of theta made outside this library */
int qr_ppo_pack(qr_ppo* ppo, const float* theta_dev, void* stream);
/* gradient only (no clipping, no optimiser step; the operand images are rebuilt first unless theta_dev is the vector they were
 * last built from / kept in step with): grad_out_dev [num_params + 4] -- the gradient followed by this
 * minibatch's statistics {sum of per-sample surrogate losses, sum of squared value errors, sum of approx-KL terms, number
 * of clipped samples}; stats_dev (may be NULL) float[4] is ACCUMULATED into with the same four sums */
int qr_ppo_grad(qr_ppo* ppo, const float* theta_dev, const float* obs_dev, const float* act_dev,
                const float* old_logp_dev, const float* adv_dev, const float* ret_dev, const int32_t* idx_dev, int32_t B,
                float clip, float vf_coef, float ent_coef, float* grad_out_dev, float* stats_dev, void* stream);
/* The same gradient at the reference's precision (round 6): every matrix product of the forward pass, the backward pass and the weight
 * gradients on the matrix core with BOTH operands as three bf16 pieces (x = X0 + X1 + X2 exactly; six matrix instructions per K-step) and f32
 * accumulation, fixed summation order (csrc/quadrace_ppo_f32.hip) -- float32-class like the reference's torch update (R:783-795; cosine against
 * float64 autograd 1 - 1e-13 instead of 0.9985), several times slower than qr_ppo_grad.  Same arguments, same grad_out layout; theta_dev is read
 * directly (no operand images); 2 <= B <= 2 097 120.  Follow with qr_ppo_apply for the step.  Its ~20 launches are replayed as one graph per distinct
 * argument set (up to 256 sets per handle are kept): pass the same buffers from call to call to hit it; QR_PPO_NO_EPOCH_GRAPH or a capturing
 * `stream` gives plain launches. */
int qr_ppo_grad_f32class(qr_ppo* ppo, const float* theta_dev, const float* obs_dev, const float* act_dev,
                         const float* old_logp_dev, const float* adv_dev, const float* ret_dev, const int32_t* idx_dev, int32_t B,
                         float clip, float vf_coef, float ent_coef, float* grad_out_dev, float* stats_dev, void* stream);
/* one complete minibatch update of theta (and the Adam moments).  adam_step = 1, 2, ...: the caller counts the updates; adam_step = 0:
 * the library's device-resident count of optimiser steps REALLY taken is used and advanced (launches turned into no-ops by the
 * early stop or a non-finite gradient norm do not count, like torch.optim.Adam under SB3) -- see qr_ppo_adam_step.  lr >= 0 (a
 * negative or NaN learning rate is QR_E_INVALID, here and in qr_ppo_apply / qr_ppo_epoch).
 * Two launches:
 * ONE gradient kernel (forward, loss, backward and the weight gradients of 128 samples per workgroup and pass; per-workgroup
 * partial sums, rounded to bf16 when they leave the workgroup) and ONE kernel that sums the partials in f32 in a fixed order, takes
 * the global norm across a grid-wide barrier, clips, applies Adam and re-packs the f16 operand images.  (Other forms of the kernels: qr_ppo_create_ex.)
 *  A non-finite gradient norm makes
 * the whole update a no-op (counted, see qr_ppo_status).  stats_dev (may be NULL) float[4] is accumulated into. */
int qr_ppo_minibatch(qr_ppo* ppo, float* theta_dev, float* adam_m_dev, float* adam_v_dev, const float* obs_dev,
                     const float* act_dev, const float* old_logp_dev, const float* adv_dev, const float* ret_dev,
                     const int32_t* idx_dev, int32_t B, float clip, float vf_coef, float ent_coef, float max_grad_norm,
                     float lr, float beta1, float beta2, float eps, int32_t adam_step, float* stats_dev, void* stream);
/* Announces an epoch: idx_dev [num_minibatches * B] is the permutation whose consecutive slices of B rows will be passed,
 * in order, to qr_ppo_minibatch / qr_ppo_grad.  One launch computes the advantage mean / std sums of every minibatch
 * (otherwise each minibatch call spends a launch on its own).  Optional. */
int qr_ppo_epoch_begin(qr_ppo* ppo, const float* adv_dev, const int32_t* idx_dev, int32_t B, int32_t num_minibatches,
                       void* stream);
/* num_epochs whole epochs: per epoch num_minibatches consecutive qr_ppo_minibatch updates on the rows perm_dev[k B .. (k + 1) B),
 * preceded by the qr_ppo_epoch_begin statistics launch -- enqueued as ONE replayed hipGraph (dependent nodes of a graph start
 * sooner after each other than dependent stream launches, and the host issues one launch instead of 2 num_minibatches + 1 per
 * epoch).  The graph is captured on first use and re-captured when an argument changes.  Uses the device-resident Adam step count
 * (adam_step = 0 semantics) and hands `lr` to the kernels through device memory, so a learning-rate schedule does not force a
 * re-capture.  perm_dev [num_minibatches * B] int32:
 *   device_shuffle == 0: the caller's permutation (num_epochs must be 1; rewrite its CONTENT in place between calls);
 *   device_shuffle != 0: the buffer is FILLED at the start of every epoch with a fresh pseudo-random permutation of
 *     [0, num_minibatches * B) -- a keyed 8-round Feistel bijection, cycle-walked, O(1) per element instead of the radix sort behind
 *     torch.randperm -- keyed by (seed, number of epochs shuffled so far), see qr_ppo_shuffle_state. */
int qr_ppo_epoch(qr_ppo* ppo, float* theta_dev, float* adam_m_dev, float* adam_v_dev, const float* obs_dev, const float* act_dev,
                 const float* old_logp_dev, const float* adv_dev, const float* ret_dev, int32_t* perm_dev, int32_t B,
                 int32_t num_minibatches, int32_t num_epochs, int32_t device_shuffle, float clip, float vf_coef, float ent_coef,
                 float max_grad_norm, float lr, float beta1, float beta2, float eps, float* stats_dev, void* stream);
/* state2[0] = seed of the on-device permutations, state2[1] = epochs shuffled so far; set == 0 reads, set != 0 writes.  Blocks. */
int qr_ppo_shuffle_state(qr_ppo* ppo, uint64_t* state2, int32_t set, void* stream);
/* device-resident optimiser step count: set == 0 reads it into *value (for a checkpoint), set != 0 writes *value (after loading
 * one).  Blocks. */
int qr_ppo_adam_step(qr_ppo* ppo, int32_t* value, int32_t set, void* stream);
/* SB3's `target_kl` early stop, decided on the device: before an optimiser step is taken the update kernel compares the
 * minibatch's mean approx-KL with 1.5 * target_kl; if it is larger the step is NOT taken and a sticky stop flag makes every
 * later qr_ppo_minibatch / qr_ppo_apply launch a no-op, until qr_ppo_control(..., clear != 0) (call it at the start of
 * each PPO.train()).  target_kl <= 0 disables the check.  No host synchronisation is involved. */
int qr_ppo_control(qr_ppo* ppo, float target_kl, int32_t clear, void* stream);
/* blocks; out4 = {stopped (0/1), optimiser steps taken since the last clear, updates skipped for a non-finite gradient
 * norm, grid-barrier timeouts (must be 0)} */
int qr_ppo_status(qr_ppo* ppo, int32_t* out4, void* stream);

/* forward pass of one network with the current operand images (net 0: action means; net 1: value in column 0): out_dev [n][4] */
int qr_ppo_forward(qr_ppo* ppo, int32_t net, int32_t n, const float* obs_dev, float* out_dev, void* stream);
/* GAE(lambda) over a rollout buffer [T][N] (SB3 RolloutBuffer.compute_returns_and_advantage; done = 0 / 1 floats).
 * term_val_dev (may be NULL) [T][N] = V(terminal observation) at the steps that ended by the time limit and 0 elsewhere:
 * gamma * term_val is added to those rewards, which is how SB3's collect_rollouts bootstraps truncated episodes (R:589-594
 * hands it `terminal_observation` / `TimeLimit.truncated`); without it truncations count as terminations.  When ep_*_dev are
 * given, the running episode return / length / gate count per env are updated (from the raw rewards) and the sums over finished
 * episodes ACCUMULATED into fin_dev[4] = {sum return, sum length, sum gates, episodes} (VecMonitor, R:769) */
int qr_ppo_gae(qr_ppo* ppo, int32_t T, int32_t N, const float* rew_dev, const float* done_dev, const float* val_dev,
               const float* last_val_dev, const float* term_val_dev, float gamma, float lam, float* adv_out_dev,
               float* ret_out_dev, float* ep_ret_dev, float* ep_len_dev, float* ep_gates_dev, float* fin_dev, void* stream);
/* data-parallel training (one process per GPU): each rank computes qr_ppo_grad on its own rows, the caller averages the
 * [num_params + 4] vector across ranks (a single all-reduce of ~250 KB over RCCL: gradient AND minibatch statistics), then
 * every rank applies the identical update: global-norm clip, Adam, operand re-pack -- and takes the identical target-KL
 * decision, because the KL sum travelled with the gradient.  B = rows per rank of this minibatch.  stats_dev as above.
 * adam_step as in qr_ppo_minibatch (0 = the device-resident count). */
int qr_ppo_apply(qr_ppo* ppo, float* theta_dev, float* adam_m_dev, float* adam_v_dev, float* grad_dev, int32_t B,
                 float max_grad_norm, float lr, float beta1, float beta2, float eps, int32_t adam_step, float* stats_dev,
                 void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* QUADRACE_H */
