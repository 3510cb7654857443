/*
 * quad3d.h -- C ABI (part of libquadrace.so) for the two PREDECESSOR environments of the reference's
 * "3D quad.ipynb" (SURVEY.md section 8(f) #4), on MI355X (gfx950).  Q3: = that notebook (cell numbers).
 *
 *   kind Q3_KIND_HOVER   class Quadcopter3DVec       Q3 cell 6   hover-at-origin task; float64 state/reward
 *                        (the reference allocates np.zeros((N,16)) = float64), float32 actions
 *   kind Q3_KIND_GATES   class Quadcopter3DVecGates  Q3 cell 14  fly the gate sequence once; float32 state,
 *                        the observation is the raw 16-state (no gate frame)
 *
 * What each entry point replaces:
 *   q3_create                  __init__                              Q3 cell 6 / cell 14
 *   q3_set_track               __init__(gates_pos, gate_yaw, start_pos)          cell 14
 *   q3_set_limits              env.max_steps / env.dt                            cell 6, 14
 *   q3_set_thresholds          env.pos_threshold ... rat_threshold               cell 6
 *   q3_seed                    seed() (a no-op upstream; resets draw from NumPy's global generator)
 *   q3_reset                   reset() / reset_(dones)
 *   q3_step                    step_async() + step_wait(): f_func (cell 2) forward-Euler step, reward,
 *                              termination, auto-reset; returns `self.states`
 *   q3_step_many               K x q3_step in one kernel with the state held in registers
 *   q3_get_state/q3_set_state  attribute access to env.states / target_gates / step_counts
 *
 * Conventions are those of quadrace.h: 0 on success, QR_E_* (<0) on error with text in qr_last_error();
 * *_dev arguments are DEVICE pointers owned by the caller; `stream` is a hipStream_t as void*; calls enqueue
 * and return; one handle per GPU, not thread-safe; NO CPU fallback.
 *
 * Element type T of states / rewards: double for Q3_KIND_HOVER, float for Q3_KIND_GATES (q3_elem_size()).
 * Layouts are the reference's row-major arrays: states [N][16] T, actions [N][4] float, rewards [N] T, dones [N] u8.
 */
#ifndef QUAD3D_H
#define QUAD3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libquadrace.so is built with -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

enum { Q3_KIND_HOVER = 0, Q3_KIND_GATES = 1 };

typedef struct q3_env q3_env;

/* env_id_base: global id of this handle's env 0 (keys the reset stream, so shards of one big env agree with it) */
int q3_create(int kind, int num_envs, int device, uint64_t env_id_base, q3_env** out);
int q3_destroy(q3_env* env);
int q3_num_envs(const q3_env* env);
int q3_elem_size(const q3_env* env); /* 8 (hover) or 4 (gates) */

/* host pointers; gate_pos [G][3], gate_yaw [G], G <= 32 (values are rounded to float32 like astype(np.float32)) */
int q3_set_track(q3_env* env, const float* gate_pos, const float* gate_yaw, int num_gates, const float start_pos[3]);
int q3_set_limits(q3_env* env, int max_steps, double dt);
int q3_set_thresholds(q3_env* env, double pos, double vel, double ang, double rat);
int q3_seed(q3_env* env, uint64_t seed);

/* reset_(mask): mask_dev = NULL resets every env (reset()); states_out_dev (may be NULL) receives env.states */
int q3_reset(q3_env* env, const uint8_t* mask_dev, void* states_out_dev, void* stream);

/* one step_wait(); any output pointer may be NULL.  trunc = the envs for which the reference sets
 * infos[i]["TimeLimit.truncated"] (hover: max_steps or out of bounds; gates: max_steps) */
int q3_step(q3_env* env, const float* actions_dev, void* states_out_dev, void* rew_out_dev, uint8_t* done_out_dev,
            uint8_t* trunc_out_dev, void* stream);

/* K steps in one launch: actions [K][N][4]; rew_out [K][N] T and done_out [K][N] (either may be NULL);
 * states_out (may be NULL) = env.states after the last step */
int q3_step_many(q3_env* env, const float* actions_dev, int num_steps, void* rew_out_dev, uint8_t* done_out_dev,
                 void* states_out_dev, void* stream);

/* K steps in one launch WITH what a trainer consumes (round 6): states_steps_out [K][N][16] T = env.states after every step (the
 * array step_wait() returns, Q3:399 / Q3:744: the first state of the next episode for an env that finished), rew_out [K][N] T,
 * done_out / trunc_out [K][N] (the last three may be NULL).  Same results as K x q3_step, bit for bit. */
int q3_rollout(q3_env* env, const float* actions_dev, int num_steps, void* states_steps_out_dev, void* rew_out_dev,
               uint8_t* done_out_dev, uint8_t* trunc_out_dev, void* stream);

/* states [N][16] T, target [N] i32 (gates only; ignored / zero for hover), steps [N] i32; any may be NULL */
int q3_get_state(q3_env* env, void* states_dev, int32_t* target_dev, int32_t* steps_dev, void* stream);
int q3_set_state(q3_env* env, const void* states_dev, const int32_t* target_dev, const int32_t* steps_dev, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
