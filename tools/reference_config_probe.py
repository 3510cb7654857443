#!/usr/bin/env python3
"""The reference's own training-cell settings on this build (R:765, R:783-795: 100 envs, n_steps = 1000, batch_size = 5000, n_epochs = 10,
gamma = 0.999, MlpPolicy 3 x 120 ReLU) through the SB3-shaped object: env-steps/s of model.learn(), and where the time goes.
Usage (GPU box): python tools/reference_config_probe.py [rollouts]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd import PPO, Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, square_track

rollouts = int(sys.argv[1]) if len(sys.argv) > 1 else 20
env = Quadcopter3DGates(100, *square_track(), gates_ahead=1, infos_mode="none", seed=0)
env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
model = PPO("MlpPolicy", env, policy_kwargs=dict(activation_fn=torch.nn.ReLU, net_arch=[dict(pi=[120, 120, 120], vf=[120, 120, 120])]),
            n_steps=1000, batch_size=5000, n_epochs=10, gamma=0.999, seed=0)
model.learn(total_timesteps=2 * 100000)          # warm-up (graph capture)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.learn(total_timesteps=rollouts * 100000, reset_num_timesteps=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tr = model._trainer
print(json.dumps({"what": "reference training-cell hyper-parameters (100 envs x 1000 steps, batch_size 5000, 10 epochs) via optimal_quad_control_rl_amd.PPO.learn",
                  "native_update": bool(tr.native_update), "fused_collect": bool(tr.fused_collect), "rollouts": rollouts,
                  "env_steps_per_s": rollouts * 100000 / dt, "ms_per_rollout": dt / rollouts * 1e3,
                  "updates_applied": tr.stats["updates"], "updates_skipped": tr.stats["skipped_nonfinite"]}))
