#!/usr/bin/env python3
"""BASELINE config 5: PPO on the MI355X race env (GPU box).

    python tools/train_ppo.py [--variant indi|e2e] [--envs 65536] [--steps 3e8] [--track square|zigzag]

Prints training progress and a final deterministic evaluation (gates per episode, seconds per gate/lap on the
4-gate square track; the reference's simulated lap times there are 2.5-3.2 s, FP:3474-3488)."""
import argparse, os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd import (Quadcopter3DGates, Quadcopter3DGatesINDI, TRAIN_DISTURBANCE_RANGES,
                                         square_track, zigzag_track)
from optimal_quad_control_rl_amd.ppo import PPO

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="indi")
ap.add_argument("--track", default="square")
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=float, default=3e8)
ap.add_argument("--n-steps", type=int, default=32)
ap.add_argument("--epochs", type=int, default=5)
ap.add_argument("--minibatches", type=int, default=4)
ap.add_argument("--lr", type=float, default=3e-4)
ap.add_argument("--target-kl", type=float, default=0.02)
ap.add_argument("--lr-final", type=float, default=0.1)
ap.add_argument("--fused", action="store_true", help="collect with the closed-loop rollout kernel (qr_rollout_policy)")
ap.add_argument("--native-update", action="store_true", help="minibatch updates in the matrix-core kernels (qr_ppo_minibatch)")
ap.add_argument("--ent-coef", type=float, default=0.0)
ap.add_argument("--gamma", type=float, default=0.999)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--out", default="")
a = ap.parse_args()

# one process per GPU under torchrun (data-parallel PPO: --envs is per GPU, env_id_base = rank * envs keys disjoint reset and
# action-noise streams; gradients are averaged per minibatch, see MfmaPpoUpdater.minibatch)
rank, world, local_rank = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
if "RANK" in os.environ:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert a.native_update, "data-parallel training uses the matrix-core update (--native-update)"
if rank != 0:
    sys.stdout = open(os.devnull, "w")

trk = square_track() if a.track == "square" else zigzag_track()
cls = Quadcopter3DGates if a.variant == "e2e" else Quadcopter3DGatesINDI
env = cls(a.envs, *trk, gates_ahead=1, infos_mode="none", seed=1 + a.seed, env_id_base=rank * a.envs)
if a.variant == "e2e":
    env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
model = PPO(env, seed=a.seed, ent_coef=a.ent_coef, gamma=a.gamma, n_steps=a.n_steps, n_epochs=a.epochs, batch_size=a.envs * a.n_steps // a.minibatches, learning_rate=a.lr,
            target_kl=a.target_kl, lr_final_frac=a.lr_final, total_timesteps_hint=int(a.steps) // world, fused_collect=a.fused,
            native_update=a.native_update)
if "RANK" in os.environ:
    model._updater.broadcast_parameters(0)   # identical start on every rank (the seed already makes it so; this guarantees it)
    model.noise_seed = a.seed                 # same key, different global env ids -> independent action noise per rank
best = {"gates": -1.0, "state": None}
def keep_best(m):
    g = m.stats.get("gates_per_episode", 0.0)
    if g > best["gates"] and m.stats.get("ep_len_mean", 0) > 600:
        best["gates"] = g
        best["state"] = {k: v.clone() for k, v in m.policy.state_dict().items()}
t0 = time.perf_counter()
model.learn(int(a.steps) // world, log_every=20, callback=keep_best)   # --steps counts env-steps of the whole job
if best["state"] is not None:
    model.policy.load_state_dict(best["state"])  # evaluate the best checkpoint (the reference saves one every 10 rollouts, R:823)
torch.cuda.synchronize()
train_s = time.perf_counter() - t0

# deterministic evaluation on a fresh env: 1200 steps = 12 s of flight, no auto-reset masking of crashes
n_eval = 4096
ev = cls(n_eval, *trk, gates_ahead=1, infos_mode="none", seed=99)
if a.variant == "e2e":
    ev.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
obs = ev.reset_device()
gates = torch.zeros(n_eval, device="cuda"); crashes = torch.zeros(n_eval, device="cuda")
for k in range(1200):
    obs, rew, done, trunc = ev.step_device(model.predict(obs).contiguous())
    gates += (rew > 5).float(); crashes += (done.float() - trunc.float()).clamp(min=0)
dt = 0.01
res = dict(world_size=world, fused_collect=a.fused, native_update=a.native_update, variant=a.variant, track=a.track, envs=a.envs, train_steps=model.num_timesteps * world, train_seconds=train_s,
           train_Msteps_per_s=model.num_timesteps * world / train_s / 1e6,
           eval_gates_per_12s=float(gates.mean()), eval_crashes_per_12s=float(crashes.mean()),
           eval_seconds_per_gate=float(1200 * dt / gates.mean().clamp(min=1e-9)),
           eval_seconds_per_lap_4gates=float(4 * 1200 * dt / gates.mean().clamp(min=1e-9)), **model.stats)
print(json.dumps(res))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
