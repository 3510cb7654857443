#!/usr/bin/env python3
"""BASELINE config 5 on the REFERENCE'S OWN RECIPE, to convergence: the training cell of '3D quad race.ipynb'
(R:765, R:783-795, R:816-831) verbatim on this build's SB3-shaped objects --

    env = VecMonitor(Quadcopter3DGates(num_envs=100, ..., gates_ahead=1)); env.venv.disturbance_ranges = ...
    model = PPO("MlpPolicy", env, policy_kwargs=dict(activation_fn=ReLU, net_arch=[dict(pi=[120]*3, vf=[120]*3)], log_std_init=0),
                n_steps=1000, batch_size=5000, n_epochs=10, gamma=0.999)            # constant lr 3e-4, SB3 defaults otherwise
    while steps < 1.03e9: model.learn(10 rollouts, reset_num_timesteps=False); model.save(...)

-- on the 4-gate square track the paper's simulated lap times are quoted on (FP:3474-3488: first lap 2.97 s, flying laps 2.51-2.59 s).
Every `--eval-every`-th checkpoint the CURRENT policy is evaluated with the training clock stopped (deterministic actions, 4 096 fresh
envs, 20 s of flight, lap times per lap like the reference's table), which gives the curve, the final policy, the best checkpoint (the
reference keeps them all and picks afterwards) and the wall-clock at which the reference's level (flying lap <= 2.6 s, <= 0.1 crashes
per 12 s) is first reached.

    python tools/reference_recipe_run.py --seed S [--steps 1.03e9] [--out profiles/r04_refrecipe_seedS.json]      (GPU box)
"""
import argparse, json, os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd import PPO, Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, VecMonitor, square_track

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--steps", type=float, default=1.03e9)
ap.add_argument("--eval-every", type=int, default=40, help="evaluate every this many checkpoints (one checkpoint = 10 rollouts = 1e6 env-steps)")
ap.add_argument("--lap-target", type=float, default=2.6)
ap.add_argument("--precision", default="f16-operands", help="f16-operands (hand-written matrix-core kernels) | f32 (f32-class collect kernel + f32-class gradient kernels: "
                "the reference's precision end to end) | f32-collect (f32-class collect kernel + the f16-operand update kernels)")
ap.add_argument("--stop-when-reached", action="store_true", help="end the run at the first evaluation that reaches the reference's level (lap-target, <= 0.1 crashes / 12 s): "
                "answers 'does this seed reach it' at a fraction of the 1.03e9 steps; the result says so (stopped_when_reached)")
ap.add_argument("--out", default="")
a = ap.parse_args()

trk = square_track()
env = Quadcopter3DGates(num_envs=100, gates_pos=trk[0], gate_yaw=trk[1], start_pos=trk[2], gates_ahead=1, seed=1 + a.seed)   # R:765
env = VecMonitor(env)                                                                                                       # R:769
env.venv.disturbance_ranges = TRAIN_DISTURBANCE_RANGES                                                                      # R:772-780
policy_kwargs = dict(activation_fn=torch.nn.ReLU, net_arch=[dict(pi=[120, 120, 120], vf=[120, 120, 120])], log_std_init=0)  # R:784
model = PPO("MlpPolicy", env, policy_kwargs=policy_kwargs, verbose=0, n_steps=1000, batch_size=5000, n_epochs=10, gamma=0.999,
            seed=a.seed, precision=a.precision)                                                                             # R:785-795
tr = model._trainer
assert tr.native_update and tr.fused_collect, "every precision runs on the hand-written kernels (collect + update)"

n_eval, G, dt = 4096, 4, 0.01
ev = Quadcopter3DGates(n_eval, *trk, gates_ahead=1, infos_mode="none", seed=99)
ev.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
ev.max_steps = 10 ** 6


@torch.no_grad()
def evaluate():
    """deterministic policy, 2 000 steps = 20 s of flight; a crash restarts that env's lap count (tools/train_ppo.py's protocol)"""
    ev.seed(99)
    obs = ev.reset_device()
    dev = obs.device
    gates12 = torch.zeros(n_eval, device=dev); crashes12 = torch.zeros(n_eval, device=dev)
    passed = torch.zeros(n_eval, device=dev); lap_start = torch.zeros(n_eval, device=dev)
    lap_sum = torch.zeros(7, device=dev); lap_cnt = torch.zeros(7, device=dev)
    for k in range(2000):
        obs, rew, done, trunc = ev.step_device(tr.act_device(obs).contiguous())
        t = (k + 1) * dt
        g = (rew > 5).float()
        if k < 1200:
            gates12 += g; crashes12 += (done.float() - trunc.float()).clamp(min=0)
        passed += g
        lap_done = (g > 0) & (passed % G == 0) & (passed > 0)
        lap_no = (passed / G).long().clamp(max=6)
        if lap_done.any():
            sel = lap_done & (passed / G <= 6)
            lap_sum.index_add_(0, lap_no[sel], (t - lap_start)[sel])
            lap_cnt.index_add_(0, lap_no[sel], torch.ones_like(lap_start)[sel])
            lap_start = torch.where(lap_done, torch.full_like(lap_start, t), lap_start)
        d = done.bool()
        passed = torch.where(d, torch.zeros_like(passed), passed)
        lap_start = torch.where(d, torch.full_like(lap_start, t), lap_start)
    laps = (lap_sum / lap_cnt.clamp(min=1)).tolist()
    fl = float(lap_sum[2:].sum() / lap_cnt[2:].sum().clamp(min=1)) if float(lap_cnt[2:].sum()) > 0 else None
    return dict(flying_lap=fl, first_lap=laps[1] if float(lap_cnt[1]) > 0 else None, gates_per_12s=float(gates12.mean()),
                crashes_per_12s=float(crashes12.mean()), laps_counted=lap_cnt[1:].tolist())


ckpt = os.path.join(tempfile.mkdtemp(prefix="refrecipe_"), "ckpt")
TIMESTEPS = model.n_steps * env.num_envs * 10                                   # R:818: a checkpoint every 10 policy rollouts
curve, paused, save_s, n_ckpt = [], 0.0, 0.0, 0
torch.cuda.synchronize()
t0 = time.perf_counter()
while model.num_timesteps < a.steps:
    model.learn(total_timesteps=TIMESTEPS, reset_num_timesteps=False)           # R:820
    s0 = time.perf_counter()
    model.save(ckpt)                                                            # R:823 (same file re-written: the run keeps one)
    save_s += time.perf_counter() - s0
    n_ckpt += 1
    if n_ckpt % a.eval_every == 0 or model.num_timesteps >= a.steps:
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        r = evaluate()
        r.update(train_seconds=e0 - t0 - paused, env_steps=int(model.num_timesteps))
        curve.append(r)
        torch.cuda.synchronize()
        paused += time.perf_counter() - e0
        if a.stop_when_reached and r["flying_lap"] is not None and r["flying_lap"] <= a.lap_target and r["crashes_per_12s"] <= 0.1:
            break
torch.cuda.synchronize()
train_s = time.perf_counter() - t0 - paused
ok = [c for c in curve if c["flying_lap"] is not None]
good = [c for c in ok if c["flying_lap"] <= a.lap_target and c["crashes_per_12s"] <= 0.1]
best = min(ok, key=lambda c: (c["crashes_per_12s"] > 0.1, c["flying_lap"])) if ok else None
res = dict(what="reference training cell (R:765, 783-795, 816-831): 100 envs x n_steps 1000, batch_size 5000, 10 epochs, gamma 0.999, constant lr "
                "3e-4, checkpoint every 10 rollouts, via optimal_quad_control_rl_amd.PPO.learn / .save; 4-gate square track",
           seed=a.seed, precision=a.precision, stopped_when_reached=bool(a.stop_when_reached and model.num_timesteps < a.steps), native_update=bool(tr.native_update), fused_collect=bool(tr.fused_collect), train_steps=int(model.num_timesteps), train_seconds=train_s,
           env_steps_per_s=model.num_timesteps / train_s, seconds_in_save=save_s, checkpoints=n_ckpt,
           updates_applied=tr.stats.get("updates"), updates_skipped_nonfinite=tr.stats.get("skipped_nonfinite", 0),
           final=curve[-1] if curve else None, best_checkpoint=best,
           reaches_reference_level_after_s=good[0]["train_seconds"] if good else None,
           reaches_reference_level_after_steps=good[0]["env_steps"] if good else None,
           reference_lap_seconds=dict(first=2.97, flying="2.51-2.59 (FP:3474-3488)"), curve=curve)
print(json.dumps({k: v for k, v in res.items() if k != "curve"}, indent=1))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
