#!/usr/bin/env python3
"""BASELINE config 5: PPO on the MI355X race env (GPU box).

    python tools/train_ppo.py [--variant indi|e2e] [--envs 65536] [--steps 3e8] [--track square|zigzag]

Prints training progress and a final deterministic evaluation on the 4-gate square track: gates per 12 s from a standing
start, and lap times the way the reference tabulates them (FP:3474-3488: lap 1 from the start, then the flying laps; its
simulated E2E policy flies 2.97 s then 2.51-2.59 s, INDI 3.20 s then 2.75-2.82 s)."""
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
ap.add_argument("--precision", default="f16-operands", choices=("f16-operands", "f32-collect", "f32"),
                help="f32: the reference-precision kernels for collect, values and update (needs --fused --native-update); f32-collect: collect only")
ap.add_argument("--ent-coef", type=float, default=0.0)
ap.add_argument("--gamma", type=float, default=0.999)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--no-trunc-bootstrap", action="store_true", help="round-1 behaviour: a time-limit truncation is a termination")
ap.add_argument("--eval-final", action="store_true", help="evaluate the FINAL policy (no best-by-training-statistic checkpoint)")
ap.add_argument("--save", default="", help="save an SB3-shaped checkpoint (optimal_quad_control_rl_amd.sb3 format) of the final model here")
ap.add_argument("--curve", type=int, default=0, help="evaluate the current policy every this many rollouts (training clock stopped)")
ap.add_argument("--lap-target", type=float, default=2.6, help="flying lap (s) that counts as the reference's level for --curve")
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
            native_update=a.native_update, truncation_bootstrap=not a.no_trunc_bootstrap,
            policy_forward="f32class" if a.precision != "f16-operands" else "torch", update_precision="f32" if a.precision == "f32" else "f16-operands")
if "RANK" in os.environ:
    model._updater.broadcast_parameters(0)   # identical start on every rank (the seed already makes it so; this guarantees it)
    model.noise_seed = a.seed                 # same key, different global env ids -> independent action noise per rank
# deterministic evaluation on a separate env: 2000 steps = 20 s of flight (six laps), crashes auto-reset and restart the lap count
n_eval = 4096
ev = cls(n_eval, *trk, gates_ahead=1, infos_mode="none", seed=99)
if a.variant == "e2e":
    ev.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
ev.max_steps = 10 ** 6
G = 4 if a.track == "square" else len(trk[0])   # square_track() lists its four gates twice: a lap is four passes
dt = 0.01


@torch.no_grad()
def evaluate(m):
    ev.seed(99)
    obs = ev.reset_device()
    dev = obs.device
    gates12 = torch.zeros(n_eval, device=dev); crashes12 = torch.zeros(n_eval, device=dev)
    passed = torch.zeros(n_eval, device=dev)            # gates passed since this env's last (re)start
    lap_start = torch.zeros(n_eval, device=dev)         # time of the last lap boundary (or restart)
    lap_sum = torch.zeros(7, device=dev); lap_cnt = torch.zeros(7, device=dev)   # laps 1..6 (index 0 unused)
    for k in range(2000):
        obs, rew, done, trunc = ev.step_device(m.act_device(obs).contiguous())
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
    return dict(eval_gates_per_12s=float(gates12.mean()), eval_crashes_per_12s=float(crashes12.mean()),
                eval_seconds_per_gate=float(1200 * dt / gates12.mean().clamp(min=1e-9)),
                eval_seconds_per_lap_4gates=float(4 * 1200 * dt / gates12.mean().clamp(min=1e-9)),
                eval_lap_seconds={f"lap{i}": laps[i] for i in range(1, 7)}, eval_laps_counted=lap_cnt[1:].tolist(),
                eval_flying_lap_seconds=float(lap_sum[2:].sum() / lap_cnt[2:].sum().clamp(min=1)))


best = {"gates": -1.0, "state": None}
def keep_best(m):
    g = m.stats.get("gates_per_episode", 0.0)
    if g > best["gates"] and m.stats.get("ep_len_mean", 0) > 600:
        best["gates"] = g
        best["state"] = {k: v.clone() for k, v in m.policy.state_dict().items()}
# --curve K: every K rollouts the CURRENT policy is evaluated (clock stopped) -> wall-clock-to-quality curve (BASELINE config 5's metric)
curve, paused, it_no = [], [0.0], [0]
def on_rollout(m):
    if not a.eval_final:
        keep_best(m)
    it_no[0] += 1
    if a.curve and it_no[0] % a.curve == 0 and rank == 0:
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        tsec = e0 - t0 - paused[0]
        r = evaluate(m)
        torch.cuda.synchronize()
        paused[0] += time.perf_counter() - e0
        curve.append(dict(train_seconds=tsec, env_steps=m.num_timesteps * world,
                          flying_lap=r["eval_flying_lap_seconds"], crashes_per_12s=r["eval_crashes_per_12s"], gates_per_12s=r["eval_gates_per_12s"]))
t0 = time.perf_counter()
model.learn(int(a.steps) // world, log_every=20, callback=on_rollout)   # --steps counts env-steps of the whole job
if best["state"] is not None and not a.eval_final:
    model.policy.load_state_dict(best["state"])  # evaluate the best checkpoint (the reference saves one every 10 rollouts, R:823)
torch.cuda.synchronize()
train_s = time.perf_counter() - t0 - paused[0]

final = evaluate(model)
res = dict(evaluated="final policy" if a.eval_final else "best checkpoint by training statistic", lr_final_frac=a.lr_final,
           world_size=world, fused_collect=a.fused, native_update=a.native_update, variant=a.variant, track=a.track, envs=a.envs,
           gamma=a.gamma, seed=a.seed, n_steps=a.n_steps, epochs=a.epochs, minibatches=a.minibatches, lr=a.lr, target_kl=a.target_kl,
           truncation_bootstrap=not a.no_trunc_bootstrap,
           train_steps=model.num_timesteps * world, train_seconds=train_s,
           train_Msteps_per_s=model.num_timesteps * world / train_s / 1e6,
           **final, **model.stats)
sk, up_n = res.get("skipped_nonfinite", 0), res.get("updates", 0)
if sk > 0.002 * max(1, sk + up_n):   # a handful of skips = diverged sims in a minibatch; more means something is wrong
    print(f"WARNING: {sk} of {sk + up_n} minibatch updates were skipped for a non-finite gradient norm", file=sys.stderr, flush=True)
if curve:
    # first time the policy flies laps like the reference's (simulated flying laps 2.51-2.59 s, FP:3474-3488) without crashing
    hit = [c for c in curve if 0 < c["flying_lap"] <= a.lap_target and c["crashes_per_12s"] <= 0.1]
    res["curve"] = curve
    res["seconds_to_reference_lap"] = hit[0]["train_seconds"] if hit else None
    res["env_steps_to_reference_lap"] = hit[0]["env_steps"] if hit else None
    res["lap_target"] = a.lap_target
print(json.dumps(res))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
