#!/usr/bin/env python3
"""Soak: per-call latency of the PPO update paths (one synchronised call at a time) -- looks for stalls that an average hides.
usage: ppo_latency_soak.py [rows_per_minibatch] [calls]   prints median / p99 / max and the outliers (> 3 x median) with their index."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
L, R = 24, 65536 * 32
dev = torch.device("cuda", 0)
torch.manual_seed(0)
obs = torch.randn((R, L), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
perm = torch.randperm(R, device=dev).to(torch.int32)
up = MfmaPpoUpdater(ActorCritic(L, 4).to(dev), L, dev, B)
nb = R // B
out = {}
def soak(name, fn, n, per):
    for k in range(20): fn(k)
    torch.cuda.synchronize()
    t = np.empty(n)
    for k in range(n):
        t0 = time.perf_counter(); fn(k); torch.cuda.synchronize(); t[k] = (time.perf_counter() - t0) / per * 1e6
    med = float(np.median(t)); bad = [(int(i), round(float(t[i]), 1)) for i in np.nonzero(t > 3 * med)[0]]
    out[name] = {"calls": n, "us_median": round(med, 2), "us_p99": round(float(np.percentile(t, 99)), 2), "us_max": round(float(t.max()), 2), "outliers_gt_3x_median": bad[:20], "n_outliers": len(bad)}
def native(k):
    if k % nb == 0: up.begin_epoch(adv, perm, B)
    up.minibatch(obs, act, old_lp, adv, ret, perm[(k % nb) * B:(k % nb + 1) * B], 3e-4)
perm_dev = perm[:nb * B].clone()
def epoch(k):
    up.epoch(obs, act, old_lp, adv, ret, perm_dev, B, 3e-4, device_shuffle=True)
soak("minibatch_stream_launches", native, N, 1)
soak("epoch_graph", epoch, max(20, N // nb * 4), nb)
st = up.status()
out["status"] = {"skipped_nonfinite": st[2], "barrier_timeouts": st[3]}
print(json.dumps(out))
