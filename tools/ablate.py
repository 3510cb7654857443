#!/usr/bin/env python3
"""Ablation timings of the fused rollout kernel (GPU box): python tools/ablate.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd import Quadcopter3DGates, Quadcopter3DGatesINDI, zigzag_track, square_track, TRAIN_DISTURBANCE_RANGES

n, K = 65536, 500
def run(name, variant, **kw):
    cls = Quadcopter3DGates if variant == "e2e" else Quadcopter3DGatesINDI
    trk = zigzag_track() if variant == "e2e" else square_track()
    env = cls(n, *trk, gates_ahead=1, infos_mode="none", **kw)
    if variant == "e2e":
        env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
    env.reset_device()
    gen = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.rand((K, n, 4), device="cuda", generator=gen) * 2 - 1
    out = env.rollout_device(acts)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout_device(acts, out); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    ts2 = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.step_sequence_device(acts, out); torch.cuda.synchronize()
        ts2.append((time.perf_counter() - t0) / K * 1e6)
    print(f"{name:38s} fused {sorted(ts)[2]:6.2f} us/step   step-launch {sorted(ts2)[1]:6.2f} us/step   done_frac {out[2].float().mean().item():.4f}")

run("e2e default", "e2e")
run("e2e no auto-reset (pause_if_collision)", "e2e", pause_if_collision=True)
run("e2e no residual MLP", "e2e", residual=None)
run("e2e no residual, no reset", "e2e", residual=None, pause_if_collision=True)
run("indi default", "indi")
run("indi no auto-reset", "indi", pause_if_collision=True)
