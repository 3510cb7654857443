#!/usr/bin/env python3
"""Probe: per-step kernel as C concurrent chains over env slices (C handles of N / C envs on C streams) against one handle of N envs.
usage: slice_probe.py [e2e|indi] [N] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for C in (1, 2, 4, 8):
    n = N // C
    envs = [bench.make_env(variant, n, 1, 0) for _ in range(C)]
    streams = [torch.cuda.Stream() for _ in range(C)]
    acts = [torch.rand((K, n, 4), device="cuda") * 2 - 1 for _ in range(C)]
    outs = []
    for c in range(C):
        with torch.cuda.stream(streams[c]):
            envs[c].reset_device()
            outs.append(envs[c].rollout_device(acts[c]))
            envs[c].step_sequence_device(acts[c], outs[c])
    torch.cuda.synchronize()
    ts = []
    for r in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for c in range(C):
            with torch.cuda.stream(streams[c]):
                envs[c].step_sequence_device(acts[c], outs[c])
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K)
    t = float(np.median(ts[2:]))
    b = 285 if variant == "e2e" else 209
    print(f"{variant} N={N} as {C} x {n} envs on {C} streams: {t*1e6:.3f} us per step of all envs  {N/t/1e9:.2f} G env-steps/s  frac {N*b/t/8e12:.3f}")
    for e in envs: e.close()
