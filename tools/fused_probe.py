#!/usr/bin/env python3
"""Fused K-step rollout throughput of a given build of the library, for A/B runs in ONE GPU call:
    QR_PROBE_LIB=optimal_quad_control_rl_amd/_dbg/libX.so python tools/fused_probe.py [e2e|indi] [envs] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
if os.environ.get("QR_PROBE_LIB"):
    B.LIB = os.path.join(ROOT, os.environ["QR_PROBE_LIB"])
    B.needs_build = lambda: False
import numpy as np
import torch
import bench

variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
env = bench.make_env(variant, n, 1, 0)
acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
env.reset_device()
out = env.rollout_device(acts)
ts = []
for r in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        env.rollout_device(acts, out)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 / K)
t = float(np.median(ts[2:]))
print(f"{os.environ.get('QR_PROBE_LIB', 'default lib'):58s} {variant} n={n} K={K}: {t*1e6:.2f} us/step  {n/t/1e9:.2f} G env-steps/s  [{env.rollout_kernel_name()}]")
