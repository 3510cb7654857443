#!/usr/bin/env python3
"""Per-step kernel time (qr_step_launches: K step kernels as one replayed graph) of a given build of the library, for A/B runs in ONE
GPU call:   QR_PROBE_LIB=optimal_quad_control_rl_amd/_dbg/libX.so python tools/step_probe.py [e2e|indi] [envs]"""
import os, sys
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
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
env = bench.make_env(variant, n, 1, 0)
K = 1000
acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
L = env.state_len
out = (torch.empty((K, n, L), device="cuda"), torch.empty((K, n), device="cuda"), torch.empty((K, n), dtype=torch.uint8, device="cuda"),
       torch.empty((K, n), dtype=torch.uint8, device="cuda"))
env.reset_device()
ts = []
for r in range(8):
    env.step_sequence_device(acts, out)
    torch.cuda.synchronize()
    ts.append(env.last_rollout_ms() / K * 1e3)
print(f"{os.environ.get('QR_PROBE_LIB', 'default lib'):60s} {variant} n={n}: per-step kernel {np.median(ts[2:]):.3f} us (min {min(ts[2:]):.3f})")
