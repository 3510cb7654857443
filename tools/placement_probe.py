#!/usr/bin/env python3
"""Probe: does the run-to-run bimodality of the large-N INDI rollout (52 vs 64 G env-steps/s on the same build) follow the PLACEMENT of
the caller's buffers?  One process, one handle; the action / output buffers are re-allocated behind dummy allocations of varying size.
usage: placement_probe.py [e2e|indi] [envs] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
variant = sys.argv[1] if len(sys.argv) > 1 else "indi"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
env = bench.make_env(variant, n, 1, 0)
env.reset_device()
keep = []
for trial, pad_mb in enumerate([0, 0, 1, 3, 7, 16, 33, 64, 0, 129, 5, 0]):
    if pad_mb:
        keep.append(torch.empty(pad_mb * 1024 * 1024 + 4096 * trial, dtype=torch.uint8, device="cuda"))
    acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
    out = env.rollout_device(acts)
    ts = []
    for r in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout_device(acts, out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K)
    t = float(np.median(ts[1:]))
    ptrs = [acts.data_ptr()] + [o.data_ptr() for o in out]
    print(f"trial {trial:2d} pad {pad_mb:4d} MB: {t*1e6:7.2f} us/step {n/t/1e9:6.2f} G env-steps/s  spread {min(ts[1:])*1e6:.2f}-{max(ts[1:])*1e6:.2f}  ptr>>21 mod 64: " + " ".join(f"{(p >> 21) % 64:2d}" for p in ptrs) + "  ptr mod 2MiB (KiB): " + " ".join(f"{(p % (1 << 21)) >> 10:4d}" for p in ptrs))
    del acts, out
