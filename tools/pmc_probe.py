#!/usr/bin/env python3
"""Workload for the PMC passes (run under rocprofv3 --pmc ...): a calibration copy of known size followed by
per-step launches and one fused rollout of the bench workload.  Kept small: counters serialise kernels."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = 64
# calibration: float4-wide copy of 256 MiB (read 256 MiB + write 256 MiB), 4 times
src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
for _ in range(4):
    dst.copy_(src)
torch.cuda.synchronize()
env = bench.make_env(variant, n, 1, 0)
L = env.state_len
gen = torch.Generator(device="cuda").manual_seed(0)
actions = torch.rand((K, n, 4), device="cuda", generator=gen) * 2 - 1
out = (torch.empty((K, n, L), device="cuda"), torch.empty((K, n), device="cuda"),
       torch.empty((K, n), dtype=torch.uint8, device="cuda"), torch.empty((K, n), dtype=torch.uint8, device="cuda"))
env.reset_device()
env.step_sequence_device(actions, out)
env.rollout_device(actions, out)
torch.cuda.synchronize()
print("pmc probe done")
