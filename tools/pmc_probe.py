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
# closed-loop kernel (policy MLP + sampling + env step), K steps per launch
from optimal_quad_control_rl_amd.policy import MfmaPolicy
from optimal_quad_control_rl_amd.ppo import ActorCritic
torch.manual_seed(0)
pol = MfmaPolicy(L, 0).load_torch(ActorCritic(L, 4).cuda().pi)
res = None
for r in range(3):
    res = env.rollout_policy_device(pol, K, torch.zeros(4), noise_seed=0, first_step=r * K, out=None if res is None else res[:6])
torch.cuda.synchronize()
print("pmc probe done")
