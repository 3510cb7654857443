#!/usr/bin/env python3
"""Policy forward at N envs: hand-written MFMA kernel vs torch float32 (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.policy import MfmaPolicy
from optimal_quad_control_rl_amd.ppo import ActorCritic

for n in (4096, 65536, 1048576):
    net = ActorCritic(24, 4).cuda()
    obs = torch.randn(n, 24, device="cuda")
    pol = MfmaPolicy(24).load_torch(net.pi)
    out = torch.empty(n, 4, device="cuda")
    def t(fn, reps=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    with torch.no_grad():
        a = t(lambda: pol.forward(obs, out)); b = t(lambda: net.pi(obs))
    flops = 2 * n * (24 * 120 + 120 * 120 * 2 + 120 * 4)
    print(f"n={n:8d}  mfma kernel {a:8.1f} us ({flops/a/1e6:7.1f} TF/s useful)   torch f32 {b:8.1f} us   speed-up {b/a:5.1f}x")
