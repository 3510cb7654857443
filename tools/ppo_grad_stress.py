#!/usr/bin/env python3
"""Two-waves-per-SIMD check of the PPO gradient kernel (chain and weight-gradient waves of a workgroup share SIMDs and both issue matrix
instructions): the same minibatch many times, every gradient compared bitwise with the first."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ppo_kernel as T
for L, B in ((24, 16384), (24, 65536), (17, 16384)):
    pol, ref, up, obs, act, old_lp, adv, ret = T._setup(L, rows=70000, seed=5, max_minibatch=65536)
    idx = torch.randperm(obs.shape[0], device=obs.device)[:B].to(torch.int32)
    g0 = up.grad(obs, act, old_lp, adv, ret, idx).clone()
    bad = 0
    for rep in range(200):
        g = up.grad(obs, act, old_lp, adv, ret, idx)
        bad += int((g.view(torch.int32) != g0.view(torch.int32)).sum())
    torch.cuda.synchronize()
    print(f"PPO gradient L={L} B={B}: {bad} differing values in 200 repeats (gradient of {g0.numel()} parameters)")
    up.close()
