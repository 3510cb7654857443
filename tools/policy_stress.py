#!/usr/bin/env python3
"""Two-waves-per-SIMD check of the stand-alone policy forward kernel (80 KB of LDS: two workgroups per CU once the grid is large):
the same 1 Mi-row batch several times in ONE launch each, against the same rows in launches of 65 536 (one workgroup per CU), bitwise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd.policy import MfmaPolicy
L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << 20
g = torch.Generator(device="cuda").manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(L, 120), torch.nn.ReLU(), torch.nn.Linear(120, 120), torch.nn.ReLU(), torch.nn.Linear(120, 120), torch.nn.ReLU(), torch.nn.Linear(120, 4))
p = MfmaPolicy(L).load_torch(net)
obs = torch.randn((n, L), device="cuda", generator=g)
ref = torch.empty((n, 4), device="cuda")
for c in range(0, n, 65536):
    p.forward(obs[c:c + 65536], ref[c:c + 65536]); torch.cuda.synchronize()
bad = 0
for rep in range(20):
    out = p.forward(obs); torch.cuda.synchronize()
    neq = (out.view(torch.int32) != ref.view(torch.int32))
    c = int(neq.sum()); bad += c
    if c: print("rep", rep, "mismatching values", c, "first rows", neq.any(1).nonzero().flatten()[:8].tolist())
print(f"policy forward L={L}: {bad} mismatching values in 20 launches of {n} rows against 16 x 65536-row launches")
