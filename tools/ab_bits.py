#!/usr/bin/env python3
"""Digest of the env kernels' outputs for an A/B of two builds of the library (arithmetic-neutrality of a kernel change):
    python tools/ab_bits.py > a.txt ; QR_PROBE_LIB=optimal_quad_control_rl_amd/_dbg/libX.so python tools/ab_bits.py > b.txt ; diff a.txt b.txt
Fused rollouts (every kernel family the selection reaches) and the per-step kernel, both variants, through auto-resets."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
if os.environ.get("QR_PROBE_LIB"):
    B.LIB = os.path.join(ROOT, os.environ["QR_PROBE_LIB"]); B.needs_build = lambda: False
import torch
import bench


def dig(ts):
    h = hashlib.sha256()
    for t in ts:
        if t is not None:
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:20]


for variant in ("e2e", "indi"):
    for n, K in ((65536, 160), (4096, 300), (1000, 300), (262144, 60), (1 << 20, 24)):
        for ga in ((0, 1, 2) if n == 4096 else (1,)):
            env = bench.make_env(variant, n, ga, 0)
            env.max_steps = 150
            env.reset_device()
            g = torch.Generator(device="cuda").manual_seed(n + ga)
            acts = torch.rand((K, n, 4), device="cuda", generator=g) * 2 - 1
            out = env.rollout_device(acts)
            st = env.get_state_tensors()
            print(f"{variant} n={n} ga={ga} fused[{env.rollout_kernel_name()}] K={K}: out {dig(out)} state {dig(st)} resets {int(out[2].sum())}")
            if n <= 65536:
                env2 = bench.make_env(variant, n, ga, 0)
                env2.max_steps = 150
                env2.reset_device()
                buf = tuple(torch.empty_like(t) for t in out)
                env2.step_sequence_device(acts[:40].contiguous(), tuple(b[:40] for b in buf))
                print(f"{variant} n={n} ga={ga} per-step K=40: out {dig([b[:40] for b in buf])} state {dig(env2.get_state_tensors())}")
