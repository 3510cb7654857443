#!/usr/bin/env python3
"""Digest of the POLICY kernels' outputs for an A/B of two builds of the library (arithmetic-neutrality of a kernel change):
    python tools/ab_policy_bits.py > a.txt ; QR_PROBE_LIB=optimal_quad_control_rl_amd/_dbg/libX.so python tools/ab_policy_bits.py > b.txt ; diff a.txt b.txt
Standalone forward (f16-operand and f32-class) at every observation length and ragged row counts; closed-loop rollouts (policy + sampling +
env step in one kernel) at both precisions, both variants, through auto-resets.  Also prints the time of the f32-class forms."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
if os.environ.get("QR_PROBE_LIB"):
    B.LIB = os.path.join(ROOT, os.environ["QR_PROBE_LIB"]); B.needs_build = lambda: False
import torch
import bench
from optimal_quad_control_rl_amd.policy import MfmaPolicy
from optimal_quad_control_rl_amd.ppo import ActorCritic


def dig(ts):
    h = hashlib.sha256()
    for t in ts:
        if t is not None:
            h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:20]


dev = torch.device("cuda", 0)
for L in (13, 17, 20, 21, 24, 25, 28, 29, 32, 36):
    torch.manual_seed(L)
    net = ActorCritic(L, 4).to(dev)
    pol = MfmaPolicy(L, 0).load_torch(net.pi)
    for n in (1, 63, 64, 1000, 65536):
        obs = torch.randn((n, L), device=dev) * 2
        print(f"forward L={L} n={n}: f16 {dig([pol.forward(obs)])} f32class {dig([pol.forward(obs, precision='f32')])}")
    if L == 24:
        obs = torch.randn((65536, L), device=dev)
        out = torch.empty((65536, 4), device=dev)
        for prec in ("f16-operands", "f32"):
            for _ in range(5): pol.forward(obs, out, precision=prec)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): pol.forward(obs, out, precision=prec)
            torch.cuda.synchronize()
            print(f"# time forward 65536 rows {prec}: {(time.perf_counter() - t0) / 50 * 1e6:.2f} us", file=sys.stderr)
for variant in ("e2e", "indi"):
    for n, K in ((4096, 120), (100, 300), (65536, 40)):
        for prec in ("f16-operands", "f32"):
            env = bench.make_env(variant, n, 1, 0)
            env.max_steps = 100
            env.reset_device()
            L = env.state_len
            torch.manual_seed(7)
            net = ActorCritic(L, 4).to(dev)
            pol = MfmaPolicy(L, 0).load_torch(net.pi)
            out = env.rollout_policy_device(pol, K, net.log_std.detach(), noise_seed=3, first_step=0, precision=prec)
            print(f"closed loop {variant} n={n} K={K} {prec}: out {dig(out)} state {dig(env.get_state_tensors())}")
            if n == 65536:
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for r in range(3): env.rollout_policy_device(pol, K, net.log_std.detach(), noise_seed=3, first_step=(r + 1) * K, out=tuple(out[:6]), precision=prec)
                torch.cuda.synchronize()
                print(f"# time closed loop {variant} 65536 envs {prec}: {(time.perf_counter() - t0) / 3 / K * 1e6:.2f} us per step", file=sys.stderr)
