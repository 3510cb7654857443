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

# closed-loop rollout: policy + sampling + env step fused, vs the separate-launch and torch loops
import bench
K = 256
for variant in ("e2e", "indi"):
    env = bench.make_env(variant, 65536, 1, 0)
    env.reset_device()
    net = ActorCritic(env.state_len, 4).cuda()
    pol = MfmaPolicy(env.state_len).load_torch(net.pi)
    log_std = torch.zeros(4)
    out = None
    ts = []
    for r in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = env.rollout_policy_device(pol, K, log_std, noise_seed=1, first_step=r * K)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
    fused = sorted(ts)[1]
    def sep_loop():
        o = env.states_tensor
        for k in range(64):
            m = pol.forward(o)
            o, _, _, _ = env.step_device(m.clamp(-1, 1))
    def torch_loop():
        o = env.states_tensor
        with torch.no_grad():
            for k in range(64):
                a, lp, v = net.act(o)
                o, _, _, _ = env.step_device(a.clamp(-1, 1).contiguous())
    def tm(fn):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 64 * 1e6
    print(f"{variant}: closed-loop fused {fused:6.2f} us/step ({65536/fused/1e3:6.2f} G env-steps/s) | policy kernel + step kernel {tm(sep_loop):7.1f} us/step | torch policy(+value) + step kernel {tm(torch_loop):7.1f} us/step")
