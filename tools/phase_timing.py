#!/usr/bin/env python3
"""Profiling aid: builds a SEPARATE instrumented copy of the library (-DQR_PHASE_TIMING) and prints where a
wave's time goes inside the step kernel (shader-clock stamps at phase boundaries, lane 0 of every wave).
The instrumentation drains the memory queues at every stamp, so it measures the dependency chain, not
the overlapped schedule.  Usage (GPU box):  python tools/phase_timing.py [e2e|indi] [num_envs]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dbg = os.path.join(ROOT, "gpurun_out", "libquadrace_dbg.so")
os.makedirs(os.path.dirname(dbg), exist_ok=True)
B.build_native(extra_flags=("-DQR_PHASE_TIMING",), out=dbg)   # the product's own pipeline (assembly rewrite + lint)
B.LIB = dbg  # make the adapter load the instrumented build
B.needs_build = lambda: False
from optimal_quad_control_rl_amd import _lib  # noqa: E402
import bench  # noqa: E402

env = bench.make_env(variant, n, 1, 0)
L = _lib.load()
L.qr_debug_set_ticks.argtypes = [C.c_void_p, C.c_void_p]
n_waves = (n + 255) // 256 * 4
ticks = torch.zeros((n_waves, 16), dtype=torch.int64, device="cuda")
env.reset_device()
a = torch.rand((n, 4), device="cuda") * 2 - 1
for _ in range(5):
    env.step_device(a)
L.qr_debug_set_ticks(env._h, C.c_void_p(ticks.data_ptr()))
reps = []
for _ in range(20):
    env.step_device(a)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[5:]  # [rep, wave, slot]
names = ["entry", "loads done", "tables staged+barrier", "sincos/rot", "mlp", "eom+logic", "reset+pre-store", "obs+stores done"]
t0 = t[:, :, 0].min(axis=1, keepdims=True)  # first wave entry per launch
print(f"{variant} n={n} waves={n_waves}: cycles since first wave entry (median over waves & launches), and per-phase delta")
prev = None
for s in range(1, 8):
    d = np.median(t[:, :, s] - t[:, :, s - 1])
    print(f"  {s} {names[s]:24s} delta {d:8.0f} cycles")
print(f"  in-wave total {np.median(t[:, :, 7] - t[:, :, 0]):8.0f} cycles")
# fused rollout kernel: one instrumented iteration (k = K/2): slots 2..7
K = 32
acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
ticks.zero_()
reps = []
for _ in range(10):
    env.rollout_device(acts)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[3:]
print(f"fused rollout kernel, one step of the K={K} loop:")
for s in range(3, 8):
    d = np.median(t[:, :, s] - t[:, :, s - 1])
    print(f"  {s} {names[s]:24s} delta {d:8.0f} cycles")
print(f"  per-step in-wave total {np.median(t[:, :, 7] - t[:, :, 2]):8.0f} cycles (instrumented: queues drained at every stamp)")

# closed-loop rollout kernel: one instrumented iteration, slots 8..13
from optimal_quad_control_rl_amd.policy import MfmaPolicy
from optimal_quad_control_rl_amd.ppo import ActorCritic
net = ActorCritic(env.state_len, 4).cuda()
pol = MfmaPolicy(env.state_len).load_torch(net.pi)
ticks.zero_()
reps = []
for r in range(8):
    env.rollout_policy_device(pol, 32, torch.zeros(4), noise_seed=1, first_step=32 * r)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[2:]
cl = ["", "policy forward", "sampling", "obs/act/logp stores", "env step (+small stores)", "observe"]
print("closed-loop rollout kernel, one step:")
for s_ in range(9, 14):
    print(f"  {cl[s_ - 8]:26s} delta {np.median(t[:, :, s_] - t[:, :, s_ - 1]):8.0f} cycles")
print(f"  per-step in-wave total {np.median(t[:, :, 13] - t[:, :, 8]):8.0f} cycles")
