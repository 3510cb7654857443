#!/usr/bin/env python3
"""Where a wave IS at each phase boundary of ONE step of the fused rollout loop, un-drained (-DQR_PHASE_TIMING -DQR_PHASE_TIMING_NODRAIN:
the stamps only wait for the scalar / LDS queue that s_memtime returns through, the stores keep streaming): the overlapped schedule as
it runs.  Usage (GPU box): python tools/fused_phase_probe.py [e2e|indi] [num_envs] [K]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dbg = os.environ.get("QR_PHASE_LIB") or os.path.join(ROOT, "optimal_quad_control_rl_amd", "_dbg", "libquadrace_phase_nodrain.so")
os.makedirs(os.path.dirname(dbg), exist_ok=True)
if not os.environ.get("QR_NO_BUILD"):
    B.build_native(extra_flags=("-DQR_PHASE_TIMING", "-DQR_PHASE_TIMING_NODRAIN"), out=dbg)
B.LIB = dbg
B.needs_build = lambda: False
from optimal_quad_control_rl_amd import _lib  # noqa: E402
import bench  # noqa: E402

env = bench.make_env(variant, n, 1, 0)
L = _lib.load()
L.qr_debug_set_ticks.argtypes = [C.c_void_p, C.c_void_p]
n_waves = (n + 255) // 256 * 4
ticks = torch.zeros((n_waves, 16), dtype=torch.int64, device="cuda")
env.reset_device()
acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
out = env.rollout_device(acts)
L.qr_debug_set_ticks(env._h, C.c_void_p(ticks.data_ptr()))
reps = []
for _ in range(12):
    env.rollout_device(acts, out)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[3:]
names = {3: "sincos / rotation", 4: "body velocity + residual MLPs", 5: "EoM, Euler, reward / termination (+ obs stores of the previous step)",
         6: "state update, reset, reward / done stores", 7: "gate row, observation, tile write"}
print(f"{variant} n={n} K={K} [{env.rollout_kernel_name()}]: step K/2 of the fused loop, cycles between un-drained stamps (median / p90 over waves and launches)")
for s in range(3, 8):
    d = (t[:, :, s] - t[:, :, s - 1]).reshape(-1)
    print(f"  {s} {names[s]:78s} {np.median(d):7.0f} {np.percentile(d, 90):7.0f}")
tot = (t[:, :, 7] - t[:, :, 2]).reshape(-1)
print(f"  step total (stamp 2 -> 7) {np.median(tot):7.0f} {np.percentile(tot, 90):7.0f}   (+ loop overhead; the stamps themselves cost ~6 x 50 cycles)")
