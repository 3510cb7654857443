#!/usr/bin/env python3
"""Un-drained phase stamps of ONE step (k = K/2) of the fused rollout kernel: where a wave IS at each stamp.
Builds (or reuses: QR_PROBE_NOBUILD=1) _dbg/libquadrace_ph<tag>.so with -DQR_PHASE_TIMING -DQR_PHASE_TIMING_NODRAIN.

    [QR_PROBE_TAG=_x QR_PROBE_CSRC=dir] python tools/phase_probe.py [e2e|indi] [envs] [--build]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 65536
tag = os.environ.get("QR_PROBE_TAG", "")
CSRC = os.environ.get("QR_PROBE_CSRC", B.CSRC)
dbg = os.path.join(B.PKG, "_dbg", "libquadrace_ph%s.so" % tag)
os.makedirs(os.path.dirname(dbg), exist_ok=True)
if not (os.environ.get("QR_PROBE_NOBUILD") == "1" and os.path.exists(dbg)):
    flags = ["-DQR_PHASE_TIMING", "-DQR_PHASE_TIMING_NODRAIN", "-DQR_GA_ONLY=1"]
    B.CSRC = CSRC
    B.build_native(extra_flags=tuple(flags), out=dbg)   # the product's own pipeline (assembly rewrite + lint)
if "--build" in sys.argv:
    sys.exit(0)
B.LIB = dbg
B.needs_build = lambda: False
import numpy as np  # noqa: E402
import torch  # noqa: E402
from optimal_quad_control_rl_amd import _lib  # noqa: E402
import bench  # noqa: E402

L = _lib.load()
L.qr_debug_set_ticks.argtypes = [C.c_void_p, C.c_void_p]
env = bench.make_env(variant, n, 1, 0)
n_waves = (n + 255) // 256 * 4
ticks = torch.zeros((n_waves, 16), dtype=torch.int64, device="cuda")
env.reset_device()
K = 64
acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
out = None
for _ in range(3):
    out = env.rollout_device(acts, out)
L.qr_debug_set_ticks(env._h, C.c_void_p(ticks.data_ptr()))
reps = []
for _ in range(8):
    out = env.rollout_device(acts, out)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[2:]
names = {2: "loop top", 3: "sincos/rot", 4: "mlp", 5: "eom+euler+reward", 6: "reset + small stores", 7: "observe + obs stores"}
slots = [s for s in range(2, 8) if (t[:, :, s] != 0).any()]
print(f"{variant} n={n} tag='{tag}' form={os.environ.get('QR_ROLLOUT_FORM','auto')}: one step (k = {K // 2}) of the fused loop, cycles (median over waves and launches; un-drained stamps cost ~100-200 cycles each)")
for a, b in zip(slots[:-1], slots[1:]):
    d = t[:, :, b] - t[:, :, a]
    print(f"  {a}->{b} {names[b]:24s} {np.median(d):8.0f}   (p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f})")
print(f"  step total {np.median(t[:, :, slots[-1]] - t[:, :, slots[0]]):8.0f}")
# per-step kernel: slots 0..7 (entry, loads issued, tables staged + barrier + weight registers, rot, mlp, eom+logic, reset+small stores, done)
ticks.zero_()
a = torch.rand((n, 4), device="cuda") * 2 - 1
reps = []
for _ in range(24):
    env.step_device(a)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[4:]
pn = ["entry", "loads issued", "tables in LDS, barrier, weight regs", "sincos/rot", "mlp", "eom+euler+reward", "reset + small stores", "observe + obs stores"]
print("per-step kernel (one launch = one step), cycles between stamps:")
for sl in range(1, 8):
    d = t[:, :, sl] - t[:, :, sl - 1]
    print(f"  {sl-1}->{sl} {pn[sl]:38s} {np.median(d):8.0f}   (p10 {np.percentile(d, 10):6.0f}  p90 {np.percentile(d, 90):6.0f})")
print(f"  in-wave total {np.median(t[:, :, 7] - t[:, :, 0]):8.0f};  first entry -> last exit {np.median((t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1))):8.0f}")
