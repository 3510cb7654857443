#!/usr/bin/env python3
"""Profiling aid: where a PPO minibatch update's WALL time goes across the two launches (instrumented build, -DQR_PHASE_TIMING):
every workgroup of ppo_grad_kernel and ppo_apply_kernel stamps the device-wide 100 MHz clock at entry / exit (apply: at its stages).
Prints, relative to the first workgroup of the gradient kernel: start skew, last exit, the gap to the apply kernel, the apply's stages.
Usage (GPU box): python tools/ppo_launch_timing.py [obs_len] [minibatch]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
extra = os.environ.get("QR_TICK_EXTRA_FLAGS", "").split()
dbg = os.path.join(ROOT, "optimal_quad_control_rl_amd", "_dbg", "libquadrace_dbg_nodrain%s.so" % ("_x" if extra else ""))
os.makedirs(os.path.dirname(dbg), exist_ok=True)
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
if "--build-only" in sys.argv or not os.path.exists(dbg) or os.path.getmtime(dbg) < max(os.path.getmtime(f) for f in srcs):
    B.build_native(extra_flags=("-DQR_PHASE_TIMING", "-DQR_PHASE_TIMING_NODRAIN", *extra), out=dbg, drop_flags=("-mllvm", "-amdgpu-mfma-vgpr-form"))
if "--build-only" in sys.argv:
    sys.exit(0)
B.LIB = dbg
B.needs_build = lambda: False
from optimal_quad_control_rl_amd import _lib
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
args = [a for a in sys.argv[1:] if not a.startswith("--")]
L_ = int(args[0]) if args else 24
Bn = int(args[1]) if len(args) > 1 else 16384
dev = torch.device("cuda", 0)
R = max(65536 * 4, 16 * Bn)
obs = torch.randn((R, L_), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
perm = torch.randperm(R, device=dev).to(torch.int32)
pol = ActorCritic(L_, 4).to(dev)
up = MfmaPpoUpdater(pol, L_, dev, Bn)
lib = _lib.load()
for fn in (lib.qr_ppo_debug_set_ticks, lib.qr_ppo_debug_set_apply_ticks):
    fn.argtypes = [C.c_void_p, C.c_void_p]
wgs = 2 * min(128, Bn // 128)
gt = torch.zeros((wgs * 8, 16), dtype=torch.int64, device=dev)
at = torch.zeros((512, 8), dtype=torch.int64, device=dev)
for k in range(5):
    up.minibatch(obs, act, old_lp, adv, ret, perm[k * Bn:(k + 1) * Bn], 3e-4)
lib.qr_ppo_debug_set_ticks(up._h, C.c_void_p(gt.data_ptr()))
lib.qr_ppo_debug_set_apply_ticks(up._h, C.c_void_p(at.data_ptr()))
rows = []
K = 14
for rep in range(K):
    at.zero_()
    torch.cuda.synchronize()
    # two back-to-back updates: the stamps that remain are the SECOND one's (its gradient kernel started right behind an apply)
    up.minibatch(obs, act, old_lp, adv, ret, perm[(2 * rep) * Bn % (R - Bn):][:Bn], 3e-4)
    up.minibatch(obs, act, old_lp, adv, ret, perm[(2 * rep + 1) * Bn % (R - Bn):][:Bn], 3e-4)
    torch.cuda.synchronize()
    g = gt.cpu().numpy().astype(np.int64); a = at.cpu().numpy().astype(np.int64)
    a = a[a[:, 0] > 0]
    g0 = g[:, 14].min()
    us = lambda x: (x - g0) / 100.0   # noqa: E731  (100 MHz -> us)
    rows.append(dict(grad_last_start=us(g[:, 14].max()), grad_first_exit=us(g[:, 15][g[:, 15] > 0].min()), grad_last_exit=us(g[:, 15].max()),   # (only the dW waves stamp their exit)
                     apply_first_start=us(a[:, 0].min()), apply_last_start=us(a[:, 0].max()),
                     apply_reduced_med=us(np.median(a[:, 1])), apply_reduced_last=us(a[:, 1].max()),
                     apply_arrived_last=us(a[:, 2].max()), apply_released_first=us(a[:, 3].min()), apply_released_last=us(a[:, 3].max()),
                     apply_last_exit=us(a[:, 4].max()), apply_wgs=len(a),
                     apply_reduced_p90=us(np.percentile(a[:, 1], 90)), apply_reduced_owner=us(a[-1, 1]), apply_start_owner=us(a[-1, 0]),
                     apply_slowest_reduced_wg=int(a[:, 1].argmax()),
                     # gradient kernel by network (grid.y: 0 = policy, 1 = value): exit of the last dW wave of each workgroup
                     grad_exit_policy_med=us(np.median(g[:len(g) // 2].reshape(-1, 8, 16)[:, 4:, 15].max(1))),
                     grad_exit_policy_last=us(g[:len(g) // 2, 15].max()),
                     grad_exit_value_med=us(np.median(g[len(g) // 2:].reshape(-1, 8, 16)[:, 4:, 15].max(1))),
                     grad_exit_value_last=us(g[len(g) // 2:, 15].max())))
print(f"one minibatch update, obs_len {L_}, {Bn} rows; wall-clock us since the FIRST workgroup of the gradient kernel started (median of {K - 4} updates)")
for k in rows[0]:
    print(f"  {k:24s} {np.median([r[k] for r in rows[4:]]):8.2f}")
# distribution of the gradient workgroups' start / exit over the grid (last repetition): percentiles per network and by workgroup id % 8
g8 = g.reshape(-1, 8, 16)
start = (g8[:, :, 14].min(1) - g0) / 100.0
exit_ = (g8[:, 4:, 15].max(1) - g0) / 100.0
half = len(g8) // 2
for name, sl in (("policy", slice(0, half)), ("value", slice(half, None))):
    s_, e_ = start[sl], exit_[sl]
    pct = lambda x: " ".join(f"{np.percentile(x, p):6.2f}" for p in (0, 10, 50, 90, 100))   # noqa: E731
    print(f"  {name:6s} workgroups: start p0/10/50/90/100 {pct(s_)} | exit {pct(e_)} | duration {pct(e_ - s_)}")
    by8 = [np.median((e_ - s_)[i::8]) for i in range(8)]
    print(f"         median duration by workgroup id % 8: " + " ".join(f"{x:6.2f}" for x in by8))
    order = np.argsort(e_)
    print(f"         earliest exits: ids {order[:6].tolist()}  latest: ids {order[-6:].tolist()}")
