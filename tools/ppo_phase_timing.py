#!/usr/bin/env python3
"""Profiling aid: where a wave's time goes inside ppo_phase_a_kernel (instrumented build, -DQR_PHASE_TIMING; queues are
drained at every stamp, so this measures the dependency chain).  Usage (GPU box): python tools/ppo_phase_timing.py"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
nodrain = os.environ.get("QR_TICK_NODRAIN") == "1"
extra = os.environ.get("QR_TICK_EXTRA_FLAGS", "").split()
dbg = os.path.join(ROOT, "optimal_quad_control_rl_amd", "_dbg", "libquadrace_dbg%s%s.so" % ("_nodrain" if nodrain else "", "_x" if extra else ""))   # travels with the snapshot (git-ignored)
os.makedirs(os.path.dirname(dbg), exist_ok=True)
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
if "--build-only" in sys.argv or not os.path.exists(dbg) or os.path.getmtime(dbg) < max(os.path.getmtime(f) for f in srcs):
    B.build_native(extra_flags=("-DQR_PHASE_TIMING", *(["-DQR_PHASE_TIMING_NODRAIN"] if nodrain else []), *extra), out=dbg, drop_flags=("-mllvm", "-amdgpu-mfma-vgpr-form"))
if "--build-only" in sys.argv:
    sys.exit(0)
B.LIB = dbg
B.needs_build = lambda: False
from optimal_quad_control_rl_amd import _lib
from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
dev = torch.device("cuda", 0)
_pa = [a for a in sys.argv[1:] if not a.startswith("--")]
L_ = int(_pa[0]) if _pa else 17
Bn = int(_pa[1]) if len(_pa) > 1 else 16384
R = max(65536 * 4, 16 * Bn)
obs = torch.randn((R, L_), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
perm = torch.randperm(R, device=dev).to(torch.int32)
pol = ActorCritic(L_, 4).to(dev)
up = MfmaPpoUpdater(pol, L_, dev, Bn)
lib = _lib.load()
lib.qr_ppo_debug_set_ticks.argtypes = [C.c_void_p, C.c_void_p]
grad4 = False   # (the round 1-2 kernel forms and their QR_PPO_* switches were removed in round 6: one role-split kernel)
waves = 2 * min(Bn // 64, 256) * 2 * (1 if grad4 else 2)    # two nets x groups x two 32-sample tiles (one chain wave each) [+ as many dW waves]
ticks = torch.zeros((waves, 16), dtype=torch.int64, device=dev)
for k in range(5):
    up.minibatch(obs, act, old_lp, adv, ret, perm[k * Bn:(k + 1) * Bn], 3e-4)
lib.qr_ppo_debug_set_ticks(up._h, C.c_void_p(ticks.data_ptr()))
reps = []
for k in range(12):
    up.minibatch(obs, act, old_lp, adv, ret, perm[k * Bn:(k + 1) * Bn], 3e-4)
    torch.cuda.synchronize()
    reps.append(ticks.cpu().numpy().copy())
t = np.stack(reps)[2:]
split = False
if split:
    names = ["entry", "image -> LDS + barrier", "index / obs gather, layer-1 operand, X0^T store", "fwd layer 1", "h1^T store", "fwd layer 2",
             "h2^T store", "fwd layer 3", "h3^T store", "output layer + loss gradient", "d4, d4^T store, d3", "d3^T store", "d2", "d2^T store",
             "d1", "d1^T store"]
    print("phase A of one wave (one 32-sample tile through forward, loss and backward; QR_PPO_SPLIT=1):")
elif not grad4:
    t8 = t.reshape(t.shape[0], -1, 8, 16)           # [rep][workgroup][wave][slot]
    ch, dw = t8[:, :, :4, :], t8[:, :, 4:, :]
    cn = ["entry", "gather issue, barrier S0 (image staged by the dW waves)", "fwd layer 1", "fwd layer 2", "fwd layer 3",
          "output layer + loss gradient", "d4^T, h3 -> LDS, barrier S1", "d3", "barrier S2, d3, h2 -> LDS, barrier S3",
          "d2 (transposed reads)", "barrier S4, d2, h1 -> LDS, barrier S5", "d1 (transposed reads)", "barrier S6, d1, x0^T -> LDS, barrier S7"]
    print("role-split gradient kernel, CHAIN wave (one 32-sample tile; %s):" % ("stamps WITHOUT draining the queues: where the waves are" if nodrain else "queues drained at every stamp"))
    for s_ in range(1, 13):
        print(f"  {s_:2d} {cn[s_]:62s} {np.median(ch[..., s_] - ch[..., s_ - 1]):8.0f} cycles")
    print(f"  chain total {np.median(ch[..., 12] - ch[..., 0]):8.0f} cycles")
    dn = {7: "dW4 (reads + MFMA) + barrier S2 + stores", 8: "wait S3", 9: "dW3 + barrier S4 + stores", 10: "wait S5",
          11: "dW2 + barrier S6 + stores", 12: "wait S7", 13: "dW1 + stores"}
    print("dW wave (slot 6 = released by S1):")
    print(f"     {'image staging (entry -> staged, before S0)':62s} {np.median(dw[..., 1] - dw[..., 0]):8.0f} cycles")
    print(f"     {'idle until S1':62s} {np.median(dw[..., 6] - dw[..., 1]):8.0f} cycles")
    for s_ in range(7, 14):
        print(f"  {s_:2d} {dn[s_]:62s} {np.median(dw[..., s_] - dw[..., s_ - 1]):8.0f} cycles")
    print(f"  workgroup: first entry -> last dW exit {np.median(dw[..., 13].max(-1) - t8[..., 0].min(-1)):8.0f} cycles")
    sys.exit(0)
else:
    names = ["entry", "gather issue, image -> LDS, barrier", "fwd layer 1", "fwd layer 2", "fwd layer 3", "output layer + loss gradient",
             "d4^T, h3^T -> LDS, barrier", "dW4 + stores, d3", "barrier, d3^T, h2^T -> LDS, barrier", "dW3 (2 x 2 tiles) + stores", "d2 (transposed reads)",
             "barrier, d2^T, h1^T -> LDS, barrier", "dW2 (2 x 2 tiles) + stores", "d1 (transposed reads)", "barrier, d1^T, x0^T -> LDS, barrier", "dW1 + stores"]
    print("fused gradient kernel, one wave (one 32-sample tile; queues drained at every stamp):")
for s in range(1, 16):
    print(f"  {s:2d} {names[s]:52s} {np.median(t[:, :, s] - t[:, :, s - 1]):8.0f} cycles")
print(f"  in-wave total {np.median(t[:, :, 15] - t[:, :, 0]):8.0f} cycles;  first entry -> last exit {np.median(t[:, :, 15].max(1) - t[:, :, 0].min(1)):8.0f} cycles (100 MHz clock64 ticks x ~24 = shader cycles)")
