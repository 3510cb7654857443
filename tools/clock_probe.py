#!/usr/bin/env python3
"""Where does a fused-rollout wave's time go, in CYCLES and in WALL time?  Builds a separate copy of the library with
-DQR_CLOCK_PROBE (un-drained stamps: kernel entry, loop start, loop end, after the final stores; each stamp = shader cycles via
s_memtime AND the constant 100 MHz counter) and prints, per env count: cycles per step per wave, the effective shader clock
under that load, the prologue / epilogue cost of a launch, and how the waves sat on the SIMDs (HW_ID).

    python tools/clock_probe.py [e2e|indi] [K] [n1,n2,...]        (GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "e2e"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sizes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4096, 65536, 131072, 262144, 1048576]
extra = os.environ.get("QR_PROBE_FLAGS", "").split()
tag = os.environ.get("QR_PROBE_TAG", "")
dbg = os.path.join(B.PKG, "_dbg", "libquadrace_clk%s.so" % tag)   # travels with the snapshot when built in the container
os.makedirs(os.path.dirname(dbg), exist_ok=True)
CSRC = os.environ.get("QR_PROBE_CSRC", B.CSRC)   # A/B against another checkout of the sources (same box, same call)
deps = [os.path.join(CSRC, h) for h in B.HEADERS]
flags = ["-DQR_CLOCK_PROBE", *extra]
if not (os.environ.get("QR_PROBE_NOBUILD") == "1" and os.path.exists(dbg)):   # the product's own pipeline (assembly rewrite + lint), stale objects only
    B.CSRC = CSRC
    B.build_native(extra_flags=tuple(flags), out=dbg)
if "--build" in sys.argv:
    sys.exit(0)
B.LIB = dbg
B.needs_build = lambda: False
import numpy as np  # noqa: E402
import torch  # noqa: E402
from optimal_quad_control_rl_amd import _lib  # noqa: E402
import bench  # noqa: E402

if os.environ.get("QR_PROBE_OLD_ABI") == "1":   # a build of the round-3 sources: it has no qr_rollout_kernel_name yet
    _lib.SIGNATURES.pop("qr_rollout_kernel_name", None)
L = _lib.load()
L.qr_debug_set_ticks.argtypes = [C.c_void_p, C.c_void_p]
print(f"# {variant} K={K} form={os.environ.get('QR_ROLLOUT_FORM', 'auto')} flags={extra}")
print("#     envs  waves | kernel_us us/step | loop cyc/step (med, p90) | eff MHz | prologue cyc(us) | tail cyc | waves/SIMD max | first->last wave entry us")
for n in sizes:
    env = bench.make_env(variant, n, 1, 0, residual=None if os.environ.get("QR_PROBE_NORES") == "1" else "default")
    n_waves = (n + 255) // 256 * 4
    ticks = torch.zeros((n_waves, 16), dtype=torch.int64, device="cuda")
    env.reset_device()
    acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
    out = None
    for _ in range(3):
        out = env.rollout_device(acts, out)
    L.qr_debug_set_ticks(env._h, C.c_void_p(ticks.data_ptr()))
    rows = []
    for _ in range(5):
        for _ in range(2):
            out = env.rollout_device(acts, out)
        torch.cuda.synchronize()
        ms = env.last_rollout_ms()
        t = ticks.cpu().numpy().astype(np.int64)
        c = t[:, 0:8:2]   # cycles: entry, loop start, loop end, after stores
        w = t[:, 1:8:2]   # 100 MHz
        loop_c = (c[:, 2] - c[:, 1]) / K
        loop_w = (w[:, 2] - w[:, 1]) / 100.0   # us
        mhz = np.median((c[:, 2] - c[:, 1]) / np.maximum(loop_w, 1e-9))
        pro_c, pro_us = np.median(c[:, 1] - c[:, 0]), np.median((w[:, 1] - w[:, 0]) / 100.0)
        tail_c = np.median(c[:, 3] - c[:, 2])
        spread = (w[:, 0].max() - w[:, 0].min()) / 100.0
        span = (w[:, 3].max() - w[:, 0].min()) / 100.0
        hw = t[:, 8]
        # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]... ; XCC in the upper word
        simd = (hw >> 4) & 3
        cu = (hw >> 8) & 15
        se = (hw >> 13) & 7
        xcc = (hw >> 32) & 15
        key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
        # co-residency: waves whose [entry, exit] intervals overlap on the same SIMD
        occ = 0
        for k in np.unique(key)[:64]:
            sel = np.where(key == k)[0]
            ev = sorted([(w[j, 0], 1) for j in sel] + [(w[j, 3], -1) for j in sel])
            cur = 0
            for _, d in ev:
                cur += d
                occ = max(occ, cur)
        rows.append((ms * 1e3, ms * 1e3 / K, np.median(loop_c), np.percentile(loop_c, 90), mhz, pro_c, pro_us, tail_c, occ, spread, span))
    r = np.median(np.array(rows), axis=0)
    print(f"{n:10d} {n_waves:6d} | {r[0]:9.1f} {r[1]:7.3f} | {r[2]:8.0f} {r[3]:8.0f} | {r[4]:7.0f} | {r[5]:7.0f} ({r[6]:5.2f}) | {r[7]:6.0f} | {int(r[8]):3d} | {r[9]:6.2f}  in-kernel span {r[10]:8.1f} us")
    env.close()
    del ticks, acts, out
