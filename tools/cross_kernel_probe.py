#!/usr/bin/env python3
"""Is the packed-f32 hazard of DESIGN section 4.1 a matter of ONE kernel's waves, or of any two waves that share a SIMD?  Two 65 536-env
handles (one workgroup per CU each: alone, a launch has one wave per SIMD and never fails) are driven from two streams at once, in the
multi-wave kernel form (255 registers per lane: two such waves fit a SIMD), and compared with their serial runs.
    QR_PROBE_LIB=optimal_quad_control_rl_amd/_dbg/libisa_ctl.so QR_ROLLOUT_STASH=0 python tools/cross_kernel_probe.py     (the failing round-4 build)
    QR_ROLLOUT_FORM=multi_wave python tools/cross_kernel_probe.py                                                         (this build)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
if os.environ.get("QR_PROBE_LIB"):
    B.LIB = os.path.join(ROOT, os.environ["QR_PROBE_LIB"]); B.needs_build = lambda: False
from optimal_quad_control_rl_amd import Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, square_track
n, K = 65536, 400
def mk(seed):
    e = Quadcopter3DGates(n, *square_track(), gates_ahead=1, seed=seed, infos_mode="none"); e.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
    if os.environ.get("QR_ROLLOUT_FORM"): e.set_rollout_form(os.environ["QR_ROLLOUT_FORM"])
    e.reset_device(); return e
acts = torch.rand((K, n, 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 2 - 1
ref = []
for seed in (11, 12):
    e = mk(seed); print(e.rollout_kernel_name()); ref.append([t.clone() for t in e.rollout_device(acts)]); e.close()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for rep in range(6):
    ea, eb = mk(11), mk(12)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1): ra = ea.rollout_device(acts)
    with torch.cuda.stream(s2): rb = eb.rollout_device(acts)
    torch.cuda.synchronize()
    bad = [int((x.view(torch.uint8) != y.view(torch.uint8)).any(-1).sum()) if x.dim() > 2 else int((x != y).sum()) for got, want in ((ra, ref[0]), (rb, ref[1])) for x, y in zip(got, want)]
    envs = (ra[0] != ref[0][0]).any(2).any(0).nonzero().flatten()
    print("rep", rep, "mismatching (obs rows, rewards, dones, truncs) handle A:", bad[:4], " handle B:", bad[4:], " lane quarters of A's bad envs:", torch.bincount((envs % 64) // 16, minlength=4).tolist())
    ea.close(); eb.close()
