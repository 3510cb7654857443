#!/usr/bin/env python3
"""Full-chip-load agreement of a fused rollout kernel with K x the per-step kernel, elementwise on the device (the check that found the
lean form's lost reward stores at two workgroups per CU; tests/test_gpu_round4.py runs the cross-process version of it):
    [QR_PROBE_LIB=<build>] [QR_ROLLOUT_FORM=multi_wave|general] python tools/lean_stress.py [envs] [e2e|indi] [gates_ahead]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimal_quad_control_rl_amd import build as B
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("QR_PROBE_LIB"):
    B.LIB = os.path.join(ROOT, os.environ["QR_PROBE_LIB"]); B.needs_build = lambda: False
from optimal_quad_control_rl_amd import Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, square_track
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
variant = sys.argv[2] if len(sys.argv) > 2 else "e2e"
ga = int(sys.argv[3]) if len(sys.argv) > 3 else 1
K = 40
from optimal_quad_control_rl_amd import Quadcopter3DGatesINDI
def mk():
    if variant == "indi":
        e = Quadcopter3DGatesINDI(n, *square_track(), gates_ahead=ga, seed=5, infos_mode='none')
    else:
        kw = dict(residual=None) if variant == "e2e_nores" else {}
        e = Quadcopter3DGates(n, *square_track(), gates_ahead=ga, seed=5, infos_mode='none', **kw); e.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
    if os.environ.get("QR_ROLLOUT_FORM"): e.set_rollout_form(os.environ["QR_ROLLOUT_FORM"])   # auto | multi_wave | general
    e.reset_device(); return e
a = torch.rand((K, n, 4), device='cuda', generator=torch.Generator(device='cuda').manual_seed(2)) * 2 - 1
A = mk(); print(A.rollout_kernel_name())
o, r, d, t = A.rollout_device(a)
B = mk()
O = torch.empty_like(o); R = torch.empty_like(r); D = torch.empty_like(d)
for k in range(K):
    ob, rw, dn, tr = B.step_device(a[k])
    O[k] = ob; R[k] = rw; D[k] = dn
for name, x, y in (("obs", o, O), ("rew", r, R), ("done", d, D)):
    neq = (x.view(torch.int32) != y.view(torch.int32)) if x.dtype == torch.float32 else (x != y)
    cnt = int(neq.sum())
    print(name, "mismatches", cnt, "of", neq.numel())
    if cnt:
        idx = neq.nonzero()[:10].tolist()
        print("  first:", idx)
        for ii in idx[:5]:
            print("   ", ii, x[tuple(ii)].item(), y[tuple(ii)].item())
        steps = neq.reshape(K, -1).any(1).nonzero().flatten().tolist(); print("  steps with mismatches:", steps[:40])
        envs = neq.reshape(K, n, -1).any(2).nonzero()[:, 1]   # which lane quarter of its wave does a mismatching env sit in?
        print("  mismatching (step, env) pairs by lane quarter:", torch.bincount((envs % 64) // 16, minlength=4).tolist())
        if x.dim() == 3:
            print("  by observation column:", neq.reshape(-1, x.shape[-1]).sum(0).tolist())
            anyk = neq.any(2)                                           # [K, n]
            bad_env = anyk.any(0).nonzero().flatten()
            first = anyk[:, bad_env].int().argmax(0)                   # first mismatching step of each bad env
            cols = neq[first, bad_env]                                  # [bad envs, L] mismatching columns at that step
            print("  envs that ever mismatch:", bad_env.numel(), " columns mismatching at an env's FIRST bad step:", cols.sum(0).tolist())
            pat, cnt = torch.unique(cols, dim=0, return_counts=True)
            for pp, cc in sorted(zip(pat.tolist(), cnt.tolist()), key=lambda t: -t[1])[:6]:
                print("    pattern", [i for i, v in enumerate(pp) if v], "x", cc)
            for j in range(min(3, bad_env.numel())):
                e, k0 = int(bad_env[j]), int(first[j]); c = cols[j].nonzero().flatten().tolist()
                print("    env", e, "lane", e % 64, "step", k0, "cols", c, "got", [x[k0, e, ci].item() for ci in c], "want", [y[k0, e, ci].item() for ci in c])
