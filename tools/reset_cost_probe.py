#!/usr/bin/env python3
"""What the auto-reset costs the fused rollout kernel (GPU box): us per step at 65 536 envs x 1000 steps for termination rates forced
through max_steps, and with resets switched off (pause_if_collision).  QR_ROLLOUT_FORM=multi_wave selects the kernel without the per-lane reset
stash.  Usage: python tools/reset_cost_probe.py [envs] [steps]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from optimal_quad_control_rl_amd import Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, zigzag_track
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
env = Quadcopter3DGates(n, *zigzag_track(), gates_ahead=1, seed=0, infos_mode="none")
import os
if os.environ.get("QR_ROLLOUT_FORM"): env.set_rollout_form(os.environ["QR_ROLLOUT_FORM"])
env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
g = torch.Generator(device="cuda").manual_seed(0)
for name, scale, bias, ms, pic in (("uniform(-1,1)", 1.0, 0.0, 1200, False), ("max_steps 20", 1.0, 0.0, 20, False), ("max_steps 5", 1.0, 0.0, 5, False), ("max_steps 2", 1.0, 0.0, 2, False), ("pause_if_collision (no resets)", 1.0, 0.0, 1200, True)):
    env.max_steps = ms; env.pause_if_collision = pic
    acts = ((torch.rand((K, n, 4), device="cuda", generator=g) * 2 - 1) * scale + bias).contiguous()
    env.reset_device()
    out = env.rollout_device(acts)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        env.reset_device(); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = env.rollout_device(acts); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    print(f"{name:26s} us/step {min(ts):.3f}  done fraction {float(out[2].float().mean()):.4f}", flush=True)
