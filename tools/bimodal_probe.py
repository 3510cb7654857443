#!/usr/bin/env python3
"""Round 6: what separates the two modes of the large-N INDI rollout (50-53 vs 60-66 G env-steps/s at 1 Mi envs on one build)?
One invocation = ONE fresh process that measures, back to back on the same allocations pattern:
  * pure write / copy bandwidth of the device (torch fill_ and copy_ of 1 GiB): the memory system without our kernels,
  * the INDI and the E2E fused rollout at 1 Mi envs (and INDI at 65 536 envs as the one-wave-per-SIMD control),
  * clocks / power / temperature right after the INDI runs (rocm-smi), the addresses of the caller's buffers,
  * optionally (argv[1] == "slab") the same with every caller buffer carved out of ONE 1 GiB-aligned slab.
tools/bimodal_probe.sh runs it N times per box; a mode that flips between processes of one box is placement, a mode that
is constant per box and differs between boxes is the box (clocks / memory system)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
tag = sys.argv[2] if len(sys.argv) > 2 else "0"


def timed(fn, reps=6, inner=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / inner)
    return float(np.median(ts[1:])), float(min(ts[1:])), float(max(ts[1:]))


res = {"mode": mode, "tag": tag, "pid": os.getpid()}
G = 1 << 30
a = torch.empty(G, dtype=torch.uint8, device="cuda")
b = torch.empty(G, dtype=torch.uint8, device="cuda")
t, lo, hi = timed(lambda: a.fill_(1))
res["fill_TBps"] = G / t / 1e12
t, lo, hi = timed(lambda: b.copy_(a))
res["copy_TBps_rw"] = 2 * G / t / 1e12
del a, b


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        c = j[sorted(j)[0]]
        keep = {}
        for k, v in c.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "power" in kl or "temperature" in kl or "socclk" in kl:
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def rollout(variant, n, K):
    env = bench.make_env(variant, n, 1, 0)
    env.reset_device()
    if mode == "slab":   # every caller buffer inside one 1 GiB-aligned slab
        L = env.state_len
        need = K * n * (16 + 4 * L + 4 + 1 + 1) + (1 << 21)
        raw = torch.empty(need + G, dtype=torch.uint8, device="cuda")
        base = (-raw.data_ptr()) % G
        off = [base]

        def carve(nbytes, dtype, shape):
            o = off[0]
            off[0] = (o + nbytes + 4095) // 4096 * 4096
            return raw[o:o + nbytes].view(dtype).view(*shape)
        acts = carve(K * n * 16, torch.float32, (K, n, 4))
        acts.uniform_(-1, 1)
        out = (carve(K * n * L * 4, torch.float32, (K, n, L)), carve(K * n * 4, torch.float32, (K, n)),
               carve(K * n, torch.uint8, (K, n)), carve(K * n, torch.uint8, (K, n)))
        env.rollout_device(acts, out)
    else:
        acts = torch.rand((K, n, 4), device="cuda") * 2 - 1
        out = env.rollout_device(acts)
    t, lo, hi = timed(lambda: env.rollout_device(acts, out), reps=7, inner=2)
    ptrs = [acts.data_ptr()] + [o.data_ptr() for o in out]
    r = {"G_env_steps_s": n * K / t / 1e9, "lo": n * K / hi / 1e9, "hi": n * K / lo / 1e9, "kernel": env.rollout_kernel_name(),
         "ptr_GiB": [round(p / G, 3) for p in ptrs]}
    del env, acts, out
    return r


res["indi_1Mi"] = rollout("indi", 1 << 20, 50)
res["smi_after_indi"] = smi()
res["e2e_1Mi"] = rollout("e2e", 1 << 20, 50)
res["indi_64Ki"] = rollout("indi", 1 << 16, 200)
res["indi_1Mi_again"] = rollout("indi", 1 << 20, 50)
print(json.dumps(res))
