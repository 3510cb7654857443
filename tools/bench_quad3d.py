#!/usr/bin/env python3
"""Throughput of the predecessor environments (include/quad3d.h: Quadcopter3DVec hover f64, Quadcopter3DVecGates f32)
on one MI355X: per-step launches and the fused K-step kernel, against the HBM roofline, with the CPU oracle timed
beside them.  `python tools/bench_quad3d.py [--envs N] [--steps K]` prints one JSON line; bench.py embeds the same
measurement (with fewer steps) as `predecessor_envs`.

Algorithmic bytes per env-step (state read once / written once, action read once, outputs written once):
  hover (float64): read 16*8 + steps 4 + action 16 = 148; write 16*8 + steps 4 + states_out 128 + reward 8 + done 1 = 269 -> 417 B
  gates (float32): read 16*4 + target 4 + steps 4 + action 16 = 88; write 64 + 4 + 4 + states_out 64 + reward 4 + done 1 = 141 -> 229 B
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
BYTES = {"hover": 417, "gates": 229}   # per-step kernel: algorithmic = measured (profiles/r01_pmc_quad3d_summary.json)
# fused rollout (q3_step_many): the state stays in registers and only the final state is written, so a step moves its action
# (16 B in) and its reward + done flag out: hover 16 + 8 + 1, gates 16 + 4 + 1 -- that kernel is VALU / latency bound
# (f64 arithmetic for hover), and an HBM fraction built on the per-step bytes would exceed 1
FUSED_BYTES = {"hover": 25, "gates": 21}
# fused rollout WITH the rows a trainer consumes (q3_rollout, round 6): action 16 B in; env.states after the step (16 x 8 / 16 x 4),
# reward, done and truncation flag out -- the kernel whose HBM fraction is comparable to the race env's fused rollout
ROLLOUT_BYTES = {"hover": 16 + 128 + 8 + 1 + 1, "gates": 16 + 64 + 4 + 1 + 1}
VALU_F32_PEAK_TF, VALU_F64_PEAK_TF = 157.3, 78.6


def _flop_per_env_step(kind):
    """f64 (hover) / f32 (gates) vector flop per env-step of q3_rollout_kernel, counted by the PMC instruction counters
    (tools/pmc_compute.py -> profiles/r02_pmc_compute.json); None when that profile is absent"""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_compute.json")))["kernels"]
    except Exception:
        return None
    for name, v in ks.items():
        if "q3_rollout_kernel" in name and (("double" in name) == (kind == "hover")):
            return v["derived"].get("f64_vector_flop_per_env_step" if kind == "hover" else "f32_vector_flop_per_env_step")
    return None
Q3_TRACK = (np.array([[-1.5, -2, -1.5], [1.5, 2, -1.5], [1.5, -2, -1.5], [-1.5, 2, -1.5]] * 2, dtype=np.float64),
            np.array([0, 0, np.pi, np.pi] * 2), np.array([-4, -2, -1.5]))     # Q3 cell 16


def _time_region(fn, repeats):
    ts = []
    for _ in range(repeats):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return float(np.median(ts))


def measure(kind, n=65536, K=200, repeats=3, cpu_seconds=3.0):
    from optimal_quad_control_rl_amd.quad3d import Quadcopter3DVec, Quadcopter3DVecGates

    env = Quadcopter3DVec(n) if kind == "hover" else Quadcopter3DVecGates(n, *Q3_TRACK)
    dev = env.device
    gen = torch.Generator(device=dev).manual_seed(0)
    acts = torch.rand((K, n, 4), device=dev, generator=gen) * 2 - 1
    env.reset_device()
    for k in range(min(K, 20)):
        env.step_device(acts[k])

    def per_step():
        for k in range(K):
            env.step_device(acts[k])

    t_step = _time_region(per_step, repeats)
    env.rollout_device(acts)
    t_fused = _time_region(lambda: env.rollout_device(acts), repeats)
    rew, done, _ = env.rollout_device(acts)
    out = {"env": "Quadcopter3DVec (hover, f64)" if kind == "hover" else "Quadcopter3DVecGates (f32, 8-gate track of Q3 cell 16)",
           "envs": n, "steps": K, "actions": "U(-1,1) pre-generated on device", "bytes_per_env_step": BYTES[kind],
           "done_fraction": float(done.float().mean())}
    gbs = BYTES[kind] * n * K / t_step / 1e9
    out["per_step_launch"] = {"us_per_step": t_step / K * 1e6, "env_steps_per_s": n * K / t_step,
                              "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": gbs / HBM_PEAK_GBS, "bytes_per_env_step": BYTES[kind]}}
    peak = VALU_F64_PEAK_TF if kind == "hover" else VALU_F32_PEAK_TF
    flop = _flop_per_env_step(kind)
    tf = None if flop is None else flop * n * K / t_fused / 1e12
    fgbs = FUSED_BYTES[kind] * n * K / t_fused / 1e9
    out["fused_rollout"] = {"us_per_step": t_fused / K * 1e6, "env_steps_per_s": n * K / t_fused,
                            "roofline": {"bound": "valu", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": None if tf is None else tf / peak,
                                         "flop_per_env_step": flop,
                                         "hbm": {"achieved": fgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fgbs / HBM_PEAK_GBS,
                                                 "bytes_per_env_step": FUSED_BYTES[kind]}}}
    env.rollout_states_device(acts[: min(K, 64)])
    Kr = min(K, 256 if kind == "hover" else 512)        # [K][N][16] rows: 128 / 64 B per env-step
    bufs = env.rollout_states_device(acts[:Kr])
    t_rows = _time_region(lambda: env.rollout_states_device(acts[:Kr], bufs), repeats)
    rgbs = ROLLOUT_BYTES[kind] * n * Kr / t_rows / 1e9
    out["fused_rollout_with_rows"] = {"what": "q3_rollout: K steps in one kernel writing env.states, reward, done, trunc of EVERY step (what a trainer consumes)",
                                      "steps": Kr, "us_per_step": t_rows / Kr * 1e6, "env_steps_per_s": n * Kr / t_rows,
                                      "roofline": {"bound": "hbm", "achieved": rgbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rgbs / HBM_PEAK_GBS,
                                                   "bytes_per_env_step": ROLLOUT_BYTES[kind]}}
    del bufs
    if cpu_seconds > 0:
        from oracle import quad3d as q3

        m = 4096
        o = q3.Quad3DOracle(q3.HOVER, m) if kind == "hover" else q3.Quad3DOracle(q3.GATES, m, *Q3_TRACK)
        o.reset()
        a = np.random.default_rng(0).uniform(-1, 1, (m, 4)).astype(np.float32)
        cores = min(16, os.cpu_count() or 1)
        res = {}
        for threads in (1, cores):
            o.set_threads(threads)
            o.step(a)
            t0, it = time.perf_counter(), 0
            while time.perf_counter() - t0 < cpu_seconds / 2:
                o.step(a)
                it += 1
            res[threads] = m * it / (time.perf_counter() - t0)
        out["cpu_baseline"] = {"kind": "port", "what": "oracle/quad3d_oracle.c (OpenMP over envs)", "sample": f"{m} envs",
                               "value": res[cores], "cores": cores, "single_thread": res[1], "unit": "env-steps/s"}
    env.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=1000)
    args = ap.parse_args()
    print(json.dumps({k: measure(k, args.envs, args.steps, repeats=5, cpu_seconds=6.0) for k in ("hover", "gates")}))
