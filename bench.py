#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the MI355X-native quadrotor race environment (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--variant e2e|indi] [--envs 65536]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one env.step() of the whole batch (residual MLPs -> Euler integration -> reward / gate logic ->
auto-reset -> gate-frame observation for every env).  Workload at N=1 = BASELINE.json configs[1] as specified in
SURVEY.md 8(d): 65 536 envs, Bebop E2E + NNDroneModel residual MLPs + training disturbance ranges, gates_ahead = 1,
7-gate zigzag track, random U(-1,1) actions PRE-GENERATED on the device [K][N][4] (torch Philox, seed = rank);
every step's obs / reward / done go to a full rollout buffer [K][N][...] (what PPO's collect phase stores), so each
step writes fresh HBM.  Inputs are resident in HBM before the timed region.

Two ways to run those K steps through the C ABI are timed, both producing bit-identical outputs:
  * `value`: qr_step_many -- ONE fused rollout kernel for the K steps (env state stays in registers between steps;
    the MI355X-native way to replay a recorded action sequence: no per-step launch, state round trip or
    end-of-kernel write-back);
  * `per_step_launch`: qr_step_launches -- K step kernels, one per env.step(), the calling pattern of a closed
    loop whose policy runs between steps (what the reference's SB3 loop does).
With --gpus N each rank simulates its own 65 536-env shard (weak scaling, no data-path collective); the
rollout-boundary RCCL all-gather of [obs|reward|done] is measured separately ("exchange").

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel of the timed region: SURVEY 8(d)
algorithmic bytes per launch / hipEvent launch duration on the launch stream; `traffic` = HBM bytes per launch from
the rocprofv3 FETCH_SIZE/WRITE_SIZE passes in profiles/pmc_summary.json), "cpu_baseline" (the CPU oracle -- a C port
of the reference, parity-pinned -- timed on this box's host cores on a bounded sample), "parity" (north_star
single-trajectory max |d state| vs the reference's recorded step()).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BYTES_PER_ENV_STEP = {"e2e": lambda ga: 189 + 4 * (20 + 4 * ga), "indi": lambda ga: 141 + 4 * (13 + 4 * ga)}
# e2e: read world 64 + dist 24 + action 16 + target 4 + steps 4 = 112; write world 64 + obs 4*(20+4G) + reward 4
#      + done 1 + target 4 + steps 4  -> 285 B at G=1.  indi: read 52+16+4+4 = 76; write 52 + 4*(13+4G) + 13 -> 209 B.


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--variant", default="e2e", choices=["e2e", "indi"])
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--gates-ahead", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=5, help="timed repetitions of the K-step region (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true")
    ap.add_argument("--no-residual", action="store_true", help="experiment: E2E without the residual MLPs")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_env(variant, n, ga, env_id_base, seed=0, residual="default"):
    from optimal_quad_control_rl_amd import (Quadcopter3DGates, Quadcopter3DGatesINDI, TRAIN_DISTURBANCE_RANGES,
                                             square_track, zigzag_track)

    if variant == "e2e":
        env = Quadcopter3DGates(n, *zigzag_track(), gates_ahead=ga, seed=seed, env_id_base=env_id_base,
                                residual=residual, infos_mode="none")
        env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES  # R:772-781
    else:
        env = Quadcopter3DGatesINDI(n, *square_track(), gates_ahead=ga, seed=seed, env_id_base=env_id_base,
                                    infos_mode="none")
    return env


def parity_probe():
    """north_star correctness: 1 env, E2E, no residual, recorded action sequence (tests/golden F5, generated
    from the real reference): free-running 100 steps, max |d state| / max(1,|state|)."""
    import torch
    from optimal_quad_control_rl_amd import Quadcopter3DGates, zigzag_track

    d = np.load(os.path.join(ROOT, "tests", "golden", "f5_traj_e2e_noresidual.npz"))
    p = "ga1_ctrl_"
    env = Quadcopter3DGates(1, *zigzag_track(), gates_ahead=1, residual=None, infos_mode="none")
    env.set_state_tensors(world=d[p + "world0"], dist=d[p + "dist0"], target=d[p + "target0"], steps=d[p + "steps0"])
    worst = 0.0
    for k in range(100):
        env.step_device(torch.as_tensor(d[p + "actions"][k]).cuda())
        w = env.get_state_tensors()[0].cpu().numpy().astype(np.float64)
        ref = d[p + "world"][k].astype(np.float64)
        worst = max(worst, float((np.abs(w - ref) / np.maximum(1.0, np.abs(ref))).max()))
    return {"max_rel_dstate_100_steps": worst, "tolerance": 1e-5, "reference": "tests/golden/f5 (ref step(), 1 env, no residual)"}


def host_path_probe(variant, n, ga):
    """PCIe-inclusive rate of the SB3-facing NumPy path (step_async + step_wait): actions H->D, obs/reward/done D->H
    every step.  Never the headline `value` (which keeps everything HBM-resident)."""
    env = make_env(variant, n, ga, env_id_base=0, seed=1)
    env.infos_mode = "reference"
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
    for _ in range(3):
        env.step(acts)
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        env.step(acts)
    dt = (time.perf_counter() - t0) / steps
    env.close()
    return {"what": "env.step(numpy actions) -> numpy obs/reward/done/infos (SB3 calling convention), PCIe + NumPy inclusive",
            "ms_per_step": dt * 1e3, "value": n / dt, "unit": "env-steps/s"}


def cpu_baseline(variant, n, ga, seconds):
    """The oracle (C port of the reference, parity-pinned against reference fixtures) on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from optimal_quad_control_rl_amd import TRAIN_DISTURBANCE_RANGES, square_track, zigzag_track
    from optimal_quad_control_rl_amd.vec_env import default_residual_blob

    cores = len(os.sched_getaffinity(0))
    trk = zigzag_track() if variant == "e2e" else square_track()
    env = O.OracleEnv(O.E2E if variant == "e2e" else O.INDI, n, *trk, gates_ahead=ga)
    if variant == "e2e":
        env.set_residual(default_residual_blob())
        env.set_disturbance(TRAIN_DISTURBANCE_RANGES, 1.0)
    env.seed(0)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(8, n, 4)).astype(np.float32)
    out = {}
    cands = sorted({1, 8, 16, 32, 64, cores} & set(range(1, cores + 1)))
    per = max(0.8, seconds / (len(cands) + 1))
    for threads in cands:
        env.set_threads(threads)
        env.reset()
        env.step(acts[0])
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < (2 * per if threads == 1 else per) and steps < 4000:
            env.step(acts[steps % 8])
            steps += 1
        dt = time.perf_counter() - t0
        out[threads] = (n * steps / dt, steps)
    best_threads = max(out, key=lambda k: out[k][0])
    return {"value": out[best_threads][0], "unit": "env-steps/s", "cores": best_threads, "kind": "port",
            "sample": f"{out[best_threads][1]} steps x {n} envs of the same workload ({variant}, random actions) on "
                      f"oracle/quadrace_oracle.c (C port of the reference, OpenMP over envs); thread sweep "
                      + ", ".join(f"{t}t: {v[0]/1e6:.1f}M/s" for t, v in sorted(out.items())),
            "value_1core": out[1][0], "host_cores": cores}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n, K, W, ga = args.envs, args.steps, args.warmup, args.gates_ahead
    env = make_env(args.variant, n, ga, env_id_base=rank * n, seed=0, residual=None if args.no_residual else "default")
    L = env.state_len
    dev = env.device
    gen = torch.Generator(device=dev).manual_seed(rank)  # torch Philox, seed = rank
    KB = max(K, W, 1)
    actions = torch.rand((KB, n, 4), device=dev, generator=gen) * 2 - 1
    out = (torch.empty((KB, n, L), dtype=torch.float32, device=dev), torch.empty((KB, n), dtype=torch.float32, device=dev),
           torch.empty((KB, n), dtype=torch.uint8, device=dev), torch.empty((KB, n), dtype=torch.uint8, device=dev))

    def view(k):
        return tuple(t[:k] for t in out)

    env.reset_device()
    if W > 0:
        env.rollout_device(actions[:W], view(W))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn):
        ts = []
        for _ in range(max(1, args.repeats)):
            barrier()
            t0 = time.perf_counter()
            fn()  # EXACTLY K steps
            barrier()
            ts.append(time.perf_counter() - t0)
        el = float(np.median(ts))
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ts

    # (1) fused rollout: qr_step_many = ONE kernel for the K steps, env state register-resident between steps
    fused_elapsed, fused_times = timed(lambda: env.rollout_device(actions[:K], view(K)))
    fused_kernel_ms = env.last_rollout_ms()  # hipEvents around the single launch, on the launch stream
    # (2) per-step launches: K x qr_step, one kernel per env.step() (closed-loop calling pattern)
    elapsed, times = timed(lambda: env.step_sequence_device(actions[:K], view(K)))
    step_region_ms = env.last_rollout_ms()  # hipEvents bracketing the K back-to-back step kernels on the launch stream
    dones_frac = float(out[2][:K].float().mean().item())

    # --- roofline ------------------------------------------------------------------------------------------------
    # Algorithmic bytes per env-step: 285 B (E2E, G=1) / 209 B (INDI): state read+written
    # once, action read once, outputs written once.
    bytes_per_step = BYTES_PER_ENV_STEP[args.variant](ga) * n
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get(f"{args.variant}_n{n}_ga{ga}", {})
        except Exception:
            pmc = {}
    # (a) fused rollout kernel: ONE launch = K steps; duration from hipEvents around that launch on its stream
    fused_launch_s = fused_kernel_ms * 1e-3
    fused_ach = bytes_per_step * K / fused_launch_s / 1e9
    fused_traffic = pmc.get("fused_hbm_bytes_per_step")
    roofline = {"bound": "hbm", "achieved": fused_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fused_ach / HBM_PEAK_GBS,
                "traffic": None if fused_traffic is None else fused_traffic * K,
                "kernel": f"qr::rollout_kernel<{args.variant},ga={ga}> (one launch = {K} steps)",
                "launch_us": fused_kernel_ms * 1e3, "us_per_step": fused_kernel_ms * 1e3 / K,
                "bytes_per_launch": bytes_per_step * K,
                "note": "algorithmic bytes per SURVEY 8(d); the fused kernel keeps the env state in registers, so its real "
                        "HBM traffic (PMC) is the action + output bytes only: frac_of_measured_traffic is its true HBM "
                        "utilisation -- the kernel is VALU/latency bound, not HBM bound",
                "frac_of_measured_traffic": None if fused_traffic is None else fused_traffic / (fused_launch_s / K) / 1e9 / HBM_PEAK_GBS}
    # (b) per-step kernel: the K step kernels run back-to-back on one stream (rocprofv3: median gap 0 ns), so the
    #     hipEvent time over the timed region / K is the average launch duration (per-launch event pairs are also
    #     reported, but the markers themselves stretch an ~8 us kernel by 2-3 us)
    mean_kernel_ms = step_region_ms / K
    Kp = min(K, 200)
    pair_kernel_ms, _ = env.profile_rollout(actions[:Kp], view(Kp))
    step_ach = bytes_per_step / (mean_kernel_ms * 1e-3) / 1e9
    step_roofline = {"bound": "hbm", "achieved": step_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_ach / HBM_PEAK_GBS,
                     "traffic": pmc.get("hbm_bytes_per_launch"), "kernel": f"qr::step_kernel<{args.variant},ga={ga}>",
                     "kernel_us": mean_kernel_ms * 1e3, "bytes_per_launch": bytes_per_step, "launches_timed": K,
                     "kernel_us_event_pair_per_launch": pair_kernel_ms * 1e3}

    # --- closed loop (config 5 collect phase): policy MLP + sampling + env step in one kernel --------------------------
    closed_loop = None
    try:
        from optimal_quad_control_rl_amd.policy import MfmaPolicy
        from optimal_quad_control_rl_amd.ppo import ActorCritic

        torch.manual_seed(0)
        net = ActorCritic(L, 4).to(dev)  # random-init weights of the reference's policy architecture (R:783)
        pol = MfmaPolicy(L, dev.index).load_torch(net.pi)
        Kc = min(K, 256)
        cl_out = None
        cl_t = []
        for r in range(3):
            barrier()
            t0 = time.perf_counter()
            cl_res = env.rollout_policy_device(pol, Kc, torch.zeros(4), noise_seed=rank, first_step=r * Kc, out=cl_out)
            barrier()
            cl_t.append(time.perf_counter() - t0)
            cl_out = cl_res[:6]
        cl_el = float(np.median(cl_t[1:]))
        closed_loop = {"what": "qr_rollout_policy: K x [obs -> policy MLP (L->120->120->120->4, f16 MFMA) -> Gaussian sample -> "
                               "env.step] in ONE kernel (PPO collect phase); random-init policy weights",
                       "steps": Kc, "ms_per_step": cl_el * 1e3 / Kc, "value": n * world * Kc / cl_el, "unit": "env-steps/s",
                       "kernel_us_per_step": env.last_rollout_ms() * 1e3 / Kc}
        del cl_out, cl_res
    except Exception as ex:  # pragma: no cover
        closed_loop = {"error": repr(ex)}

    # --- rollout-boundary exchange (config 4): RCCL all-gather of [obs | reward | done] --------------------------
    exchange = None
    if world > 1 and not args.no_exchange:
        from optimal_quad_control_rl_amd.sharded import pack_rollout

        Kx = min(K, 64)
        packed = pack_rollout(out[0][:Kx], out[1][:Kx], out[2][:Kx])
        gathered = torch.empty((world * Kx,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, packed)
        barrier()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(gathered, packed)
        barrier()
        dt = time.perf_counter() - t0
        exchange = {"op": "all_gather_into_tensor(RCCL)", "steps": Kx, "bytes_per_rank": packed.numel() * 4,
                    "ms": dt * 1e3, "GBps_in_per_gpu": packed.numel() * 4 * (world - 1) / dt / 1e9}

    if rank == 0:
        total_steps = n * world * K
        result = {
            "metric": "env-steps/sec at N=65536 envs per GPU (Quadcopter3DGates.step, "
                      + ("E2E + residual MLPs" if args.variant == "e2e" else "INDI inner loop") + ")",
            "value": total_steps / fused_elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": fused_elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} envs/GPU, " + ("Bebop E2E (motor-cmd actions) + NNDroneModel residual MLPs + "
                                                         "training disturbance ranges, 7-gate zigzag"
                                                         if args.variant == "e2e" else "INDI inner-loop variant, 4-gate square (x2)")
                       + f", gates_ahead={ga}, U(-1,1) actions pre-generated on device [K][N][4], outputs to a [K][N] rollout buffer",
                       "envs_per_gpu": n, "variant": args.variant, "gates_ahead": ga, "obs_len": L,
                       "sharding": f"{world} independent shard(s), env_id_base = rank*N"},
            "path": "qr_step_many (fused K-step rollout kernel)",
            "repeats": len(fused_times), "all_ms_per_step": [t * 1e3 / K for t in fused_times], "done_fraction": dones_frac,
            "roofline": roofline,
            "per_step_launch": {
                "what": "qr_step_launches: the same K steps as K step-kernel launches (one per env.step(); bit-identical "
                        "outputs) -- the closed-loop calling pattern",
                "value": total_steps / elapsed, "unit": "env-steps/s", "ms_per_step": elapsed * 1e3 / K,
                "all_ms_per_step": [t * 1e3 / K for t in times], "roofline": step_roofline},
        }
        if closed_loop:
            result["closed_loop"] = closed_loop
        if exchange:
            result["exchange"] = exchange
        if world == 1:
            if not args.no_parity:
                try:
                    result["parity"] = parity_probe()
                except Exception as ex:  # pragma: no cover
                    result["parity"] = {"error": repr(ex)}
            if not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                try:  # the PPO minibatch update on the matrix cores (qr_ppo_minibatch) vs torch; measured
                    # before the CPU legs (their worker threads would compete with the launch thread: six launches per update)
                    from bench_ppo_update import measure as ppo_measure

                    result["ppo_update"] = ppo_measure(L, 16384, 65536 * 8, 100)
                except Exception as ex:  # pragma: no cover
                    result["ppo_update"] = {"error": repr(ex)}
                try:
                    result["host_numpy_path"] = host_path_probe(args.variant, n, ga)
                except Exception as ex:  # pragma: no cover
                    result["host_numpy_path"] = {"error": repr(ex)}
                result["cpu_baseline"] = cpu_baseline(args.variant, n, ga, args.cpu_seconds)
                try:  # the predecessor envs of "3D quad.ipynb" (include/quad3d.h), short measurement
                    from bench_quad3d import measure as q3_measure

                    result["predecessor_envs"] = {k: q3_measure(k, n, 200, repeats=3, cpu_seconds=2.0) for k in ("hover", "gates")}
                except Exception as ex:  # pragma: no cover
                    result["predecessor_envs"] = {"error": repr(ex)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
