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
  * `value`: qr_step_many -- ONE fused rollout kernel for the K steps (env state stays in registers between steps);
  * `per_step_launch`: qr_step_launches -- K step kernels, one per env.step() (closed-loop calling pattern).
Timing: the K-step region is repeated back-to-back R times inside ONE barrier + synchronize bracket, R chosen (the
same on every rank) so that the bracket holds >= 20 ms of work; `ms_per_step` = bracket / (R K); the whole bracket is
measured `--repeats` times (median reported, all values listed).  `steps` stays K.

Roofline objects (every `frac` is against the roof that binds that kernel, and can be recomputed from `profiles/`):
  * fused rollout kernel (the dominant kernel of `value`): bound "hbm" on the bytes THAT kernel moves (action in, obs / reward /
    done out: 118 B E2E, 90 B INDI per env-step; `traffic` = the PMC FETCH_SIZE / WRITE_SIZE measurement, profiles/pmc_summary.json),
    with the f32 vector + matrix-core flop rate (PMC instruction counts, profiles/r02_pmc_compute.json) vs the 157.3 TFLOP/s
    f32 vector peak beside it as `valu` -- the larger fraction names the binding roof;
  * per-step kernel: bound "hbm" -- SURVEY 8(d) algorithmic bytes (= the measured traffic, ratio 1.01) / average launch
    duration vs 8 TB/s, with the launch floor of an empty kernel of the same shape (profiles/r02_launch_floor.json);
  * closed-loop kernel: bound "mfma" -- policy-MLP f16 flop per env-step / duration vs the 2.5 PFLOP/s dense f16 peak.
Extra objects: "indi" (BASELINE config 3, same measurements), "config4" (N > 1: 32 768 envs per GPU + RCCL all-gather of
the rollout), "cpu_baseline" (the CPU oracle -- a C port of the reference, parity-pinned -- on this box's host cores, bounded
sample), "parity" (north_star single-trajectory max |d state| vs the reference's recorded step()).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_F32_PEAK_TF = 157.3  # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
MFMA_F16_PEAK_TF = 2500.0  # dense f16 / bf16 matrix peak
BYTES_PER_ENV_STEP = {"e2e": lambda ga: 189 + 4 * (20 + 4 * ga), "indi": lambda ga: 141 + 4 * (13 + 4 * ga)}
# e2e: read world 64 + dist 24 + action 16 + target 4 + steps 4 = 112; write world 64 + obs 4*(20+4G) + reward 4
#      + done 1 + target 4 + steps 4  -> 285 B at G=1.  indi: read 52+16+4+4 = 76; write 52 + 4*(13+4G) + 13 -> 209 B.
# fused rollout kernel (qr_step_many): the env state never leaves registers, so per env-step it moves only the action in and
# obs / reward / done / trunc out: E2E 16 + 4*(20+4G) + 6 = 118 B, INDI 16 + 4*(13+4G) + 6 = 90 B at G=1 (PMC: 121.5 / 92.3 B)
FUSED_BYTES_PER_ENV_STEP = {"e2e": lambda ga: 16 + 4 * (20 + 4 * ga) + 6, "indi": lambda ga: 16 + 4 * (13 + 4 * ga) + 6}
MIN_TIMED_MS = 20.0
# Numbers in the line that were NOT measured by this run: PMC counter figures collected by the builder with rocprofv3 (separate
# --pmc passes) and committed under profiles/.  They are labelled with the file and the commit that added it.
PMC_TRAFFIC_FILE = ("profiles", "r06_pmc_traffic.json")     # tools/run_pmc.sh: FETCH_SIZE / WRITE_SIZE passes, keyed by kernel symbol
PMC_COMPUTE_FILE = ("profiles", "r06_pmc_compute.json")     # tools/run_pmc_compute.sh: SQ instruction / cycle counters, by kernel symbol
PMC_SOURCES = {
    "traffic": "builder-measured, not this run: profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
               "tools/run_pmc.sh; the file records the commit and the kernel symbols it was collected on)",
    "flop": "builder-measured, not this run: profiles/r06_pmc_compute.json (rocprofv3 --pmc SQ_INSTS_VALU_* / MFMA_MOPS, tools/run_pmc_compute.sh)",
    "launch_floor": "builder-measured, not this run: profiles/r02_launch_floor.json @ 333de90 (tools/ubench/launch_floor.hip)",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--variant", default="e2e", choices=["e2e", "indi"])
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--gates-ahead", type=int, default=1)
    ap.add_argument("--repeats", type=int, default=5, help="timed repetitions of the >= 20 ms bracket (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true")
    ap.add_argument("--no-residual", action="store_true", help="experiment: E2E without the residual MLPs")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the indi / ppo / predecessor / host-path objects")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


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
    if os.environ.get("QR_ROLLOUT_FORM"):   # A/B runs of the probe tools: auto | multi_wave | general | general_multi_wave (the library reads no env var)
        env.set_rollout_form(os.environ["QR_ROLLOUT_FORM"])
    return env


# ---------------------------------------------------------------------------------------------------------------------
# distributed plumbing: one process per GPU (RCCL) -- or per CPU rank over gloo in tests/test_bench_gloo.py
# ---------------------------------------------------------------------------------------------------------------------
class Runtime:
    def __init__(self, rank=0, local_rank=0, world=1, device=None, use_cuda=True, collectives=None):
        self.rank, self.local_rank, self.world, self.device, self.use_cuda = rank, local_rank, world, device, use_cuda
        # collectives on: a process group exists.  Always for world > 1; QR_BENCH_FORCE_DIST=1 also turns it on for ONE rank, so
        # that the RCCL code path (init, barrier, MAX all-reduce, all-gather, config 4) can be exercised on a 1-GPU box
        self.collectives = (world > 1) if collectives is None else bool(collectives)

    @classmethod
    def from_env(cls, expected_world, standin=False):
        """One rank of the job, as the launcher (torch.distributed.run: the driver's, or spawn_ranks() below) described it in the
        environment.  The world size the launcher gave, the one the process group reports and --gpus must all agree: a run that
        would print another n_gpus than it was asked for exits non-zero instead.  `standin`: CPU ranks over gloo (tests only)."""
        import torch
        import torch.distributed as dist

        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if expected_world != world:
            raise SystemExit(f"bench.py: --gpus {expected_world} but WORLD_SIZE={world}: refusing to print a line for another job size")
        collectives = world > 1 or os.environ.get("QR_BENCH_FORCE_DIST", "0") == "1"
        if standin:
            device, backend, kw = torch.device("cpu"), "gloo", {}
        else:
            assert torch.cuda.is_available(), "bench.py needs an MI355X"
            if local_rank >= torch.cuda.device_count():
                raise SystemExit(f"bench.py: rank {rank} (local rank {local_rank}) has no GPU: {torch.cuda.device_count()} visible")
            torch.cuda.set_device(local_rank)
            device, backend, kw = torch.device("cuda", local_rank), "nccl", {"device_id": torch.device("cuda", local_rank)}
        if collectives:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            if dist.get_world_size() != expected_world:
                raise SystemExit(f"bench.py: --gpus {expected_world} but the process group has {dist.get_world_size()} ranks")
        return cls(rank, local_rank, world, device, not standin, collectives)

    def env_id_base(self, envs_per_rank):
        """global index of this rank's env 0: rank r owns global envs [r n, (r + 1) n) (keys the reset RNG stream)"""
        return self.rank * int(envs_per_rank)

    def sync(self):
        if self.use_cuda:
            import torch

            torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.collectives:
            import torch.distributed as dist

            dist.barrier()
            self.sync()

    def max_over_ranks(self, x):
        if not self.collectives:
            return float(x)
        import torch
        import torch.distributed as dist

        t = torch.tensor([float(x)], device=self.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        if self.collectives:
            import torch.distributed as dist

            dist.barrier()
            dist.destroy_process_group()


def timed_region(rt, fn, repeats, min_ms=MIN_TIMED_MS, max_reps=100000, batch=1):
    """Time fn() (= EXACTLY K steps) with barrier + synchronize on both sides.  fn is repeated R times inside one bracket
    so that the bracket holds >= min_ms of work (R is derived from an all-reduced estimate: identical on every rank);
    returns (seconds per fn() = median bracket / R, maximum over ranks; list of per-bracket seconds per fn(); R).
    `batch` > 1: one call of fn() runs the K-step region `batch` times (a replayed graph of `batch` launches); times are still
    reported per K-step region and R counts regions."""
    def bracket(R):
        rt.barrier()
        t0 = time.perf_counter()
        for _ in range(max(1, R // batch)):
            fn()
        rt.barrier()
        return (time.perf_counter() - t0) * (R / (batch * max(1, R // batch)))   # per R regions

    R = batch
    for _ in range(6):  # grow R until one bracket holds >= min_ms (decisions on the all-reduced time: same R on every rank)
        el = rt.max_over_ranks(bracket(R))
        if el >= min_ms * 1e-3 or R >= max_reps:
            break
        R = int(min(max_reps, max(R + 1, math.ceil(1.15 * R * min_ms * 1e-3 / max(el, 1e-7)))))
        R = (R + batch - 1) // batch * batch
    for _ in range(4):
        ts = [bracket(R) / R for _ in range(max(1, repeats))]
        # the calibration bracket can be a slow outlier (first touch, a noisy host): if the MEDIAN bracket of the measurement
        # came out shorter than asked for, grow R on the all-reduced figure (same decision on every rank) and measure again
        med = rt.max_over_ranks(float(np.median(ts)))
        if med * R >= min_ms * 1e-3 or R >= max_reps:
            break
        R = int(min(max_reps, max(R + 1, math.ceil(1.25 * min_ms * 1e-3 / max(med, 1e-9)))))
        R = (R + batch - 1) // batch * batch
    return med, ts, R


def graph_of_launches(rt, fn, batch):
    """`batch` back-to-back calls of fn() -- each ONE K-step launch through the C ABI -- captured into a graph and replayed:
    consecutive kernel nodes of a graph start ~1 us sooner after each other than stream launches (tools/ubench/launch_floor.hip),
    which is what a short K-step region (the driver's --steps 20 = one 60 us kernel) otherwise loses between launches.
    Returns (replay callable, batch) or (fn, 1) when capture is not possible (CPU stand-in, capture error)."""
    if not rt.use_cuda or batch <= 1:
        return fn, 1
    import torch

    try:
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (e.g. the RCCL watchdog of a multi-rank run) may keep calling the runtime
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(batch):
                fn()
        g.replay()
        torch.cuda.synchronize()
        return g.replay, batch
    except Exception:  # pragma: no cover
        torch.cuda.synchronize()
        return fn, 1


def _load_json(*path):
    try:
        return json.load(open(os.path.join(ROOT, *path)))
    except Exception:
        return {}


def exchange_probe(rt, obs, rew, done, gather=None):
    """rollout-boundary exchange (config 4): the rollout buffers as the kernels wrote them -- obs f32, reward f32, done u8 -- go
    out with one all_gather_into_tensor each into receive buffers [world][K][n][...] that are allocated once
    (optimal_quad_control_rl_amd.sharded.RolloutGather): no packing pass, no float copy of `done`, no rearranging copy."""
    from optimal_quad_control_rl_amd.sharded import RolloutGather

    gather = gather or RolloutGather()
    g = gather.gather(obs, rew, done)   # first call allocates the receive buffers
    rt.barrier()
    t0 = time.perf_counter()
    g = gather.gather(obs, rew, done)
    rt.barrier()
    dt = rt.max_over_ranks(time.perf_counter() - t0)
    nbytes = sum(t.numel() * t.element_size() for t in (obs, rew, done))
    return {"op": "3 x all_gather_into_tensor (obs f32, reward f32, done u8) into preallocated [world][K][n][...] buffers",
            "steps": int(obs.shape[0]), "bytes_per_rank": nbytes, "ms": dt * 1e3,
            "GBps_in_per_gpu": nbytes * (rt.world - 1) / dt / 1e9, "gathered_shape": list(g.obs.shape), "gathered_bytes": g.nbytes}


def measure_env(rt, env, variant, n, K, W, ga, repeats, closed_loop=True):
    """Times the fused rollout, the per-step launches and the closed-loop kernel of one env shard; returns the objects of
    the JSON line (rates are whole-job: n x world envs)."""
    import torch

    L, dev = env.state_len, env.device
    gen = torch.Generator(device=dev).manual_seed(rt.rank)  # torch Philox, seed = rank
    KB = max(K, W, 1)
    actions = torch.rand((KB, n, 4), device=dev, generator=gen) * 2 - 1
    out = (torch.empty((KB, n, L), dtype=torch.float32, device=dev), torch.empty((KB, n), dtype=torch.float32, device=dev),
           torch.empty((KB, n), dtype=torch.uint8, device=dev), torch.empty((KB, n), dtype=torch.uint8, device=dev))

    def view(k):
        return tuple(t[:k] for t in out)

    env.reset_device()
    if W > 0:
        env.rollout_device(actions[:W], view(W))
    total_steps = n * rt.world * K
    vidx = 0 if variant == "e2e" else 1
    # the symbol rocprofv3 will list for the fused launch (the library's own selection: env count, variant, mode), and the counter
    # evidence collected under exactly that name
    fused_symbol = env.rollout_kernel_name() if hasattr(env, "rollout_kernel_name") else f"qr::rollout_kernel<{vidx}, {ga}>"
    step_symbol = f"qr::step_kernel<{vidx}, {ga}>"
    policy_symbol = f"qr::rollout_policy_kernel<{vidx}, {ga}>"
    pmc_all = _load_json(*PMC_TRAFFIC_FILE)
    pmc = pmc_all.get(f"n{n}", {})
    pmcc = _load_json(*PMC_COMPUTE_FILE).get("kernels", {})
    floor = _load_json("profiles", "r02_launch_floor.json").get("us_per_launch", {})
    bytes_per_step = BYTES_PER_ENV_STEP[variant](ga) * n

    # (1) fused rollout: qr_step_many = ONE kernel for the K steps, env state register-resident between steps.  A short region
    # (the driver's --steps 20 is one ~60 us kernel) is replayed as a graph of `gb` such launches, so that the gaps between
    # launches do not dominate it; the library's per-call hipEvent bracket is off inside the timed region and switched back on for
    # ONE extra launch, whose duration is the kernel time of the roofline.
    def one_region():
        env.rollout_device(actions[:K], view(K))

    timing_knob = hasattr(env, "set_timing")
    if timing_knob:
        env.set_timing(False)
    fused_fn, gb = graph_of_launches(rt, one_region, max(1, min(16, -(-320 // max(K, 1)))))
    fused_s, fused_ts, fused_R = timed_region(rt, fused_fn, repeats, batch=gb)
    if timing_knob:
        env.set_timing(True)
    kernel_ms_samples = []
    for _ in range(3):       # (information only) the library's hipEvent pair around ONE launch: it brackets two marker packets as well,
        for _ in range(3):   # 4-5 us on a 20-step launch
            one_region()
        kernel_ms_samples.append(env.last_rollout_ms())
    fused_event_single_ms = float(np.median(kernel_ms_samples))
    # THE kernel time of the roofline -- one fixed measurement (no min() over different clocks): HIP events on the launch
    # stream (torch's current stream is the stream the library launches on) around a region of back-to-back launches, divided by the
    # number of launches, median of five regions.  The wall-clock bracket per launch and the single-launch event pair are printed
    # beside it and a disagreement beyond 10 % is flagged, not hidden.  rocprofv3's average for the same command: profiles/.
    if rt.use_cuda:
        regs = []
        n_rep = max(1, int(round(0.01 / max(fused_s * gb, 1e-7))))   # ~10 ms of launches per region
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n_rep):
                fused_fn()
            e1.record()
            e1.synchronize()
            regs.append(e0.elapsed_time(e1) / (n_rep * gb))
        fused_kernel_ms = float(np.median(regs))
    else:
        fused_kernel_ms = fused_event_single_ms
    fused_event_ms = fused_event_single_ms
    launch_s = fused_kernel_ms * 1e-3
    flop = pmcc.get(fused_symbol, {}).get("derived", {}).get("f32_flop_per_env_step")
    traffic = (pmc.get(fused_symbol) or {}).get("hbm_bytes_per_step")
    fused_bytes = FUSED_BYTES_PER_ENV_STEP[variant](ga) * n
    ach = fused_bytes * K / launch_s / 1e9
    tf = None if flop is None else flop * n * K / launch_s / 1e12
    roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None if traffic is None else traffic * K,
                "kernel": fused_symbol, "steps_per_launch": K,
                "launch_us": fused_kernel_ms * 1e3, "launch_us_is": "HIP events around a region of back-to-back launches / launches",
                "launch_us_hipevent": fused_event_ms * 1e3, "launch_us_bracket": fused_s * 1e6,
                "timing_consistent": bool(abs(fused_kernel_ms - fused_s * 1e3) <= 0.1 * fused_s * 1e3),
                "us_per_step": fused_kernel_ms * 1e3 / K,
                "traffic_commit": pmc_all.get("commit"), "traffic_kernel": fused_symbol if traffic is not None else None,
                "bytes_per_launch": fused_bytes * K, "bytes_per_env_step": FUSED_BYTES_PER_ENV_STEP[variant](ga),
                "note": "algorithmic bytes of THIS kernel: the env state stays in registers for the K steps, so a step moves its action in "
                        "and obs / reward / done / trunc out (PMC-measured traffic agrees within 3 percent); the SURVEY 8(d) per-step figure "
                        "(%d B, state read + written every step) describes the per-step kernel and is used only there"
                        % BYTES_PER_ENV_STEP[variant](ga),
                "valu": {"bound": "valu", "achieved": tf, "peak": VALU_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": None if tf is None else tf / VALU_F32_PEAK_TF, "flop_per_env_step": flop,
                         "flop_source": "PMC: 64 x (ADD + MUL + TRANS + 2 FMA f32 wave-instructions) + 512 x MFMA_MOPS_F32 per env-step, "
                                        "profiles/r06_pmc_compute.json; the larger of the two fractions names the binding roof"}}
    roofline["frac_on_8d_bytes"] = bytes_per_step * K / launch_s / 1e9 / HBM_PEAK_GBS
    roofline["frac_on_8d_bytes_note"] = ("SURVEY 8(d)'s %d B per env-step (state read + written every step) divided by this kernel's time: "
                                         "NOT its traffic (the state stays in registers) -- BASELINE.md section 4's throughput yardstick only"
                                         % BYTES_PER_ENV_STEP[variant](ga))
    res = {"value": total_steps / fused_s, "value_kernel_only": n * K / launch_s * rt.world,
           "launch": ("graph of %d K-step launches, replayed" % gb) if gb > 1 else "stream launches",
           "ms_per_step": fused_s * 1e3 / K, "repeats": len(fused_ts), "launches_per_bracket": fused_R,
           "timed_ms_per_bracket": fused_s * fused_R * 1e3, "all_ms_per_step": [t * 1e3 / K for t in fused_ts], "roofline": roofline}

    # (2) per-step launches: K step kernels (one per env.step()), captured once into a graph by qr_step_launches and replayed
    step_s, step_ts, step_R = timed_region(rt, lambda: env.step_sequence_device(actions[:K], view(K)), repeats)
    region_ms = env.last_rollout_ms()       # hipEvents bracketing the K back-to-back step kernels on the launch stream
    res["done_fraction"] = float(out[2][:K].float().mean().item())
    kernel_ms, Kp = region_ms / K, K
    if rt.use_cuda and K < 200:
        # the hipEvent pair around a SHORT sequence (the driver's --steps 20) adds its two marker packets to 20 kernels (+ 0.2 us each):
        # the per-step KERNEL time is taken from a sequence of 200 launches of the same kernel on the same env (own buffers, capped
        # at 2 GB of observations); `value` of this leg stays the K-step bracket
        Kp = int(max(K, min(200, 2e9 // (n * L * 4))))
        if Kp > K:
            gen2 = torch.Generator(device=dev).manual_seed(1000 + rt.rank)
            a2 = torch.rand((Kp, n, 4), device=dev, generator=gen2) * 2 - 1
            o2 = (torch.empty((Kp, n, L), dtype=torch.float32, device=dev), torch.empty((Kp, n), dtype=torch.float32, device=dev),
                  torch.empty((Kp, n), dtype=torch.uint8, device=dev), torch.empty((Kp, n), dtype=torch.uint8, device=dev))
            samples = []
            for _ in range(4):
                env.step_sequence_device(a2, o2)
                samples.append(env.last_rollout_ms() / Kp)
            kernel_ms = float(np.median(samples[1:]))
            del a2, o2
    step_timing_consistent = bool(kernel_ms <= 1.1 * step_s * 1e3 / K)   # events over the region vs the wall-clock bracket: flagged, never mixed
    ach = bytes_per_step / (kernel_ms * 1e-3) / 1e9
    res["per_step_launch"] = {
        "what": "qr_step_launches: the same K steps as K step kernels (one per env.step(); bit-identical outputs), the "
                "closed-loop calling pattern; the K launches are one captured graph, replayed",
        "value": total_steps / step_s, "unit": "env-steps/s", "ms_per_step": step_s * 1e3 / K, "launches_per_bracket": step_R * K,
        "timed_ms_per_bracket": step_s * step_R * 1e3, "all_ms_per_step": [t * 1e3 / K for t in step_ts],
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": (pmc.get(step_symbol) or {}).get("hbm_bytes_per_launch"), "kernel": step_symbol,
                     "kernel_us": kernel_ms * 1e3, "bytes_per_launch": bytes_per_step, "launches_timed": Kp, "timing_consistent": step_timing_consistent,
                     "launch_floor_us": {k: floor.get(k) for k in ("empty_b256", "empty_b256_graph", "copy_nt_b256", "copy_nt_b256_graph")
                                         if k in floor},
                     "launch_floor_note": "tools/ubench/launch_floor.hip at the same shape (256 workgroups x 256 threads, back-to-back "
                                          "dependent launches): empty kernel / a kernel that only moves the E2E step's 285 B per env"}}

    # (3) closed loop (config 5 collect phase): policy MLP + sampling + env step in one kernel
    if closed_loop:
        try:
            from optimal_quad_control_rl_amd.policy import MfmaPolicy
            from optimal_quad_control_rl_amd.ppo import ActorCritic

            torch.manual_seed(0)
            net = ActorCritic(L, 4).to(dev)  # random-init weights of the reference's policy architecture (R:783)
            pol = MfmaPolicy(L, dev.index).load_torch(net.pi)
            Kc = min(K, 256)
            state = {"out": None, "r": 0}

            def cl():
                r = env.rollout_policy_device(pol, Kc, torch.zeros(4), noise_seed=rt.rank, first_step=state["r"] * Kc, out=state["out"])
                state["out"], state["r"] = r[:6], state["r"] + 1

            cl()
            cl_s, cl_ts, cl_R = timed_region(rt, cl, min(repeats, 3))
            cl_ms = cl_s * 1e3   # bracket time per launch (back-to-back launches; rocprofv3's per-kernel average agrees with it,
            #                      while the hipEvent pair of a queued launch can include part of its predecessor)
            pol_flop = 2.0 * (16 * ((L + 16) // 16) * 128 + 2 * 128 * 128 + 128 * 32)   # f16 MACs x 2 as issued (padded tiles): PMC 81 920 at L = 17 / 24
            pol_flop_useful = 2.0 * ((L + 1) * 120 + 2 * 121 * 120 + 121 * 4)           # the network's own multiply-adds (biases included)
            cl_tf = pol_flop * n * Kc / (cl_ms * 1e-3) / 1e12
            cl_bytes = (pmc.get(policy_symbol) or {}).get("hbm_bytes_per_step")
            res["closed_loop"] = {
                "what": "qr_rollout_policy: K x [obs -> policy MLP (L->120->120->120->4, f16 MFMA) -> Gaussian sample -> env.step] in "
                        "ONE kernel (PPO collect phase); random-init policy weights",
                "steps": Kc, "ms_per_step": cl_s * 1e3 / Kc, "value": n * rt.world * Kc / cl_s, "unit": "env-steps/s",
                "kernel_us_per_step": cl_ms * 1e3 / Kc,
                "roofline": {"bound": "mfma", "achieved": cl_tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": cl_tf / MFMA_F16_PEAK_TF,
                             "flop_per_env_step": pol_flop, "flop_useful_per_env_step": pol_flop_useful,
                             "frac_useful": pol_flop_useful * n * Kc / (cl_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
                             "kernel": policy_symbol,
                             "traffic": None if cl_bytes is None else cl_bytes * Kc,
                             "hbm": None if cl_bytes is None else {
                                 "achieved": cl_bytes / (cl_ms * 1e-3 / Kc) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": cl_bytes / (cl_ms * 1e-3 / Kc) / 1e9 / HBM_PEAK_GBS},
                             "note": "f16 matrix-core flop of the policy MLP as issued (160-180 v_mfma_f32_32x32x16_f16 per 64 envs); the "
                                     "kernel is one wave per SIMD: issue-order-bound between MFMA chain, sampling and the env step "
                                     "(profiles/r02_pmc_compute.json: MFMA busy ~39 percent of wave cycles)"}}
            del state
        except Exception as ex:  # pragma: no cover
            res["closed_loop"] = {"error": repr(ex)}
    res["_buffers"] = (out, K)
    return res


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
    for _ in range(10):
        env.step(acts)
    per_step = []
    for _ in range(40):   # median of per-call times: a host-side hiccup (page faults of fresh pinned buffers, a busy neighbour on the
        t0 = time.perf_counter()   # box) in one call must not set the figure
        env.step(acts)
        per_step.append(time.perf_counter() - t0)
    dt = float(np.median(per_step))
    env.close()
    return {"what": "env.step(numpy actions) -> numpy obs/reward/done/infos (SB3 calling convention), PCIe + NumPy inclusive",
            "ms_per_step": dt * 1e3, "value": n / dt, "unit": "env-steps/s"}


def config5_probe(n, iters=3, mb=16384, n_steps=32):
    """BASELINE config 5 as a whole loop, short: PPO on the 4-gate square track with the E2E model (residual MLPs +
    disturbances), reference hyper-parameters where they are the reference's (gamma 0.999, 10 epochs, constant lr 3e-4, target_kl
    None, 3 x 120 ReLU nets, R:784-795; SB3's time-limit bootstrap) and this build's rollout shape (n envs x 32 steps).  Collect =
    qr_rollout_policy, GAE = qr_ppo_gae, update = qr_ppo_epoch (all epochs of a train() as one replayed graph, permutations drawn on
    the device); every one of the epochs x minibatches updates runs.  `mb` rows per minibatch: 16 384 (128 minibatches per epoch,
    the round-2 recipe) or 65 536 (32 per epoch); n_steps = 40 with mb = n x 40 / 20 is the REFERENCE'S OWN SPLIT -- 20 minibatches
    per epoch (100 envs x 1000 steps / batch_size 5000, R:785-792).  Reports end-to-end env-steps/s."""
    import torch
    from optimal_quad_control_rl_amd import Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, square_track
    from optimal_quad_control_rl_amd.ppo import PPO

    env = Quadcopter3DGates(n, *square_track(), gates_ahead=1, infos_mode="none", seed=1)
    env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
    mb = min(mb, n * n_steps)
    model = PPO(env, seed=0, gamma=0.999, n_steps=n_steps, n_epochs=10, batch_size=mb, learning_rate=3e-4,
                target_kl=None, fused_collect=True, native_update=True)
    model.collect(); model.train()   # warm-up iteration (captures the epoch graph)
    model.collect(); model.train()
    torch.cuda.synchronize()
    applied0, skipped0 = model.stats["updates"], model.stats["skipped_nonfinite"]
    t0 = time.perf_counter()
    tc = 0.0
    for _ in range(iters):
        c0 = time.perf_counter()
        model.collect()
        torch.cuda.synchronize()
        tc += time.perf_counter() - c0
        model.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = iters * n * n_steps
    updates = iters * 10 * ((n * n_steps) // mb)
    applied, skipped = model.stats["updates"] - applied0, model.stats["skipped_nonfinite"] - skipped0
    env.close()
    # "every update taken" is CHECKED, not assumed: the device counts optimiser steps really applied (a non-finite gradient norm
    # skips one; round 3 found a graph-replay bug that silently skipped whole epochs this way)
    assert applied == updates and skipped == 0 and not model.stats["early_stop"], (applied, skipped, updates)
    return {"what": "config 5, whole loop (fused collect + GAE + all %d updates per rollout of %d rows each, constant lr, no early stop), "
                    "%d iterations" % (updates // iters, mb, iters),
            "envs": n, "n_steps": n_steps, "minibatch": mb, "epochs": 10, "value": steps / dt, "unit": "env-steps/s",
            "collect_ms_per_rollout": tc / iters * 1e3, "update_ms_per_rollout": (dt - tc) / iters * 1e3,
            "us_per_update": (dt - tc) / updates * 1e6, "updates_applied": applied, "updates_skipped": skipped}


def cpu_quota():
    """CPU bandwidth the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown.  The
    affinity mask can list every core of the box while the cgroup grants a few: threads beyond the quota only add contention (round 3:
    43.9 M env-steps/s at 16 threads, 0.3 M at 256 on a box whose affinity mask said 256)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except Exception:
        pass
    return None


def cpu_baseline(variant, n, ga, seconds):
    """The oracle (C port of the reference, parity-pinned against reference fixtures) on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from optimal_quad_control_rl_amd import TRAIN_DISTURBANCE_RANGES, square_track, zigzag_track
    from optimal_quad_control_rl_amd.vec_env import default_residual_blob

    affinity = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    cores = affinity if quota is None else max(1, min(affinity, int(math.ceil(quota))))   # the thread sweep stops at the quota
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
            "value_1core": out[1][0], "host_cores": affinity, "cpu_quota_cores": quota,
            "threads_swept_up_to": cores}


def rccl_report(rt, local_ms_per_step, local=None):
    """What the collectives backend saw, so that a mis-launched multi-GPU run is visible in the JSON: world size and backend from
    the process group, the RCCL version, and every rank's OWN ms_per_step (the headline uses the maximum).  `local` = this rank's
    diagnosis row (kernel symbol, kernel time, device clock, device name): gathered as objects so that a straggler rank -- a
    different kernel, a throttled clock, a slower box -- is visible in the first real --gpus 8 line."""
    if not rt.collectives:
        return None
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(local_ms_per_step), float(rt.local_rank)], device=rt.device, dtype=torch.float64)
    allv = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, t)
    rep = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(),
           "per_rank_ms_per_step": [float(v[0]) for v in allv], "per_rank_local_rank": [int(v[1]) for v in allv]}
    if local is not None:
        rows = [None] * dist.get_world_size()
        dist.all_gather_object(rows, local)
        for key in sorted({k for r in rows if r for k in r}):
            rep["per_rank_" + key] = [None if r is None else r.get(key) for r in rows]
    if rt.use_cuda:
        try:
            rep["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as ex:  # pragma: no cover
            rep["rccl_version"] = repr(ex)
        rep["device"] = torch.cuda.get_device_name(rt.device)
        rep["visible_gpus"] = torch.cuda.device_count()
    return rep


def _short(x, n=160):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 3] + "..."


def headline(result):
    """The ONE line the driver parses: the contract's fields plus a FLAT `roofline` object that carries every fraction quoted in
    DESIGN.md section 5 (fused kernel on its own bytes, the vector-flop view, SURVEY 8(d)'s yardstick, the per-step kernel, the
    closed loop with as-issued and useful flop) and the provenance of anything not measured by this run.  Everything else stays in
    the full object (stderr + gpurun_out/bench_full_*.json).  Kept under 4 KB."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "value_kernel_only", "launch")
    h = {k: result[k] for k in keep if k in result}
    c = result["config"]
    h["config"] = {"workload": _short(c["workload"], 220), "envs_per_gpu": c["envs_per_gpu"], "variant": c["variant"],
                   "gates_ahead": c["gates_ahead"], "obs_len": c["obs_len"]}
    r = result.get("roofline", {})
    flat = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "steps_per_launch", "launch_us",
                                  "launch_us_hipevent", "launch_us_bracket", "timing_consistent", "us_per_step", "bytes_per_env_step",
                                  "frac_on_8d_bytes", "traffic_commit")}
    flat["frac_on_8d_bytes_is"] = "throughput yardstick of BASELINE.md section 4 (SURVEY 8(d) bytes / this kernel's time), NOT traffic: the state stays in registers"
    flat["traffic_source"] = PMC_SOURCES["traffic"]
    v = r.get("valu") or {}
    flat["valu_frac"], flat["valu_flop_per_env_step"], flat["valu_flop_source"] = v.get("frac"), v.get("flop_per_env_step"), PMC_SOURCES["flop"]
    ps = result.get("per_step_launch") or {}
    pr = ps.get("roofline") or {}
    flat.update({"per_step_kernel": pr.get("kernel"), "per_step_kernel_us": pr.get("kernel_us"), "per_step_frac": pr.get("frac"),
                 "per_step_bytes_per_launch": pr.get("bytes_per_launch"), "per_step_traffic": pr.get("traffic"),
                 "per_step_timing_consistent": pr.get("timing_consistent"),
                 "per_step_value": ps.get("value")})
    cl = result.get("closed_loop") or {}
    cr = cl.get("roofline") or {}
    flat.update({"closed_loop_us_per_step": cl.get("kernel_us_per_step"), "closed_loop_value": cl.get("value"),
                 "closed_loop_mfma_frac_as_issued": cr.get("frac"), "closed_loop_mfma_frac_useful": cr.get("frac_useful"),
                 "closed_loop_flop_as_issued": cr.get("flop_per_env_step"), "closed_loop_flop_useful": cr.get("flop_useful_per_env_step")})
    h["roofline"] = flat
    cb = result.get("cpu_baseline")
    if cb:
        h["cpu_baseline"] = {k: _short(cb.get(k), 200) for k in ("value", "unit", "cores", "kind", "sample", "value_1core", "host_cores",
                                                                    "cpu_quota_cores", "threads_swept_up_to")}
    o = result.get("indi") or result.get("e2e") or {}
    if o and "error" not in o:
        orr, ops = o.get("roofline") or {}, (o.get("per_step_launch") or {})
        h["other_variant"] = {"variant": (o.get("config") or {}).get("variant"), "value": o.get("value"), "ms_per_step": o.get("ms_per_step"),
                              "frac": orr.get("frac"), "us_per_step": orr.get("us_per_step"),
                              "per_step_kernel_us": (ops.get("roofline") or {}).get("kernel_us"),
                              "per_step_frac": (ops.get("roofline") or {}).get("frac"),
                              "cpu_baseline_value": (o.get("cpu_baseline") or {}).get("value")}
    pu, c5, c5b = result.get("ppo_update") or {}, result.get("config5") or {}, result.get("config5_mb65536") or {}
    c5r = result.get("config5_ref_ratio") or {}
    if pu or c5:
        h["ppo"] = {"dtype": "f16 matrix-core operands, f32 accumulation / parameters / Adam (precision='f32' = f32-class kernels, three bf16 pieces per operand: A/B path, update_f32class_*)",
                    "update_us_per_16384_rows": pu.get("epoch_us"), "update_useful_TFLOPs": pu.get("epoch_useful_TFLOPs"),
                    "update_f32class_us_per_16384_rows": pu.get("f32class_us"), "update_f32class_us_per_5000_rows": pu.get("f32class_us_5000_rows"),
                    "update_mfma_frac": None if pu.get("epoch_useful_TFLOPs") is None else pu["epoch_useful_TFLOPs"] / MFMA_F16_PEAK_TF,
                    "update_launch": "qr_ppo_epoch: one replayed graph per epoch (what training calls)",
                    "stream_launch_us_per_update": pu.get("native_us"), "torch_us": pu.get("torch_us"),
                    "config5_value": c5.get("value"), "config5_us_per_update": c5.get("us_per_update"),
                    "config5_what": _short(c5.get("what"), 150),
                    "config5_mb65536_value": c5b.get("value"), "config5_mb65536_us_per_update": c5b.get("us_per_update"),
                    "config5_20_minibatches_per_epoch_value": c5r.get("value"), "config5_20_minibatches_per_epoch_rows": c5r.get("minibatch"),
                    "config5_20_minibatches_per_epoch_us_per_update": c5r.get("us_per_update")}
    if "parity" in result:
        h["parity"] = {k: result["parity"].get(k) for k in ("max_rel_dstate_100_steps", "tolerance", "error") if k in result["parity"]}
    if result.get("rccl"):
        h["rccl"] = result["rccl"]
    for k in ("exchange", "config4"):
        if k in result:
            e = result[k] if k == "exchange" else dict(result[k], exchange=None)
            h[k] = {kk: _short(vv, 120) for kk, vv in e.items() if kk not in ("what", "exchange") and not isinstance(vv, (dict, list))}
            if k == "config4":
                h[k]["exchange_ms"] = result[k]["exchange"]["ms"]
                h[k]["exchange_GBps_in_per_gpu"] = result[k]["exchange"]["GBps_in_per_gpu"]
    h["full_object"] = result.get("_full_path")
    # stay under 4 KB whatever the run produced (eight ranks add per-rank lists): shed explanatory strings first, then optional objects
    shed = [("roofline", "frac_on_8d_bytes_is"), ("roofline", "valu_flop_source"), ("ppo", "config5_what"), ("ppo", "update_launch"), ("ppo", "dtype"),
            ("cpu_baseline", "sample"), ("config", "workload"), ("roofline", "traffic_source"), ("parity", None), ("other_variant", None),
            ("ppo", None), ("exchange", None), ("config4", None)]
    for obj, key in shed:
        if len(json.dumps(h)) < 3900:
            break
        if obj in h and key is None:
            h[obj] = "see full_object"
        elif obj in h and isinstance(h[obj], dict) and key in h[obj]:
            h[obj][key] = _short(str(h[obj][key]), 40)
    return h


def workload_text(variant, n, ga):
    return (f"{n} envs/GPU, " + ("Bebop E2E (motor-cmd actions) + NNDroneModel residual MLPs + training disturbance ranges, 7-gate zigzag"
                                 if variant == "e2e" else "INDI inner-loop variant, 4-gate square (x2)")
            + f", gates_ahead={ga}, U(-1,1) actions pre-generated on device [K][N][4], outputs to a [K][N] rollout buffer")


def run(args, rt, env_factory=make_env, closed_loop=True):
    """The measurement proper (also driven with a CPU stand-in over gloo by tests/test_bench_gloo.py)."""
    n, K, W, ga = args.envs, args.steps, args.warmup, args.gates_ahead
    env = env_factory(args.variant, n, ga, rt.env_id_base(n), 0, None if args.no_residual else "default")
    m = measure_env(rt, env, args.variant, n, K, W, ga, args.repeats, closed_loop=closed_loop)
    (out, _) = m.pop("_buffers")
    L = env.state_len
    result = {
        "metric": "env-steps/sec at N=%d envs per GPU (Quadcopter3DGates.step, " % n
                  + ("E2E + residual MLPs" if args.variant == "e2e" else "INDI inner loop") + ")",
        "value": m.pop("value"), "unit": "env-steps/s", "n_gpus": rt.world, "steps": K, "warmup": W,
        "ms_per_step": m.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_text(args.variant, n, ga), "envs_per_gpu": n, "variant": args.variant, "gates_ahead": ga,
                   "obs_len": L, "sharding": f"{rt.world} independent shard(s), env_id_base = rank*N"},
        "path": "qr_step_many (fused K-step rollout kernel)",
    }
    result.update(m)
    local = {"kernel": (m.get("roofline") or {}).get("kernel"), "kernel_us_per_step": (m.get("roofline") or {}).get("us_per_step")}
    if rt.use_cuda:
        import torch
        prop = torch.cuda.get_device_properties(rt.device)
        local["clock_mhz_max"] = getattr(prop, "clock_rate", 0) / 1e3 or None
        local["device"] = prop.name
    result["rccl"] = rccl_report(rt, float(np.median(m["all_ms_per_step"])), local)
    result["provenance"] = PMC_SOURCES

    # --- rollout-boundary exchange: RCCL all-gather of [obs | reward | done] of this run's shard ------------------------
    if rt.collectives and not args.no_exchange:
        Kx = min(K, 64)
        result["exchange"] = exchange_probe(rt, out[0][:Kx], out[1][:Kx], out[2][:Kx])
    del out

    # --- BASELINE config 4: 32 768 envs per GPU (262 144 on 8), all-gather of every rollout --------------------------------
    if rt.collectives and not args.no_exchange and not args.no_extras:
        import torch

        n4, K4 = 32768 if n >= 32768 else n, min(K, 64)
        env4 = env_factory(args.variant, n4, ga, rt.env_id_base(n4), 0, None if args.no_residual else "default")
        env4.reset_device()
        dev = env4.device
        gen = torch.Generator(device=dev).manual_seed(1000 + rt.rank)
        a4 = torch.rand((K4, n4, 4), device=dev, generator=gen) * 2 - 1
        o4 = None
        state = {}

        def roll4():
            state["o"] = env4.rollout_device(a4, state.get("o"))

        roll4()
        s4, ts4, R4 = timed_region(rt, roll4, min(args.repeats, 3))
        ex4 = exchange_probe(rt, state["o"][0], state["o"][1], state["o"][2])
        sim_ms = s4 * 1e3
        result["config4"] = {
            "what": "BASELINE config 4: %d envs sharded over %d GPUs (%d per GPU); per rollout of %d steps: fused rollout kernel, then one "
                    "RCCL all-gather of the packed [obs | reward | done] shard" % (n4 * rt.world, rt.world, n4, K4),
            "envs_total": n4 * rt.world, "envs_per_gpu": n4, "steps": K4, "simulate_ms": sim_ms, "simulate_value": n4 * rt.world * K4 / s4,
            "exchange": ex4, "xgmi_in_peak_GBps_per_gpu": 7 * 153.0,
            "value_incl_exchange": n4 * rt.world * K4 / (s4 + ex4["ms"] * 1e-3), "unit": "env-steps/s"}
        del state, a4, o4
        env4.close()

    if rt.rank == 0 and rt.world == 1 and rt.use_cuda:
        if not args.no_parity:
            try:
                result["parity"] = parity_probe()
            except Exception as ex:  # pragma: no cover
                result["parity"] = {"error": repr(ex)}
        if not args.no_extras:
            # --- BASELINE config 3 (INDI inner loop) driver-timed beside config 2 -----------------------------------------
            other = "indi" if args.variant == "e2e" else "e2e"
            try:
                env_o = env_factory(other, n, ga, rt.env_id_base(n), 0, "default")
                mo = measure_env(rt, env_o, other, n, K, W, ga, min(args.repeats, 3), closed_loop=closed_loop)
                mo.pop("_buffers")
                mo["unit"] = "env-steps/s"
                mo["config"] = {"workload": workload_text(other, n, ga), "variant": other}
                result[other] = mo
                env_o.close()
            except Exception as ex:  # pragma: no cover
                result[other] = {"error": repr(ex)}
        if not args.no_cpu_baseline and not args.no_extras:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:  # the PPO minibatch update on the matrix cores (qr_ppo_minibatch) vs torch; measured
                # before the CPU legs (their worker threads would compete with the launch thread)
                from bench_ppo_update import measure as ppo_measure

                result["ppo_update"] = ppo_measure(L, 16384, 65536 * 8, 100)
            except Exception as ex:  # pragma: no cover
                result["ppo_update"] = {"error": repr(ex)}
            # each probe under its own guard: a failure of a later, larger one must not erase the results already measured
            # the reference splits a rollout into 20 minibatches per epoch (100 envs x 1000 steps / batch_size 5000, R:785-792):
            # 40 steps per rollout give exactly that split with whole 64-row groups (n x 40 / 20 = 2 n rows per minibatch)
            for key, kw in (("config5", {}), ("config5_mb65536", {"mb": 65536}), ("config5_ref_ratio", {"mb": (n * 40) // 20, "n_steps": 40})):
                try:
                    result[key] = config5_probe(n, **kw)
                except Exception as ex:  # pragma: no cover
                    result[key] = {"error": repr(ex)}
            try:
                result["host_numpy_path"] = host_path_probe(args.variant, n, ga)
            except Exception as ex:  # pragma: no cover
                result["host_numpy_path"] = {"error": repr(ex)}
        # CPU legs LAST: the oracle's OpenMP workers spin for a while after every parallel region, and a host-launched GPU measurement
        # right behind them is starved of its launch thread (that is what the 884 us / 557 us "stream_launch_us_per_update" outliers
        # of the r03 / early r04 lines were: the PPO leg ran straight after the other variant's CPU baseline)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.variant, n, ga, args.cpu_seconds)
            other = "indi" if args.variant == "e2e" else "e2e"
            if not args.no_extras and isinstance(result.get(other), dict) and "error" not in result[other]:
                result[other]["cpu_baseline"] = cpu_baseline(other, n, ga, min(args.cpu_seconds, 6.0))
        if not args.no_cpu_baseline and not args.no_extras:
            try:  # the predecessor envs of "3D quad.ipynb" (include/quad3d.h), short measurement
                from bench_quad3d import measure as q3_measure

                result["predecessor_envs"] = {k: q3_measure(k, n, 200, repeats=3, cpu_seconds=2.0) for k in ("hover", "gates")}
            except Exception as ex:  # pragma: no cover
                result["predecessor_envs"] = {"error": repr(ex)}
    return result


def _standin_factory():
    """TEST HOOK (tests/test_bench_gloo.py): QR_BENCH_TEST_FACTORY="module:callable" names an env factory with make_env()'s
    signature that runs on CPU ranks over gloo, so that the launcher logic of this file (rank spawning, world-size checks, the
    multi-rank bracket) can be exercised where there is no GPU.  A stand-in run labels its line `data: "cpu-standin (test only)"`;
    nothing in a measurement run reads this variable's target."""
    spec = os.environ.get("QR_BENCH_TEST_FACTORY")
    if not spec:
        return None
    import importlib

    mod, _, name = spec.partition(":")
    return getattr(importlib.import_module(mod), name)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks HERE (re-exec under
    torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) and return their exit code.  Fewer than N visible GPUs is
    an error, never a smaller job."""
    import socket
    import subprocess

    if not os.environ.get("QR_BENCH_TEST_FACTORY"):
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this box has {have}: not running a smaller job")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("bench.py: --gpus %d without a launcher: spawning the ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the CPU baseline's OpenMP workers sleep between regions instead of spinning
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    factory = _standin_factory()
    rt = Runtime.from_env(args.gpus, standin=factory is not None)
    if factory is not None:
        result = run(args, rt, env_factory=factory, closed_loop=False)
        result["data"] = "cpu-standin (test only)"
    else:
        result = run(args, rt)
    if rt.rank == 0:
        # the full object: a file next to the profiles scratch (pulled back by gpurun) and stderr; stdout gets ONE compact line, last
        full_path = None
        try:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            full_path = os.path.join(d, "bench_full_n%d_%s_k%d.json" % (rt.world, args.variant, args.steps))
            json.dump(result, open(full_path, "w"), indent=1)
        except Exception:  # pragma: no cover
            full_path = None
        result["_full_path"] = None if full_path is None else os.path.relpath(full_path, ROOT)
        print(json.dumps({k: v for k, v in result.items() if k != "_full_path"}), file=sys.stderr, flush=True)
        print(json.dumps(headline(result)), flush=True)
    rt.finish()


if __name__ == "__main__":
    main()
