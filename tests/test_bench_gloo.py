"""bench.py's multi-rank code path on CPU: two gloo ranks drive bench.run() with a stand-in env built on the TEST-ONLY
oracle -- rank -> env_id_base mapping, the barrier / max-over-ranks timing bracket (same repeat count on every rank), the
rollout-boundary all-gather ("exchange") and the BASELINE config-4 object.  On the GPU box the same function runs the HIP
product over RCCL (python -m torch.distributed.run ... bench.py --gpus N)."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class CpuBenchEnv:
    """The device API bench.measure_env() uses, on the CPU oracle."""

    created = []

    def __init__(self, variant, n, ga, env_id_base, seed=0, residual=None):
        import parity as P
        from oracle import oracle as O
        from oracle_adapter import OracleAdapter

        v = O.E2E if variant == "e2e" else O.INDI
        self.a = OracleAdapter(v, n, P.tracks()["zigzag" if variant == "e2e" else "square"], gates_ahead=ga, seed=seed,
                               env_id_base=env_id_base)
        self.num_envs, self.device = n, torch.device("cpu")
        self.state_len = (16 + 4 * ga + 4) if variant == "e2e" else (13 + 4 * ga)
        self._ms = 0.0
        CpuBenchEnv.created.append((variant, n, env_id_base))

    def reset_device(self):
        return torch.from_numpy(self.a.reset())

    def rollout_device(self, actions, out=None):
        import time

        K = actions.shape[0]
        if out is None:
            out = (torch.empty((K, self.num_envs, self.state_len)), torch.empty((K, self.num_envs)),
                   torch.empty((K, self.num_envs), dtype=torch.uint8), torch.empty((K, self.num_envs), dtype=torch.uint8))
        t0 = time.perf_counter()
        for k in range(K):
            o, r, d, t = self.a.step(actions[k].numpy())
            out[0][k] = torch.from_numpy(o)
            out[1][k] = torch.from_numpy(r)
            out[2][k] = torch.from_numpy(d.astype(np.uint8))
            out[3][k] = torch.from_numpy(t.astype(np.uint8))
        self._ms = (time.perf_counter() - t0) * 1e3
        return out

    step_sequence_device = rollout_device

    def last_rollout_ms(self):
        return self._ms

    def close(self):
        pass



def _free_port():
    """a TCP port the kernel just handed out (a pid-derived port collided with a socket in TIME_WAIT once)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    args = bench.parse(["--gpus", str(world), "--steps", "6", "--warmup", "2", "--envs", "64", "--repeats", "2",
                        "--no-cpu-baseline", "--no-parity", "--variant", "indi"])
    rt = bench.Runtime(rank, 0, world, torch.device("cpu"), use_cuda=False)
    assert rt.env_id_base(64) == rank * 64
    res = bench.run(args, rt, env_factory=CpuBenchEnv, closed_loop=False)
    # every env this rank built is keyed by rank * n (main shard and the config-4 shard)
    assert CpuBenchEnv.created and all(base == rank * n for (_, n, base) in CpuBenchEnv.created), CpuBenchEnv.created
    # same repeat count and same (max-reduced) time on every rank
    probe = torch.tensor([res["launches_per_bracket"], res["ms_per_step"]], dtype=torch.float64)
    both = [torch.empty_like(probe) for _ in range(world)]
    dist.all_gather(both, probe)
    assert torch.equal(both[0], both[1])
    if rank == 0:
        json.dumps(res)  # serialisable
        assert res["n_gpus"] == world and res["steps"] == 6 and res["scaling"] == "weak"
        assert abs(res["value"] - 64 * world * 6 / (res["ms_per_step"] * 1e-3 * 6)) < 1e-6 * res["value"]
        assert res["timed_ms_per_bracket"] >= 0.9 * bench.MIN_TIMED_MS
        ex = res["exchange"]   # obs f32 + reward f32 + done u8, gathered into [world][K][n][...] receive buffers
        assert ex["gathered_shape"] == [world, 6, 64, 17] and ex["bytes_per_rank"] == 6 * 64 * (17 * 4 + 4 + 1)
        assert ex["gathered_bytes"] == world * ex["bytes_per_rank"]
        c4 = res["config4"]
        assert c4["envs_total"] == 64 * world and c4["exchange"]["gathered_shape"][:2] == [world, c4["steps"]]
        # the line says what the collectives backend saw: world size from the process group and every rank's own time
        rc = res["rccl"]
        assert rc["rccl_world_size"] == world and rc["backend"] == "gloo" and len(rc["per_rank_ms_per_step"]) == world
        assert max(rc["per_rank_ms_per_step"]) <= res["ms_per_step"] * 1.5 and min(rc["per_rank_ms_per_step"]) > 0
        # the compact headline: one line under 4 KB with the contract's fields and a flat roofline object
        res["_full_path"] = None
        line = json.dumps(bench.headline(res))
        assert len(line) < 4096
        h = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in h, k
        assert all(not isinstance(v, (dict, list)) for v in h["roofline"].values())
        for k in ("frac", "frac_on_8d_bytes", "per_step_frac", "per_step_kernel_us", "valu_frac", "traffic_source"):
            assert k in h["roofline"], k
        assert h["rccl"]["rccl_world_size"] == world and "exchange_ms" in h["config4"]
        assert res["per_step_launch"]["roofline"]["bound"] == "hbm" and res["roofline"]["bound"] == "hbm" and res["roofline"]["frac"] < 1.0
        open(os.path.join(tmp, "ok"), "w").write("ok")
    rt.finish()


def test_bench_multirank_path_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def bench_standin_factory(variant, n, ga, env_id_base, seed=0, residual=None):
    """QR_BENCH_TEST_FACTORY target (bench.py's test hook): the CPU stand-in env of this file."""
    return CpuBenchEnv(variant, n, ga, env_id_base, seed, residual)


def _run_bench(argv, extra_env=None, timeout=600):
    import subprocess

    env = dict(os.environ, QR_BENCH_TEST_FACTORY="test_bench_gloo:bench_standin_factory",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_without_a_launcher_spawns_two_ranks():
    """`python bench.py --gpus 2` launched PLAINLY (no torchrun around it, WORLD_SIZE unset) must start its two
    ranks itself -- it used to run one rank and print n_gpus 1.  CPU ranks over gloo stand in for the GPUs."""
    r = _run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--envs", "64", "--repeats", "2", "--no-cpu-baseline", "--no-parity",
                    "--variant", "indi", "--no-extras"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl"]["rccl_world_size"] == 2 and line["rccl"]["backend"] == "gloo"
    assert len(line["rccl"]["per_rank_ms_per_step"]) == 2
    # every rank's own diagnosis row (kernel symbol, kernel time per step): a straggler must be visible in the line itself
    assert len(line["rccl"]["per_rank_kernel"]) == 2 and len(line["rccl"]["per_rank_kernel_us_per_step"]) == 2
    assert line["data"].startswith("cpu-standin")          # a stand-in run can never pass for a measurement
    assert abs(line["value"] - 64 * 2 * 6 / (line["ms_per_step"] * 1e-3 * 6)) < 1e-6 * line["value"]


def test_bench_refuses_a_world_size_other_than_gpus():
    """a launcher that started another number of ranks than --gpus says: every rank exits non-zero, no line is printed"""
    r = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--envs", "64", "--no-cpu-baseline", "--no-parity", "--no-extras"],
                   extra_env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and not r.stdout.strip()


def test_timed_region_fills_the_bracket():
    import time

    import bench

    rt = bench.Runtime(0, 0, 1, torch.device("cpu"), use_cuda=False)
    per, ts, R = bench.timed_region(rt, lambda: time.sleep(0.001), repeats=2, min_ms=20.0)
    assert R >= 10 and 0.0009 < per < 0.003 and len(ts) == 2
