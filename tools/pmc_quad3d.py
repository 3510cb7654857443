#!/usr/bin/env python3
"""HBM traffic of the predecessor-env step kernels (include/quad3d.h) from the PMC counters, 1 Mi envs.
  probe (under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes):   python tools/pmc_quad3d.py probe
  summary:   python tools/pmc_quad3d.py summarise FETCH.csv WRITE.csv out.json"""
import csv, json, os, sys
csv.field_size_limit(1 << 30)
CAL, N = 256 * 1024 * 1024, 1 << 20
ALG = {"hover": 417, "gates": 229}   # algorithmic bytes per env-step (tools/bench_quad3d.py)
if sys.argv[1] == "probe":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_quad3d import Q3_TRACK
    from optimal_quad_control_rl_amd.quad3d import Quadcopter3DVec, Quadcopter3DVecGates
    src = torch.empty(CAL // 4, device="cuda"); dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    for env in (Quadcopter3DVec(N), Quadcopter3DVecGates(N, *Q3_TRACK)):
        env.reset_device()
        a = torch.rand((N, 4), device="cuda") * 2 - 1
        for _ in range(12):
            env.step_device(a)
        torch.cuda.synchronize()
else:
    def load(path, counter):
        return [(r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]

    def per_kernel(rows):
        cal = [v for k, v in rows if "copy" in k.lower() or "elementwise" in k.lower()]
        cal = [v for v in cal if v > 0.9 * max(cal)]
        out = {"cal": sum(cal) / len(cal)}
        for name, key in (("hover", "q3_step_kernelId"), ("gates", "q3_step_kernelIf")):
            v = [x for k, x in rows if key in k or ("q3_step_kernel<" + ("double" if name == "hover" else "float")) in k][2:]
            out[name] = sum(v) / max(1, len(v))
        return out
    f = per_kernel(load(sys.argv[2], "FETCH_SIZE")); w = per_kernel(load(sys.argv[3], "WRITE_SIZE"))
    fs, ws = CAL / (f["cal"] * 1024), CAL / (w["cal"] * 1024)
    res = {"envs": N, "fetch_scale": fs, "write_scale": ws}
    for k in ("hover", "gates"):
        rd, wr = f[k] * 1024 * fs, w[k] * 1024 * ws
        res[k] = {"read_MB": rd / 1e6, "write_MB": wr / 1e6, "bytes_per_env_step": (rd + wr) / N, "algorithmic_bytes_per_env_step": ALG[k]}
    json.dump(res, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(res, indent=1))
