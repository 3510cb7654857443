#!/usr/bin/env python3
"""HBM traffic of the PPO update kernels from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, calibrated on a
256 MiB copy in the same run, like tools/pmc_traffic.py).

  probe (under rocprofv3):   python tools/pmc_ppo.py probe
  summary:                   python tools/pmc_ppo.py summarise FETCH.csv WRITE.csv out.json
"""
import csv, json, os, sys
csv.field_size_limit(1 << 30)
CAL = 256 * 1024 * 1024
ITERS = 20

if sys.argv[1] == "probe":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater
    dev = torch.device("cuda", 0)
    src = torch.empty(CAL // 4, device=dev); dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)                      # calibration copies (256 MiB read + 256 MiB written each)
    L, B, R = int(os.environ.get("QR_PMC_OBS_LEN", "24")), 16384, 65536 * 8
    obs = torch.randn((R, L), device=dev); act = torch.randn((R, 4), device=dev) * 0.5
    old_lp = torch.randn(R, device=dev) * 0.1 - 3.0; adv = torch.randn(R, device=dev); ret = torch.randn(R, device=dev)
    perm = torch.randperm(R, device=dev).to(torch.int32)
    up = MfmaPpoUpdater(ActorCritic(L, 4).to(dev), L, dev, B)
    up.begin_epoch(adv, perm[:ITERS * B].contiguous(), B)   # one adv-statistics launch for the ITERS minibatches
    for k in range(ITERS):
        up.minibatch(obs, act, old_lp, adv, ret, perm[k * B:(k + 1) * B], 3e-4)
    torch.cuda.synchronize()
else:
    def load(path, counter):
        rows = []
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
        return rows

    def per_kernel(rows):
        cal = [v for k, v in rows if "copy" in k.lower() or "elementwise" in k.lower()]
        cal = [v for v in cal if v > 0.9 * max(cal)]
        out = {"cal": sum(cal) / len(cal)}
        for name in ("ppo_adv_stats", "ppo_grad", "ppo_phase_a", "ppo_phase_b", "ppo_apply"):   # ppo_grad: the fused form (default);
            v = [x for k, x in rows if name in k][4:]                                            # (the removed two-kernel form)
            if v:
                out[name] = sum(v) / len(v)
        return out

    f = per_kernel(load(sys.argv[2], "FETCH_SIZE")); w = per_kernel(load(sys.argv[3], "WRITE_SIZE"))
    fs, ws = CAL / (f["cal"] * 1024), CAL / (w["cal"] * 1024)
    res = {"counters": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB; scales from a 256 MiB copy in the same run",
           "fetch_scale": fs, "write_scale": ws, "minibatch": 16384, "obs_len": int(os.environ.get("QR_PMC_OBS_LEN", "24")), "commit": os.environ.get("QR_COMMIT"),
           "per_launch_MB": {}}
    for name in f:
        if name != "cal":
            res["per_launch_MB"][name] = {"read": f[name] * 1024 * fs / 1e6, "write": w[name] * 1024 * ws / 1e6}
    json.dump(res, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(res["per_launch_MB"], indent=1))
