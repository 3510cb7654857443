#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md
prescribes: they do not fit one pass) into HBM bytes per launch for the env kernels.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D/fetch -o f -- python tools/pmc_probe.py e2e
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D/write -o w -- python tools/pmc_probe.py e2e
    python tools/pmc_traffic.py e2e 65536 D/fetch/f_counter_collection.csv D/write/w_counter_collection.csv

Units / corrections: both counters are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (guide, section HBM) -- confirmed here by the calibration copy in tools/pmc_probe.py
(256 MiB copied: FETCH_SIZE = 131 088 KiB) -- so reads are scaled by the calibration factor measured in the same
run; WRITE_SIZE is calibrated the same way against the 256 MiB the copy writes.
"""
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)
CAL_BYTES = 256 * 1024 * 1024
K_FUSED = 64  # steps per rollout launch in tools/pmc_probe.py


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def summarise(rows):
    cal = [v for k, v in rows if "copyBuffer" in k]
    cal = [v for v in cal if v > 0.9 * max(cal)]  # only the 256 MiB calibration copies, not small H2D uploads
    step = [v for k, v in rows if "step_kernel" in k]
    roll = [v for k, v in rows if "rollout_kernel" in k or "rollout_stash_kernel" in k]
    rpol = [v for k, v in rows if "rollout_policy_kernel" in k]
    return (sum(cal) / len(cal), sum(step[4:]) / max(1, len(step[4:])), sum(roll) / max(1, len(roll)),
            sum(rpol) / max(1, len(rpol)))


def main(variant, n, fetch_csv, write_csv, out="profiles/pmc_summary.json"):
    n = int(n)
    f_cal, f_step, f_roll, f_rpol = summarise(load(fetch_csv, "FETCH_SIZE"))
    w_cal, w_step, w_roll, w_rpol = summarise(load(write_csv, "WRITE_SIZE"))
    f_scale = CAL_BYTES / (f_cal * 1024.0)  # bytes per reported KiB*1024 (expected 2.0 on gfx950)
    w_scale = CAL_BYTES / (w_cal * 1024.0)
    res = {
        "counters": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB",
        "calibration": {"copy_bytes": CAL_BYTES, "FETCH_SIZE_KiB": f_cal, "WRITE_SIZE_KiB": w_cal,
                        "fetch_scale": f_scale, "write_scale": w_scale},
        "step_kernel": {"FETCH_SIZE_KiB": f_step, "WRITE_SIZE_KiB": w_step,
                        "read_bytes": f_step * 1024 * f_scale, "write_bytes": w_step * 1024 * w_scale},
        "rollout_kernel_per_step": {"FETCH_SIZE_KiB": f_roll / K_FUSED, "WRITE_SIZE_KiB": w_roll / K_FUSED,
                                    "read_bytes": f_roll * 1024 * f_scale / K_FUSED,
                                    "write_bytes": w_roll * 1024 * w_scale / K_FUSED},
    }
    res["rollout_policy_kernel_per_step"] = {"read_bytes": f_rpol * 1024 * f_scale / K_FUSED, "write_bytes": w_rpol * 1024 * w_scale / K_FUSED}
    res["closed_loop_hbm_bytes_per_step"] = (res["rollout_policy_kernel_per_step"]["read_bytes"] +
                                             res["rollout_policy_kernel_per_step"]["write_bytes"])
    res["hbm_bytes_per_launch"] = res["step_kernel"]["read_bytes"] + res["step_kernel"]["write_bytes"]
    res["fused_hbm_bytes_per_step"] = (res["rollout_kernel_per_step"]["read_bytes"] +
                                       res["rollout_kernel_per_step"]["write_bytes"])
    allres = {}
    if os.path.exists(out):
        allres = json.load(open(out))
    allres[f"{variant}_n{n}_ga1"] = res
    json.dump(allres, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("hbm_bytes_per_launch", "fused_hbm_bytes_per_step", "calibration")}))


if __name__ == "__main__":
    main(*sys.argv[1:])
