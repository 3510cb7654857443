#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md
prescribes: they do not fit one pass) into HBM bytes per launch / per step for the env kernels, KEYED BY KERNEL SYMBOL
(round 4: bench.py looks the evidence up under the symbol the library reports for its launch).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d D/fetch -o f -- python tools/pmc_probe.py e2e [n]
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d D/write -o w -- python tools/pmc_probe.py e2e [n]
    python tools/pmc_traffic.py e2e 65536 D/fetch/f_counter_collection.csv D/write/w_counter_collection.csv out.json

Units / corrections: both counters are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (guide, section HBM) -- confirmed here by the calibration copy in tools/pmc_probe.py
(256 MiB copied: FETCH_SIZE = 131 088 KiB) -- so reads are scaled by the calibration factor measured in the same
run; WRITE_SIZE is calibrated the same way against the 256 MiB the copy writes.
"""
import csv
import json
import os
import re
import subprocess
import sys

csv.field_size_limit(1 << 30)
CAL_BYTES = 256 * 1024 * 1024
K_FUSED = 64  # steps per rollout launch in tools/pmc_probe.py
SYM = re.compile(r"(qr::[a-z_0-9]+<\d+, \d+>)")


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def per_symbol(rows):
    cal = [v for k, v in rows if "copyBuffer" in k]
    cal = [v for v in cal if v > 0.9 * max(cal)]  # only the 256 MiB calibration copies, not small H2D uploads
    acc = {}
    for k, v in rows:
        m = SYM.search(k)
        if m:
            acc.setdefault(m.group(1), []).append(v)
    return sum(cal) / len(cal), acc


def commit_id():
    """the commit the measured library was built from: QR_COMMIT (the GPU box has no .git) or git itself"""
    if os.environ.get("QR_COMMIT"):
        return os.environ["QR_COMMIT"]
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)),
                                       stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return "unknown"


def main(variant, n, fetch_csv, write_csv, out="profiles/r06_pmc_traffic.json"):
    n = int(n)
    f_cal, f = per_symbol(load(fetch_csv, "FETCH_SIZE"))
    w_cal, w = per_symbol(load(write_csv, "WRITE_SIZE"))
    f_scale = CAL_BYTES / (f_cal * 1024.0)  # bytes per reported KiB*1024 (expected 2.0 on gfx950)
    w_scale = CAL_BYTES / (w_cal * 1024.0)
    allres = json.load(open(out)) if os.path.exists(out) else {}
    allres["commit"] = commit_id()
    allres["counters"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB, scaled by the 256 MiB calibration copy of the "
                          "same run; per kernel symbol: mean over its launches (per-step kernel: after the first four)")
    sec = allres.setdefault(f"n{n}", {})
    sec[f"calibration_{variant}"] = {"copy_bytes": CAL_BYTES, "FETCH_SIZE_KiB": f_cal, "WRITE_SIZE_KiB": w_cal, "fetch_scale": f_scale,
                                     "write_scale": w_scale}
    for sym in sorted(set(f) | set(w)):
        fv, wv = f.get(sym, [0.0]), w.get(sym, [0.0])
        if "step_kernel" in sym:
            fv, wv = fv[4:] or fv, wv[4:] or wv
        rb, wb = sum(fv) / len(fv) * 1024 * f_scale, sum(wv) / len(wv) * 1024 * w_scale
        e = {"launches": len(fv), "read_bytes_per_launch": rb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": rb + wb}
        if "rollout" in sym:
            e.update(steps_per_launch=K_FUSED, hbm_bytes_per_step=(rb + wb) / K_FUSED, hbm_bytes_per_env_step=(rb + wb) / K_FUSED / n)
        else:
            e.update(hbm_bytes_per_env_step=(rb + wb) / n)
        sec[sym] = e
    json.dump(allres, open(out, "w"), indent=1)
    print(json.dumps({k: round(v.get("hbm_bytes_per_env_step", 0), 2) for k, v in sec.items() if k.startswith("qr::")}))


if __name__ == "__main__":
    main(*sys.argv[1:])
