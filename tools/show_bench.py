#!/usr/bin/env python3
"""Pretty-print bench.py JSON lines: python tools/show_bench.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as ex:
        print(path, "unreadable:", ex)
        continue
    r, ps = d["roofline"], d.get("per_step_launch")
    line = (f"{d['config'].get('variant','?'):5s} n_gpus={d['n_gpus']} fused: {d['ms_per_step']*1e3:6.2f} us/step "
            f"{d['value']/1e9:6.2f} G/s | kernel {r['us_per_step']:5.2f} us/step frac(alg) {r['frac']:.3f}"
            f" frac(traffic) {r.get('frac_of_measured_traffic')}")
    if ps:
        pr = ps["roofline"]
        line += (f" | per-step: {ps['ms_per_step']*1e3:5.2f} us/step {ps['value']/1e9:6.2f} G/s kernel {pr['kernel_us']:5.2f} us"
                 f" frac {pr['frac']:.3f} traffic {pr['traffic']}")
    if "closed_loop" in d and "value" in d["closed_loop"]:
        c = d["closed_loop"]
        line += f" | closed-loop: {c['ms_per_step']*1e3:5.2f} us/step {c['value']/1e9:5.2f} G/s"
    if "parity" in d:
        line += f" | parity {d['parity'].get('max_rel_dstate_100_steps')}"
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        line += f" | cpu {c['value']/1e6:.2f} M/s on {c['cores']} cores"
    print(line)
