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
    r, f = d["roofline"], d.get("fused_rollout")
    line = (f"{d['config'].get('variant','?'):5s} n_gpus={d['n_gpus']} step-launch: {d['ms_per_step']*1e3:6.2f} us/step "
            f"{d['value']/1e9:6.2f} G/s | kernel {r['kernel_us']:5.2f} us (events) frac {r['frac']:.3f}")
    if f:
        line += (f" | fused: {f['ms_per_step']*1e3:5.2f} us/step {f['value']/1e9:6.2f} G/s kernel {f['kernel_us_per_step']:5.2f} us"
                 f" frac(alg) {f['roofline_frac_algorithmic']:.3f}")
    if "parity" in d:
        line += f" | parity {d['parity'].get('max_rel_dstate_100_steps')}"
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        line += f" | cpu {c['value']/1e6:.2f} M/s on {c['cores']} cores"
    print(line)
