#!/usr/bin/env python3
"""Table of the arithmetic A/B on the reference's recipe (tools/reference_recipe_run.py runs, profiles/r04_ab_{f16,f32}_seed*.json):
per run the env-steps at which the reference's level (flying lap <= 2.6 s, <= 0.1 crashes per 12 s) is first reached, the flying lap at
the end, the best evaluation, and how many evaluations flew no lap at all.   python tools/ab_summary.py [dir]"""
import glob, json, os, sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
for arith in ("f16", "f32"):
    rows = []
    for f in sorted(glob.glob(os.path.join(d, f"r04_ab_{arith}_seed*.json"))):
        r = json.load(open(f))
        c = r["curve"]
        laps = [x["flying_lap"] for x in c if x["flying_lap"] is not None]
        rows.append((r["seed"], r["reaches_reference_level_after_steps"], r["final"]["flying_lap"], r["final"]["crashes_per_12s"],
                     min(laps) if laps else None, sum(1 for x in c if x["flying_lap"] is None), len(c), r["train_seconds"]))
    if not rows:
        continue
    print(f"## {arith}: {len(rows)} runs")
    print("| seed | reference level after (env-steps) | lap at 6e7 steps (s), crashes / 12 s | best evaluation (s) | evaluations without a lap | train s |")
    print("|---|---|---|---|---|---|")
    for s, st, fl, cr, best, nolap, n, ts in rows:
        print(f"| {s} | {'never' if st is None else '%.1e' % st} | {'no lap' if fl is None else '%.2f' % fl}, {cr:.2f} | {'-' if best is None else '%.2f' % best} | {nolap} of {n} | {ts:.0f} |")
    reached = [r[1] for r in rows if r[1] is not None]
    fin = sorted(r[2] for r in rows if r[2] is not None)
    print(f"reached: {len(reached)} of {len(rows)}; median steps to reach {sorted(reached)[len(reached)//2] if reached else None}; "
          f"final laps median {fin[len(fin)//2] if fin else None}; runs ending <= 2.6 s: {sum(1 for x in fin if x <= 2.6)}; <= 2.9 s: {sum(1 for x in fin if x <= 2.9)}")
