#!/usr/bin/env python3
"""A/B helper: run tools/bench_ppo_update.measure() against an experiment build (_dbg/libexp_<name>.so).  usage: exp_run.py <name|main> [L] [B]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
name = sys.argv[1]
if name != "main":
    B.LIB = os.path.join(B.PKG, "_dbg", "libexp_%s.so" % name)
    B.needs_build = lambda: False
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_ppo_update as U
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
Bn = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
r = U.measure(L=L, B=Bn, iters=100, with_torch=False)
print(name, "bf16", json.dumps({k: r[k] for k in r if k.endswith("_us") or k.startswith("status_")}))
