#!/usr/bin/env python3
"""Run a script against an experiment build: tools/with_lib.py <name> <script.py> [args...]   (library _dbg/libexp_<name>.so)"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
B.LIB = os.path.join(B.PKG, "_dbg", "libexp_%s.so" % sys.argv[1])
B.needs_build = lambda: False
script = sys.argv[2]
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
