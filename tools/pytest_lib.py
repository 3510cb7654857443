#!/usr/bin/env python3
"""Run pytest against an experiment build: tools/pytest_lib.py <name> <pytest args...>  (library _dbg/libexp_<name>.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optimal_quad_control_rl_amd import build as B
B.LIB = os.path.join(B.PKG, "_dbg", "libexp_%s.so" % sys.argv[1])
B.needs_build = lambda: False
import pytest
sys.exit(pytest.main(sys.argv[2:]))
