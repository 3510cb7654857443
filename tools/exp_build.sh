#!/bin/bash
# build an experiment variant of the library: tools/exp_build.sh <name> <extra flags...>   -> optimal_quad_control_rl_amd/_dbg/libexp_<name>.so
# (the product's own pipeline: compile through assembly, hazardous packed-f32 forms rewritten, final code objects linted)
cd /root/repo; name=$1; shift
python - "$name" "$@" <<'PY'
import sys, os
sys.path.insert(0, "/root/repo")
from optimal_quad_control_rl_amd import build as B
name, extra = sys.argv[1], tuple(sys.argv[2:])
out = os.path.join(B.PKG, "_dbg", "libexp_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
print("built", B.build_native(force=False, extra_flags=extra, out=out))
PY
