#!/bin/bash
# build an experiment variant of the library: tools/exp_build.sh <name> <extra flags...>
cd /root/repo; name=$1; shift
python - "$name" "$@" <<'PY'
import sys, os, subprocess
sys.path.insert(0, "/root/repo")
from optimal_quad_control_rl_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(B.PKG, "_dbg", "libexp_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
subprocess.check_call([B._hipcc(), *B.FLAGS, "-Wl,--version-script=" + os.path.join(B.CSRC, "exports.map"), *extra, "-o", out] + srcs)
print("built", out)
PY
