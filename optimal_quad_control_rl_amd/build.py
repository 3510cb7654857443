"""Builds libquadrace.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m optimal_quad_control_rl_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so that it travels with the
repository snapshot to the GPU box (it is git-ignored, never pip-installed).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libquadrace.so")
SOURCES = ["quadrace_kernels.hip", "quadrace_abi.hip", "quadrace_policy.hip"]
HEADERS = ["quadrace_device.hpp", "quadrace_policy.hpp", os.path.join("..", "..", "include", "quadrace.h")]
# -ffp-contract=off: FMAs are written explicitly (fmaf) in the kernels, so the arithmetic is fixed by the source and
# the per-step kernel and the fused rollout kernel produce bit-identical trajectories.
# -amdgpu-mfma-vgpr-form: keep the MFMA accumulators of the residual MLP in VGPRs (gfx950 has a unified register file),
# so the VALU epilogue reads them directly instead of through 64 v_accvgpr_read per step.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-result"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libquadrace.so cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), *FLAGS, *extra_flags, "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv, verbose=True)
    print("built", LIB)
