"""Import harness for the upstream notebooks (build container only).

Executes the hot-path code cells of the reference notebooks
('3D quad race.ipynb' cells 2,4,6 -> EoM, residual MLPs, Quadcopter3DGates;
'3D quad race INDI inner loop.ipynb' cells 2,5) inside a private namespace so
that golden vectors can be generated from the *real* reference.  Nothing from
the reference is written to disk by this module; only numeric input/output
vectors produced by tools/gen_golden.py are committed.

/root/reference does not exist on the GPU box, so nothing under tests/ (gpu
marker), bench.py or __graft_entry__.smoke() may import this file.
"""
import json
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("QUADRACE_REFERENCE", "/root/reference")
E2E_NB = "3D quad race.ipynb"
INDI_NB = "3D quad race INDI inner loop.ipynb"


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, E2E_NB))


class _Box:
    """Minimal stand-in for gymnasium.spaces.Box (not installed here)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.dtype = dtype


class _VecEnv:
    """Minimal stand-in for stable_baselines3.common.vec_env.VecEnv."""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


def _install_stubs():
    spaces = types.ModuleType("spaces")
    spaces.Box = _Box
    for name in ("gymnasium", "gym"):
        mod = types.ModuleType(name)
        mod.spaces = spaces
        sys.modules.setdefault(name, mod)
        sys.modules.setdefault(name + ".spaces", spaces)
    sb3 = types.ModuleType("stable_baselines3")
    sb3.__version__ = "stub"
    common = types.ModuleType("stable_baselines3.common")
    vec_env = types.ModuleType("stable_baselines3.common.vec_env")
    vec_env.VecEnv = _VecEnv
    sb3.common = common
    common.vec_env = vec_env
    sys.modules.setdefault("stable_baselines3", sb3)
    sys.modules.setdefault("stable_baselines3.common", common)
    sys.modules.setdefault("stable_baselines3.common.vec_env", vec_env)


def _cells(nb_name):
    with open(os.path.join(REF_ROOT, nb_name)) as f:
        nb = json.load(f)
    return nb["cells"]


def _exec_cells(nb_name, cell_ids, ns):
    cells = _cells(nb_name)
    for cid in cell_ids:
        src = "".join(cells[cid]["source"])
        src = "\n".join(l for l in src.split("\n") if not l.lstrip().startswith("%"))
        exec(compile(src, f"{nb_name}#cell{cid}", "exec"), ns)


def load_e2e():
    """Namespace with f_func, get_body_velocity, thrust/moment models, Quadcopter3DGates (E2E)."""
    import torch

    _install_stubs()
    ns = {"np": np, "torch": torch, "__name__": "ref_e2e"}
    cwd = os.getcwd()
    real_load = torch.load
    torch.load = lambda p, *a, **k: real_load(p, *a, weights_only=False, map_location="cpu", **k)
    os.chdir(REF_ROOT)
    try:
        import contextlib, io

        with contextlib.redirect_stdout(io.StringIO()):
            _exec_cells(E2E_NB, (2, 4, 6), ns)
    finally:
        os.chdir(cwd)
        torch.load = real_load
    return ns


def load_indi():
    """Namespace with f_func and Quadcopter3DGates (INDI inner-loop variant)."""
    import torch

    _install_stubs()
    ns = {"np": np, "torch": torch, "__name__": "ref_indi"}
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        import contextlib, io

        with contextlib.redirect_stdout(io.StringIO()):
            _exec_cells(INDI_NB, (2, 5), ns)
    finally:
        os.chdir(cwd)
    return ns


Q3_NB = "3D quad.ipynb"


def load_q3():
    """Namespace with the predecessor notebook's f_func, Quadcopter3DVec (hover) and Quadcopter3DVecGates."""
    import torch

    _install_stubs()
    ns = {"np": np, "torch": torch, "__name__": "ref_q3"}
    cwd = os.getcwd()
    os.chdir(REF_ROOT)
    try:
        import contextlib, io

        with contextlib.redirect_stdout(io.StringIO()):
            _exec_cells(Q3_NB, (2, 6, 14), ns)
    finally:
        os.chdir(cwd)
    return ns
