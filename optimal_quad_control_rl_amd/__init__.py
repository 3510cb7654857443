"""optimal_quad_control_rl_amd -- MI355X-native vectorised quadrotor race environment.

Drop-in for the `Quadcopter3DGates` VecEnv classes of tudelft/optimal_quad_control_RL (E2E motor-command
model with residual thrust/moment MLPs, and the INDI inner-loop variant).  The hot path is hand-written HIP
for gfx950 behind a C ABI (include/quadrace.h -> libquadrace.so); this package is the thin ctypes adapter.
"""
from .tracks import TRAIN_DISTURBANCE_RANGES, square_track, zigzag_track  # noqa: F401

__all__ = ["Quadcopter3DGates", "Quadcopter3DGatesINDI", "zigzag_track", "square_track", "TRAIN_DISTURBANCE_RANGES",
           "default_residual_blob", "ShardedRaceEnv", "Quadcopter3DVec", "Quadcopter3DVecGates", "PPO", "VecMonitor"]


def __getattr__(name):  # lazy: importing the package must not require torch / a GPU
    if name in ("Quadcopter3DGates", "Quadcopter3DGatesINDI", "default_residual_blob", "Box"):
        from . import vec_env

        return getattr(vec_env, name)
    if name in ("Quadcopter3DVec", "Quadcopter3DVecGates"):  # predecessor envs of "3D quad.ipynb"
        from . import quad3d

        return getattr(quad3d, name)
    if name in ("PPO", "VecMonitor"):  # SB3-shaped model object around the on-device PPO (R:783-831, R:3985-3996)
        from . import sb3

        return getattr(sb3, name)
    if name == "ShardedRaceEnv":
        from . import sharded

        return sharded.ShardedRaceEnv
    raise AttributeError(name)
