"""ctypes binding of the TEST-ONLY CPU oracle for the predecessor environments (oracle/quad3d_oracle.c).

Only tests/ and bench.py's cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C

import numpy as np

from . import oracle as _o

HOVER, GATES = 0, 1

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_ready = False


def lib():
    global _ready
    L = _o.lib()
    if not _ready:
        L.q3o_create.restype = C.c_void_p
        L.q3o_create.argtypes = [C.c_int, C.c_int, C.c_uint64]
        L.q3o_destroy.argtypes = [C.c_void_p]
        L.q3o_set_track.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, _f32p]
        L.q3o_set_limits.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.q3o_set_thresholds.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.q3o_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.q3o_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.q3o_reset.argtypes = [C.c_void_p, _u8p]
        L.q3o_step.argtypes = [C.c_void_p, _f32p, C.c_void_p, C.c_void_p, _u8p, _u8p]
        L.q3o_states.restype = C.c_void_p
        L.q3o_states.argtypes = [C.c_void_p]
        for name, rt in (("q3o_target", C.POINTER(C.c_int32)), ("q3o_steps", C.POINTER(C.c_int32)),
                         ("q3o_episode", C.POINTER(C.c_uint32))):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.q3o_f_f32.argtypes = [_f32p, _f32p, _f32p]
        L.q3o_f_f64.argtypes = [_f64p, _f32p, _f64p]
        L.q3o_normal_pair.argtypes = [C.c_uint32, C.c_uint32, _f32p, _f32p]
        _ready = True
    return L


def f_func(states, actions):
    """Q3 cell 2 f_func row by row; dtype of `states` (float32 / float64) selects the variant."""
    L = lib()
    s = np.ascontiguousarray(states)
    u = np.ascontiguousarray(actions, dtype=np.float32)
    out = np.empty_like(s)
    fn, ptr = (L.q3o_f_f64, _f64p) if s.dtype == np.float64 else (L.q3o_f_f32, _f32p)
    for i in range(s.shape[0]):
        fn(s[i].ctypes.data_as(ptr), u[i].ctypes.data_as(_f32p), out[i].ctypes.data_as(ptr))
    return out


class Quad3DOracle:
    """kind HOVER: Quadcopter3DVec (float64 states/rewards); kind GATES: Quadcopter3DVecGates (float32)."""

    def __init__(self, kind, num_envs, gate_pos=None, gate_yaw=None, start_pos=None, env_id_base=0, seed=0):
        self.L = lib()
        self.kind, self.n = kind, num_envs
        self.dtype = np.float64 if kind == HOVER else np.float32
        self.h = self.L.q3o_create(kind, num_envs, env_id_base)
        if kind == GATES:
            gp = np.ascontiguousarray(gate_pos, dtype=np.float32)
            gy = np.ascontiguousarray(gate_yaw, dtype=np.float32)
            sp = np.ascontiguousarray(start_pos, dtype=np.float32)
            self.num_gates = gp.shape[0]
            self.L.q3o_set_track(self.h, gp.ctypes.data_as(_f32p), gy.ctypes.data_as(_f32p), gp.shape[0],
                                 sp.ctypes.data_as(_f32p))
        self.L.q3o_seed(self.h, seed)
        n = num_envs
        sp_ = C.cast(self.L.q3o_states(self.h), C.POINTER(C.c_double if kind == HOVER else C.c_float))
        self.states = np.ctypeslib.as_array(sp_, shape=(n, 16))
        self.target = np.ctypeslib.as_array(self.L.q3o_target(self.h), shape=(n,))
        self.steps = np.ctypeslib.as_array(self.L.q3o_steps(self.h), shape=(n,))
        self.episode = np.ctypeslib.as_array(self.L.q3o_episode(self.h), shape=(n,))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.q3o_destroy(self.h)
            self.h = None

    def seed(self, seed):
        self.L.q3o_seed(self.h, seed)

    def set_limits(self, max_steps, dt=0.01):
        self.L.q3o_set_limits(self.h, max_steps, dt)

    def set_threads(self, t):
        self.L.q3o_set_threads(self.h, t)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).ctypes.data_as(_u8p)
        self.L.q3o_reset(self.h, m)
        return self.states.copy()

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        rew = np.empty(self.n, self.dtype)
        done = np.empty(self.n, np.uint8)
        trunc = np.empty(self.n, np.uint8)
        self.L.q3o_step(self.h, a.ctypes.data_as(_f32p), None, rew.ctypes.data_as(C.c_void_p), done.ctypes.data_as(_u8p),
                        trunc.ctypes.data_as(_u8p))
        return self.states.copy(), rew, done.astype(bool), trunc.astype(bool)
