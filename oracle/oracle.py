"""ctypes binding of the TEST-ONLY CPU oracle (oracle/quadrace_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (optimal_quad_control_rl_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# QR_ORACLE_LIB selects another build of the same sources (the sanitizer build of `make -C oracle asan-test`)
_LIB_PATH = os.environ.get("QR_ORACLE_LIB") or os.path.join(_HERE, "libquadrace_oracle.so")
_REF_LIB_PATH = os.path.join(_HERE, "_ref", "libref_residual.so")

E2E, INDI = 0, 1

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is mounted)."""
    srcs = [os.path.join(_HERE, f) for f in ("quadrace_oracle.c", "quad3d_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    if force or not os.path.exists(_REF_LIB_PATH):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.qro_create.restype = C.c_void_p
        L.qro_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
        L.qro_destroy.argtypes = [C.c_void_p]
        L.qro_set_track.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, _f32p]
        L.qro_set_residual.argtypes = [C.c_void_p, _f32p]
        L.qro_set_disturbance.argtypes = [C.c_void_p, _f32p, C.c_float]
        L.qro_set_limits.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.qro_set_pause.argtypes = [C.c_void_p, C.c_int]
        L.qro_set_terminal_obs.argtypes = [C.c_void_p, _f32p]
        L.qro_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.qro_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.qro_reset.argtypes = [C.c_void_p, _u8p, _f32p]
        L.qro_step.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _u8p, _u8p]
        L.qro_observe.argtypes = [C.c_void_p, _f32p]
        L.qro_get_track_tables.argtypes = [C.c_void_p, _f32p, _f32p]
        for name, rt in (("qro_world", _f32p), ("qro_dist", _f32p), ("qro_obs", _f32p), ("qro_target", _i32p),
                         ("qro_steps", _i32p), ("qro_episode", _u32p)):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.qro_state_len.argtypes = [C.c_void_p]
        L.qro_obs_len.argtypes = [C.c_void_p]
        L.qro_f_e2e.argtypes = [_f32p, _f32p, _f32p, _f32p]
        L.qro_f_indi.argtypes = [_f32p, _f32p, _f32p]
        L.qro_body_velocity.argtypes = [_f32p, _f32p]
        L.qro_residual.argtypes = [_f32p, _f32p, _f32p, _f32p]
        L.qro_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        _lib = L
    return _lib


def ref_residual_lib():
    """The reference's own generated C (c_code/nn_thrust.c, nn_moment.c) compiled by oracle/Makefile; None if absent."""
    build()
    if not os.path.exists(_REF_LIB_PATH):
        return None
    L = C.CDLL(_REF_LIB_PATH)
    L.nn_thrust_forward.argtypes = [_f32p, _f32p]
    L.nn_moment_forward.argtypes = [_f32p, _f32p]
    return L


def ref_policy_lib():
    """The reference's generated policy C (c_code/neural_network.c: nn_forward + baked weights); None if absent."""
    build()
    path = os.path.join(_HERE, "_ref", "libref_policy.so")
    if not os.path.exists(path):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.nn_forward.argtypes = [_f32p, _f32p]
    return L


def _p(a, t=_f32p):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- free functions -------------------------------------------------------------------------------
def f_e2e(state, control, dist):
    state, control, dist = _f32(state), _f32(control), _f32(dist)
    out = np.empty_like(state)
    L = lib()
    for i in range(state.shape[0]):
        L.qro_f_e2e(_p(state[i]), _p(control[i]), _p(dist[i]), _p(out[i]))
    return out


def f_indi(state, control):
    state, control = _f32(state), _f32(control)
    out = np.empty_like(state)
    L = lib()
    for i in range(state.shape[0]):
        L.qro_f_indi(_p(state[i]), _p(control[i]), _p(out[i]))
    return out


def body_velocity(state):
    state = _f32(state)
    out = np.empty((state.shape[0], 3), np.float32)
    L = lib()
    for i in range(state.shape[0]):
        L.qro_body_velocity(_p(state[i]), _p(out[i]))
    return out


def residual(blob, state):
    blob, state = _f32(blob), _f32(state)
    thrust = np.empty((state.shape[0], 1), np.float32)
    moment = np.empty((state.shape[0], 3), np.float32)
    L = lib()
    for i in range(state.shape[0]):
        L.qro_residual(_p(blob), _p(state[i]), _p(thrust[i]), _p(moment[i]))
    return thrust, moment


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32)
    key = np.ascontiguousarray(key, np.uint32)
    out = np.empty(4, np.uint32)
    lib().qro_philox4x32_10(_p(ctr, _u32p), _p(key, _u32p), _p(out, _u32p))
    return out


# ---- environment ----------------------------------------------------------------------------------
class OracleEnv:
    """CPU restatement of Quadcopter3DGates (R:287-620 / I:142-410) with NumPy views on its state."""

    def __init__(self, variant, num_envs, gate_pos, gate_yaw, start_pos, gates_ahead=0, pause_if_collision=False,
                 env_id_base=0):
        self.L = lib()
        self.variant, self.num_envs = variant, int(num_envs)
        self.h = C.c_void_p(self.L.qro_create(variant, self.num_envs, gates_ahead, int(pause_if_collision),
                                              env_id_base))
        gp, gy, sp = _f32(gate_pos), _f32(gate_yaw), _f32(start_pos)
        self.num_gates = gp.shape[0]
        self.L.qro_set_track(self.h, _p(gp), _p(gy), self.num_gates, _p(sp))
        self.state_len = self.L.qro_state_len(self.h)
        self.obs_len = self.L.qro_obs_len(self.h)
        n = self.num_envs
        as_np = np.ctypeslib.as_array
        self.world_states = as_np(self.L.qro_world(self.h), (n, self.state_len))
        self.disturbances = as_np(self.L.qro_dist(self.h), (n, 6))
        self.states = as_np(self.L.qro_obs(self.h), (n, self.obs_len))
        self.target_gates = as_np(self.L.qro_target(self.h), (n,))
        self.step_counts = as_np(self.L.qro_steps(self.h), (n,))
        self.episode = as_np(self.L.qro_episode(self.h), (n,))

    def __del__(self):
        try:
            self.L.qro_destroy(self.h)
        except Exception:
            pass

    def track_tables(self):
        pr = np.empty((self.num_gates, 3), np.float32)
        yr = np.empty(self.num_gates, np.float32)
        self.L.qro_get_track_tables(self.h, _p(pr), _p(yr))
        return pr, yr

    def set_residual(self, blob):
        self.L.qro_set_residual(self.h, None if blob is None else _p(_f32(blob)))

    def set_disturbance(self, ranges, scale=1.0):
        self.L.qro_set_disturbance(self.h, _p(_f32(ranges)), float(scale))

    def set_limits(self, max_steps=1200, dt=0.01):
        self.L.qro_set_limits(self.h, int(max_steps), float(dt))

    def set_pause(self, pause):
        self.L.qro_set_pause(self.h, int(bool(pause)))

    def set_terminal_obs(self, buf):
        """buf: float32 [num_envs, obs_len] array (kept alive here) receiving the pre-reset observation of every env that
        finishes at a step (twin of qr_set_terminal_obs), or None."""
        self._term_obs = None if buf is None else buf
        assert buf is None or (buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"])
        self.L.qro_set_terminal_obs(self.h, None if buf is None else _p(buf))

    def set_threads(self, threads):
        """cpu_baseline only: spread the independent envs over OpenMP threads (results are unchanged)."""
        return self.L.qro_set_threads(self.h, int(threads))

    def seed(self, seed):
        self.L.qro_seed(self.h, int(seed))

    def reset(self, mask=None):
        m = None if mask is None else _p(np.ascontiguousarray(mask, np.uint8), _u8p)
        self.L.qro_reset(self.h, m, None)
        return self.states.copy()

    def observe(self):
        self.L.qro_observe(self.h, None)
        return self.states.copy()

    def step(self, actions):
        a = _f32(actions)
        n = self.num_envs
        rew = np.empty(n, np.float32)
        done = np.empty(n, np.uint8)
        trunc = np.empty(n, np.uint8)
        self.L.qro_step(self.h, _p(a), None, _p(rew), _p(done, _u8p), _p(trunc, _u8p))
        return self.states.copy(), rew, done.astype(bool), trunc.astype(bool)
