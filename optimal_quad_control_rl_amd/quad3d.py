"""Host-side mirrors of the predecessor environments of the reference's "3D quad.ipynb" (SURVEY.md 8(f) #4):

    Quadcopter3DVec(num_envs)                                   hover task            Q3 cell 6   (float64 states)
    Quadcopter3DVecGates(num_envs, gates_pos, gate_yaw, start_pos)   gate sequence    Q3 cell 14  (float32 states)

Same constructors, methods and public attributes (`states`, `step_counts`, `target_gates`, `max_steps`, `dt`, the hover
thresholds, `render()`), on top of include/quad3d.h in libquadrace.so: one HIP kernel per step over all envs, plus a
device-tensor path (`reset_device`, `step_device`, `rollout_device`) without host round trips.  No NumPy fallback.

Differences that are this build's own (documented, not the reference's): resets draw from a counter-based Philox
stream (`seed`, `env_id_base`) instead of NumPy's global generator; the per-step `print()` calls of the reference
are not reproduced; `infos` reproduces the reference's shared-dict behaviour (`infos = [{}] * N`).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .vec_env import _Base, _f32p, _make_box, _ptr

Q3_KIND_HOVER, Q3_KIND_GATES = 0, 1
_KEYS = ['x', 'y', 'z', 'vx', 'vy', 'vz', 'phi', 'theta', 'psi', 'p', 'q', 'r', 'w1', 'w2', 'w3', 'w4']


class _Quad3DBase(_Base):
    KIND = None
    DTYPE = None        # torch dtype of states / rewards
    NP_DTYPE = None

    def _create(self, num_envs, device, seed, env_id_base):
        self._h = None
        self._L = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError(f"{type(self).__name__} needs a gfx950 GPU: libquadrace has no CPU fallback")
        self._dev_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self._dev_index)
        h = C.c_void_p()
        _lib.check(self._L.q3_create(self.KIND, int(num_envs), self._dev_index, int(env_id_base), C.byref(h)))
        self._h = h
        # action space and observation space exactly as written upstream (low = -inf, high = -inf for the first 12!)
        action_space = _make_box(-1, 1, shape=(4,))
        try:
            observation_space = _make_box(np.array([-np.inf] * 12 + [-1] * 4), np.array([-np.inf] * 12 + [1] * 4))
        except Exception:  # a Box implementation that insists on low <= high
            observation_space = _make_box(np.array([-np.inf] * 12 + [-1] * 4), np.array([np.inf] * 12 + [1] * 4))
        _Base.__init__(self, int(num_envs), observation_space, action_space)
        n = self.num_envs
        self._max_steps, self._dt = 1000, 0.01
        self.actions = np.zeros((n, 4), dtype=np.float32)
        self._states_d = torch.zeros((n, 16), dtype=self.DTYPE, device=self.device)
        self._rew_d = torch.zeros(n, dtype=self.DTYPE, device=self.device)
        self._done_d = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._trunc_d = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._act_d = torch.zeros((n, 4), dtype=torch.float32, device=self.device)
        self._act_h = torch.zeros((n, 4), dtype=torch.float32).pin_memory()
        _lib.check(self._L.q3_seed(self._h, int(seed)))

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            self._L.q3_destroy(self._h)
            self._h = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- attributes of the reference classes -------------------------------------------------------------------
    @property
    def max_steps(self):
        return self._max_steps

    @max_steps.setter
    def max_steps(self, v):
        _lib.check(self._L.q3_set_limits(self._h, int(v), float(self._dt)))
        self._max_steps = int(v)

    @property
    def dt(self):
        return self._dt

    @dt.setter
    def dt(self, v):
        _lib.check(self._L.q3_set_limits(self._h, int(self._max_steps), float(v)))
        self._dt = float(v)

    def _get(self, want_states=True):
        n = self.num_envs
        st = torch.empty((n, 16), dtype=self.DTYPE, device=self.device) if want_states else None
        tg = torch.empty(n, dtype=torch.int32, device=self.device)
        sc = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self._L.q3_get_state(self._h, _ptr(st), _ptr(tg), _ptr(sc), self._stream()))
        return st, tg, sc

    def _set(self, states=None, target=None, steps=None):
        def dev(a, dtype):
            if a is None:
                return None
            t = torch.as_tensor(np.ascontiguousarray(a) if not torch.is_tensor(a) else a)
            return t.to(device=self.device, dtype=dtype).contiguous()

        st, tg, sc = dev(states, self.DTYPE), dev(target, torch.int32), dev(steps, torch.int32)
        for t, shape in ((st, (self.num_envs, 16)), (tg, (self.num_envs,)), (sc, (self.num_envs,))):
            if t is not None and tuple(t.shape) != shape:
                raise ValueError(f"expected shape {shape}, got {tuple(t.shape)}")
        _lib.check(self._L.q3_set_state(self._h, _ptr(st), _ptr(tg), _ptr(sc), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    @property
    def states(self):
        return self._get()[0].cpu().numpy()

    @states.setter
    def states(self, v):
        self._set(states=v)

    @property
    def step_counts(self):
        return self._get(False)[2].cpu().numpy().astype(self.NP_STEPS)

    @step_counts.setter
    def step_counts(self, v):
        self._set(steps=np.asarray(v).astype(np.int32))

    def get_state_tensors(self):
        """(states [N,16], target_gates [N] int32, step_counts [N] int32) as device tensors."""
        return self._get()

    def set_state_tensors(self, states=None, target=None, steps=None):
        self._set(states, target, steps)

    def seed(self, seed=None):
        """Upstream `seed()` is a no-op (resets use NumPy's global generator); here it re-keys the Philox reset stream."""
        if seed is not None:
            _lib.check(self._L.q3_seed(self._h, int(seed)))

    # ---- device-tensor path ----------------------------------------------------------------------------------------
    def reset_device(self, mask=None):
        m = None
        if mask is not None:
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
        _lib.check(self._L.q3_reset(self._h, _ptr(m), _ptr(self._states_d), self._stream()))
        return self._states_d

    def step_device(self, actions):
        """actions: float32 CUDA tensor [N,4] -> (states [N,16], rewards [N], dones [N] u8, truncated [N] u8), all views
        of internal device buffers that the next call overwrites."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or tuple(actions.shape) != (self.num_envs, 4):
            raise ValueError("actions must be a contiguous float32 CUDA tensor of shape [num_envs, 4]")
        if actions.device != self.device:
            raise ValueError(f"actions live on {actions.device}, the env on {self.device}")
        _lib.check(self._L.q3_step(self._h, _ptr(actions), _ptr(self._states_d), _ptr(self._rew_d), _ptr(self._done_d),
                                   _ptr(self._trunc_d), self._stream()))
        return self._states_d, self._rew_d, self._done_d, self._trunc_d

    def rollout_device(self, actions, want_states=True):
        """K steps in ONE kernel.  actions: float32 CUDA tensor [K,N,4] -> (rewards [K,N], dones [K,N] u8, final states)."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.dim() != 3 or \
                tuple(actions.shape[1:]) != (self.num_envs, 4) or actions.device != self.device:
            raise ValueError("actions must be a contiguous float32 tensor [K, num_envs, 4] on the env's device")
        K = int(actions.shape[0])
        rew = torch.empty((K, self.num_envs), dtype=self.DTYPE, device=self.device)
        done = torch.empty((K, self.num_envs), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.q3_step_many(self._h, _ptr(actions), K, _ptr(rew), _ptr(done),
                                        _ptr(self._states_d) if want_states else None, self._stream()))
        return rew, done, (self._states_d if want_states else None)

    def rollout_states_device(self, actions, out=None):
        """K steps in ONE kernel WITH the per-step rows a trainer consumes (q3_rollout): actions float32 CUDA [K,N,4] ->
        (states [K,N,16] = what step_wait() returns after each step, rewards [K,N], dones [K,N] u8, truncs [K,N] u8).  Bit for bit
        K x step_device()."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.dim() != 3 or \
                tuple(actions.shape[1:]) != (self.num_envs, 4) or actions.device != self.device:
            raise ValueError("actions must be a contiguous float32 tensor [K, num_envs, 4] on the env's device")
        K, n = int(actions.shape[0]), self.num_envs
        if out is None:
            out = (torch.empty((K, n, 16), dtype=self.DTYPE, device=self.device), torch.empty((K, n), dtype=self.DTYPE, device=self.device),
                   torch.empty((K, n), dtype=torch.uint8, device=self.device), torch.empty((K, n), dtype=torch.uint8, device=self.device))
        st, rew, done, trunc = out
        _lib.check(self._L.q3_rollout(self._h, _ptr(actions), K, _ptr(st), _ptr(rew), _ptr(done), _ptr(trunc), self._stream()))
        self._states_d.copy_(st[K - 1])
        return out

    # ---- the VecEnv surface of the reference ---------------------------------------------------------------------------
    def reset_(self, dones):
        return self.reset_device(np.asarray(dones).astype(np.uint8)).cpu().numpy()

    def reset(self):
        return self.reset_device().cpu().numpy()

    def step_async(self, actions):
        self.actions = actions

    def step_wait(self):
        self._act_h.copy_(torch.as_tensor(np.ascontiguousarray(self.actions, dtype=np.float32)))
        self._act_d.copy_(self._act_h, non_blocking=True)
        dev = self.step_device(self._act_d)
        # async D->H into pinned buffers, ONE synchronisation (two alternating sets: like upstream, which hands out its own
        # self.states, the state array is the env's buffer -- it survives the next step and is recycled by the one after)
        if getattr(self, "_host_sets", None) is None:
            self._host_sets = tuple(tuple(torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in dev) for _ in range(2))
            self._host_flip = 0
        self._host_flip ^= 1
        packed = self._host_sets[self._host_flip]
        for h, d in zip(packed, dev):
            h.copy_(d, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        states, rewards = packed[0].numpy(), packed[1].numpy().copy()
        dones, truncs = packed[2].numpy().astype(bool), packed[3].numpy().astype(bool)
        # Write info dicts: upstream builds `[{}] * num_envs`, i.e. ONE dict shared by every env
        info = {}
        if dones.any():
            info["terminal_observation"] = states[np.flatnonzero(dones)[-1]]  # written after reset_, like upstream
        if truncs.any():
            info["TimeLimit.truncated"] = True
        return states, rewards, dones, [info] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        pass

    def set_attr(self, attr_name, value, indices=None):
        pass

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        pass

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def render(self, mode='human'):
        """Dict with the 16 state columns and u1..u4 = (action + 1) / 2, as upstream."""
        state_dict = dict(zip(_KEYS, self.states.T))
        action_dict = dict(zip(['u1', 'u2', 'u3', 'u4'], (np.array(np.asarray(self.actions).T) + 1) / 2))
        return {**state_dict, **action_dict}


class Quadcopter3DVec(_Quad3DBase):
    """Hover task (Q3 cell 6): reach and hold the origin; float64 like the reference's np.zeros((N,16)) state."""

    KIND, DTYPE, NP_DTYPE, NP_STEPS = Q3_KIND_HOVER, torch.float64, np.float64, np.float64

    def __init__(self, num_envs, *, device=None, seed=0, env_id_base=0):
        self._create(num_envs, device, seed, env_id_base)
        self._thr = dict(pos=0.3, vel=0.3, ang=10 * np.pi / 180, rat=10 * np.pi / 180)

    def _set_thr(self, key, v):
        t = dict(self._thr, **{key: float(v)})
        _lib.check(self._L.q3_set_thresholds(self._h, t["pos"], t["vel"], t["ang"], t["rat"]))
        self._thr = t

    pos_threshold = property(lambda self: self._thr["pos"], lambda self, v: self._set_thr("pos", v))
    vel_threshold = property(lambda self: self._thr["vel"], lambda self, v: self._set_thr("vel", v))
    ang_threshold = property(lambda self: self._thr["ang"], lambda self, v: self._set_thr("ang", v))
    rat_threshold = property(lambda self: self._thr["rat"], lambda self, v: self._set_thr("rat", v))


class Quadcopter3DVecGates(_Quad3DBase):
    """Gate-sequence task (Q3 cell 14): float32, raw-state observation, episode ends after the last gate."""

    KIND, DTYPE, NP_DTYPE, NP_STEPS = Q3_KIND_GATES, torch.float32, np.float32, np.float32

    def __init__(self, num_envs, gates_pos, gate_yaw, start_pos, *, device=None, seed=0, env_id_base=0):
        self._create(num_envs, device, seed, env_id_base)
        self.start_pos = np.asarray(start_pos).astype(np.float32)
        self.gate_pos = np.asarray(gates_pos).astype(np.float32)
        self.gate_yaw = np.asarray(gate_yaw).astype(np.float32)
        self.num_gates = int(self.gate_pos.shape[0])
        gp, gy, sp = (np.ascontiguousarray(a) for a in (self.gate_pos, self.gate_yaw, self.start_pos))
        _lib.check(self._L.q3_set_track(self._h, _f32p(gp), _f32p(gy), self.num_gates, _f32p(sp)))

    @property
    def target_gates(self):
        return self._get(False)[1].cpu().numpy().astype(int)

    @target_gates.setter
    def target_gates(self, v):
        self._set(target=np.asarray(v).astype(np.int32))
