"""Host-side mirror of the reference's `Quadcopter3DGates` VecEnv (R:287-620, I:142-410) on top of libquadrace.

Same constructor, methods and public attributes as the reference class, so the reference's training cell
(SB3 `VecMonitor` + `PPO`, R:765-795), `animate_policy` (R:800-810) and the C-controller test (R:4487-4519)
can consume it unchanged -- but every step is one fused HIP kernel over all envs on the MI355X, and a
device-tensor fast path (`step_device`, `rollout_device`) avoids any host round trip.

There is no NumPy fallback: constructing an env without a gfx950 GPU raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import QR_VARIANT_E2E, QR_VARIANT_INDI

try:  # the reference imports `gymnasium.spaces` (R:284) / `gym.spaces` (I:139); neither is required here
    from gymnasium import spaces as _spaces  # type: ignore
except Exception:  # pragma: no cover - depends on the environment
    try:
        from gym import spaces as _spaces  # type: ignore
    except Exception:
        _spaces = None

try:
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv  # type: ignore
except Exception:  # pragma: no cover
    _SB3VecEnv = None


class Box:
    """Minimal stand-in for gymnasium.spaces.Box when gymnasium is not installed."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else tuple(np.shape(low))
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
        self.dtype = np.dtype(dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def _make_box(low, high, shape=None):
    if _spaces is not None:
        if shape is not None:
            return _spaces.Box(low=low, high=high, shape=shape, dtype=np.float32)
        return _spaces.Box(low=np.asarray(low, np.float32), high=np.asarray(high, np.float32), dtype=np.float32)
    return Box(low, high, shape)


class _VecEnvBase:
    """What SB3's abstract VecEnv provides to the reference class: the constructor contract and step()."""

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space
        self.render_mode = None

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()


_Base = _SB3VecEnv if _SB3VecEnv is not None else _VecEnvBase

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def default_residual_blob():
    """The reference's trained residual thrust/moment MLP weights (NNDroneModel/*.pt, R:227-229) as 740 f32."""
    return np.fromfile(os.path.join(_DATA, "residual_mlp_f32.bin"), dtype=np.float32)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _DeviceBackedArray(np.ndarray):
    """Host copy of a device-resident env attribute that writes itself back when it is modified in place.

    In the reference `world_states`, `disturbances`, `target_gates` and `step_counts` are plain NumPy arrays owned by
    the env, and callers edit them in place (`env.world_states[i] = ...`, `env.target_gates[:] = 0`, `R:4499-4500`).
    Here the state lives in HBM, so the attribute is a snapshot; item / slice assignment and the in-place operators
    on the snapshot -- or on any view derived from it -- push the whole (small, evaluation-time) array back to the
    device, which keeps that idiom working instead of silently editing a temporary."""

    def __new__(cls, arr, writeback):
        obj = np.asarray(arr).view(cls)
        obj._root, obj._writeback = obj, writeback
        return obj

    def __array_finalize__(self, src):
        if src is None:
            return
        root = getattr(src, "_root", None)
        same_memory = root is not None and np.shares_memory(self, root) if root is not None else False
        self._root = root if same_memory else None          # copies / results of arithmetic are ordinary arrays
        self._writeback = getattr(src, "_writeback", None) if same_memory else None

    def _push(self):
        if getattr(self, "_writeback", None) is not None and self._root is not None:
            self._writeback(np.asarray(self._root))

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        self._push()

    def _inplace(name):
        def op(self, other):
            r = getattr(np.ndarray, name)(self, other)
            self._push()
            return r
        op.__name__ = name
        return op

    for _n in ("__iadd__", "__isub__", "__imul__", "__itruediv__", "__ifloordiv__", "__imod__", "__iand__", "__ior__"):
        locals()[_n] = _inplace(_n)
    del _n, _inplace

    def fill(self, value):
        super().fill(value)
        self._push()


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Quadcopter3DGates(_Base):
    """End-to-end (motor command) Bebop race environment, vectorised on one MI355X.  Mirrors R:287-620.

    Extra keyword-only arguments (not in the reference): `device` (cuda ordinal), `seed`, `residual`
    ('default' = the reference's NNDroneModel weights, None = no residual model, or a 740-float blob),
    `env_id_base` (global index of env 0 when sharding over GPUs), `infos_mode` ('reference' reproduces the
    shared-dict behaviour of R:589-594, 'per_env' gives one dict per done env, 'sb3' gives one dict per done env whose
    `terminal_observation` is the TRUE final observation of the episode -- what SB3's time-limit bootstrap expects --
    and `TimeLimit.truncated` only for the envs that hit the time limit, 'none' skips the list).
    """

    VARIANT = QR_VARIANT_E2E
    STATE_LEN = 16
    _RENDER_KEYS = ['x', 'y', 'z', 'vx', 'vy', 'vz', 'phi', 'theta', 'psi', 'p', 'q', 'r', 'w1', 'w2', 'w3', 'w4']

    def __init__(self, num_envs, gates_pos, gate_yaw, start_pos, gates_ahead=0, pause_if_collision=False, *,
                 device=None, seed=0, residual="default", env_id_base=0, infos_mode="reference"):
        self._h = None
        self._L = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("Quadcopter3DGates needs a gfx950 GPU: libquadrace has no CPU fallback")
        self._dev_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self._dev_index)

        # Define the race track (R:298-302)
        self.start_pos = np.asarray(start_pos).astype(np.float32)
        self.gate_pos = np.asarray(gates_pos).astype(np.float32)
        self.gate_yaw = np.asarray(gate_yaw).astype(np.float32)
        self.num_gates = int(self.gate_pos.shape[0])
        self.gates_ahead = int(gates_ahead)
        self._pause_if_collision = bool(pause_if_collision)
        self.infos_mode = infos_mode

        cfg = _lib.QrConfig(self.VARIANT, int(num_envs), self.gates_ahead, self._dev_index,
                            int(self._pause_if_collision), 0, int(env_id_base))
        h = C.c_void_p()
        _lib.check(self._L.qr_create(C.byref(cfg), C.byref(h)))
        self._h = h
        gp = np.ascontiguousarray(self.gate_pos)
        gy = np.ascontiguousarray(self.gate_yaw)
        sp = np.ascontiguousarray(self.start_pos)
        _lib.check(self._L.qr_set_track(self._h, _f32p(gp), _f32p(gy), self.num_gates, _f32p(sp)))
        # relative gates as computed by the library (R:307-319)
        self.gate_pos_rel = np.zeros((self.num_gates, 3), dtype=np.float32)
        self.gate_yaw_rel = np.zeros(self.num_gates, dtype=np.float32)
        _lib.check(self._L.qr_get_track_tables(self._h, _f32p(self.gate_pos_rel), _f32p(self.gate_yaw_rel)))

        self.state_len = int(self._L.qr_obs_len(self._h))  # R:330 / I:185
        action_space = _make_box(-1, 1, shape=(4,))
        observation_space = _make_box(np.array([-np.inf] * self.state_len), np.array([np.inf] * self.state_len))
        _Base.__init__(self, int(num_envs), observation_space, action_space)

        n, dev = self.num_envs, self.device
        self._obs = torch.zeros((n, self.state_len), dtype=torch.float32, device=dev)
        self._last_obs = self._obs      # where the current observation lives: the env's own buffer, or row K-1 of the caller's
                                        # rollout buffer after a K-step call (a view: no copy kernel behind every rollout)
        self._rew = torch.zeros(n, dtype=torch.float32, device=dev)
        self._done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._trunc = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._act = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        self.actions = np.zeros((n, 4), dtype=np.float32)
        self.dones = np.zeros(n, dtype=bool)
        self.final_gate_passed = np.zeros(n, dtype=bool)

        self._max_steps = 1200          # R:345
        self._dt = np.float32(0.01)     # R:346
        self._pause = False             # R:360
        self._disturbance_ranges = np.zeros((6, 2), dtype=np.float32)  # R:355
        self._disturbance_scale = 1     # R:358
        if self.VARIANT == QR_VARIANT_E2E:
            if isinstance(residual, str):
                if residual != "default":
                    raise ValueError("residual must be 'default', None or a 740-float array")
                residual = default_residual_blob()
            self.set_residual(residual)
        self.seed(seed)

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._L.qr_destroy(self._h)
            self._h = None

    def set_residual(self, blob):
        if blob is None:
            _lib.check(self._L.qr_set_residual(self._h, None, 0))
        else:
            b = np.ascontiguousarray(blob, dtype=np.float32)
            _lib.check(self._L.qr_set_residual(self._h, _f32p(b), b.size))

    def _to_dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    # ------------------------------------------------------------------ reference attributes
    @property
    def max_steps(self):
        return self._max_steps

    @max_steps.setter
    def max_steps(self, v):  # `test_env.max_steps = 10000` (I:648)
        self._max_steps = int(v)
        _lib.check(self._L.qr_set_limits(self._h, self._max_steps, float(self._dt)))

    @property
    def dt(self):
        return self._dt

    @dt.setter
    def dt(self, v):
        self._dt = np.float32(v)
        _lib.check(self._L.qr_set_limits(self._h, self._max_steps, float(self._dt)))

    @property
    def pause(self):
        return self._pause

    @pause.setter
    def pause(self, v):
        self._pause = bool(v)
        _lib.check(self._L.qr_set_pause(self._h, int(self._pause)))

    @property
    def pause_if_collision(self):
        return self._pause_if_collision

    @pause_if_collision.setter
    def pause_if_collision(self, v):  # R:293 is a plain attribute read by step_wait (R:573): assignable at any time
        self._pause_if_collision = bool(v)
        if getattr(self, "_h", None) is not None:
            _lib.check(self._L.qr_set_pause_if_collision(self._h, int(self._pause_if_collision)))

    @property
    def disturbance_ranges(self):
        return self._disturbance_ranges

    @disturbance_ranges.setter
    def disturbance_ranges(self, v):  # `env.venv.disturbance_ranges = ...` (R:780-781)
        self._disturbance_ranges = np.array(v)
        self._push_disturbance()

    @property
    def disturbance_scale(self):
        return self._disturbance_scale

    @disturbance_scale.setter
    def disturbance_scale(self, v):
        self._disturbance_scale = v
        self._push_disturbance()

    def _push_disturbance(self):
        if self.VARIANT != QR_VARIANT_E2E:
            return
        r = np.ascontiguousarray(self._disturbance_ranges, dtype=np.float32).reshape(6, 2)
        _lib.check(self._L.qr_set_disturbance(self._h, _f32p(r), float(self._disturbance_scale)))

    def get_state_tensors(self):
        """(world[N,S], disturbances[N,6] or None, target[N] i32, steps[N] i32, episode[N] u32->i64) on device."""
        n, dev = self.num_envs, self.device
        world = torch.empty((n, self.STATE_LEN), dtype=torch.float32, device=dev)
        dist = torch.empty((n, 6), dtype=torch.float32, device=dev) if self.VARIANT == QR_VARIANT_E2E else None
        target = torch.empty(n, dtype=torch.int32, device=dev)
        steps = torch.empty(n, dtype=torch.int32, device=dev)
        episode = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(self._L.qr_get_state(self._h, _ptr(world), _ptr(dist), _ptr(target), _ptr(steps), _ptr(episode),
                                        self._stream()))
        return world, dist, target, steps, episode

    def set_state_tensors(self, world=None, dist=None, target=None, steps=None, episode=None):
        w = None if world is None else self._to_dev(world, torch.float32)
        d = None if dist is None or self.VARIANT != QR_VARIANT_E2E else self._to_dev(dist, torch.float32)
        t = None if target is None else self._to_dev(target, torch.int32)
        s = None if steps is None else self._to_dev(steps, torch.int32)
        ep = None if episode is None else self._to_dev(episode, torch.int32)
        if w is not None:
            assert tuple(w.shape) == (self.num_envs, self.STATE_LEN), w.shape
        _lib.check(self._L.qr_set_state(self._h, _ptr(w), _ptr(d), _ptr(t), _ptr(s), _ptr(ep), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()  # staging tensors may be freed after return

    @property
    def world_states(self):
        return _DeviceBackedArray(self.get_state_tensors()[0].cpu().numpy(), lambda a: self.set_state_tensors(world=a))

    @world_states.setter
    def world_states(self, v):
        self.set_state_tensors(world=v)

    @property
    def disturbances(self):
        d = self.get_state_tensors()[1]
        return None if d is None else _DeviceBackedArray(d.cpu().numpy(), lambda a: self.set_state_tensors(dist=a))

    @disturbances.setter
    def disturbances(self, v):
        self.set_state_tensors(dist=v)

    @property
    def target_gates(self):
        return _DeviceBackedArray(self.get_state_tensors()[2].cpu().numpy().astype(np.int64),
                                  lambda a: self.set_state_tensors(target=np.asarray(a, dtype=np.int32)))

    @target_gates.setter
    def target_gates(self, v):
        self.set_state_tensors(target=np.asarray(v, dtype=np.int32))

    @property
    def step_counts(self):
        return _DeviceBackedArray(self.get_state_tensors()[3].cpu().numpy().astype(np.int64),
                                  lambda a: self.set_state_tensors(steps=np.asarray(a, dtype=np.int32)))

    @step_counts.setter
    def step_counts(self, v):
        self.set_state_tensors(steps=np.asarray(v, dtype=np.int32))

    @property
    def states(self):
        """Observation array [N, state_len] of the last reset()/step() (R:343, read by animate_policy R:803)."""
        return self._current_obs().cpu().numpy()

    @property
    def states_tensor(self):
        """The current observation on the device: always the env's OWN buffer.  A K-step call (rollout_device /
        step_sequence_device) leaves its last observation in row K-1 of the caller's rollout buffer and does not copy it (that copy
        was a 6 MB kernel behind every rollout); the FIRST read here takes the copy, so readers get a stable buffer that later
        reuse of the rollout buffer cannot change.  Read it before overwriting that buffer yourself."""
        return self._current_obs()

    def _current_obs(self):
        if self._last_obs is not self._obs:     # first read after a K-step call: row K-1 of the caller's buffer -> own buffer
            self._obs.copy_(self._last_obs)
            self._last_obs = self._obs
        return self._obs

    # ------------------------------------------------------------------ reference methods
    def update_states(self):
        """update_states_gate (R:365-450)."""
        _lib.check(self._L.qr_observe(self._h, _ptr(self._obs), self._stream()))
        self._last_obs = self._obs

    update_states_gate = update_states

    def update_states_world(self):
        """Alternative observation of the reference (R:362-363, unused by default): the raw world state."""
        return self.world_states

    def reset_(self, dones):
        mask = self._to_dev(np.asarray(dones, dtype=np.uint8), torch.uint8)
        _lib.check(self._L.qr_reset(self._h, _ptr(mask), _ptr(self._obs), self._stream()))
        self._last_obs = self._obs
        return self.states

    def reset(self):
        self.reset_device()
        return self.states

    def step_async(self, actions):
        """Stores the actions (R:498-499) and, unlike the reference, already enqueues the work on the device: the H->D copy of the
        actions, the step kernel and the D->H copies of its results run while the caller does whatever it does before
        step_wait(), which only synchronises."""
        self.actions = actions
        self._pending = self._enqueue_host_step()

    def _ensure_host(self):
        """Pinned host staging: TWO sets of result buffers used alternately (the arrays handed out by one step stay intact
        during the next one) and one action buffer."""
        if getattr(self, "_host", None) is None:
            n = self.num_envs

            def one_set():
                return (torch.empty((n, self.state_len), dtype=torch.float32).pin_memory(),
                        torch.empty(n, dtype=torch.float32).pin_memory(),
                        torch.empty(n, dtype=torch.uint8).pin_memory(),
                        torch.empty(n, dtype=torch.uint8).pin_memory())

            self._host = (one_set(), one_set())
            self._host_act = torch.empty((n, 4), dtype=torch.float32).pin_memory()
            self._host_flip = 0

    def _host_buffers(self):
        """the result buffer set of THIS step (alternates)"""
        self._ensure_host()
        self._host_flip ^= 1
        return self._host[self._host_flip]

    def _actions_to_device(self, actions):
        """H->D copy of the SB3-side action array through a pinned buffer into the env's own device tensor (a pageable
        source makes the runtime stage the copy itself, synchronously: 60 us instead of 29 us for the 1 MB at N = 65 536)."""
        if isinstance(actions, torch.Tensor):
            if actions.is_cuda:
                return self._to_dev(actions, torch.float32)
            actions = actions.detach().numpy()
        a = np.asarray(actions, dtype=np.float32)
        if a.shape != (self.num_envs, 4):
            raise ValueError(f"actions must have shape ({self.num_envs}, 4), got {a.shape}")
        self._ensure_host()
        self._host_act.numpy()[...] = a
        self._act.copy_(self._host_act, non_blocking=True)
        return self._act

    def _enqueue_host_step(self):
        act = self._actions_to_device(self.actions)
        if self.infos_mode == "sb3" and getattr(self, "_term_obs_buf", None) is None:
            self.set_terminal_obs_buffer(torch.zeros((self.num_envs, self.state_len), dtype=torch.float32, device=self.device))
        dev = self.step_device(act)
        host = self._host_buffers()
        for h, d in zip(host, dev):
            h.copy_(d, non_blocking=True)
        return host

    def step_wait(self):
        """SB3-facing step: NumPy in, NumPy out (one H->D copy of the actions through a pinned buffer, one batch of async
        D->H copies into pinned buffers, one synchronisation).  Like the reference, which returns its own `self.states`
        (R:595), the observation array is the env's buffer, not a copy: here one of two alternating pinned buffers, so it is
        overwritten by the step AFTER the next one.  Rewards and dones are fresh arrays.  Use step_device() to stay on the GPU."""
        host = getattr(self, "_pending", None)
        if host is None:   # step_wait() without a step_async() before it: the reference steps again with self.actions
            host = self._enqueue_host_step()
        self._pending = None
        torch.cuda.current_stream(self.device).synchronize()
        obs_np, rew_np = host[0].numpy(), host[1].numpy().copy()
        done_np, trunc_np = host[2].numpy().astype(bool), host[3].numpy().view(np.bool_)   # kernels write 0 / 1
        self.dones = done_np
        infos = self._make_infos(obs_np, done_np, trunc_np)
        return obs_np, rew_np, done_np, infos

    def _make_infos(self, obs_np, done_np, trunc_np):
        if self.infos_mode == "none":
            return []
        if self.infos_mode == "reference":
            # R:589-594: `[{}] * N` aliases ONE dict, so every env sees terminal_observation (the post-reset row
            # of the last done env) as soon as any env is done, and TimeLimit.truncated if any env timed out.
            shared = {}
            if done_np.any():
                shared["terminal_observation"] = obs_np[np.nonzero(done_np)[0][-1]]
            if trunc_np.any():
                shared["TimeLimit.truncated"] = True
            return [shared] * self.num_envs
        infos = [{} for _ in range(self.num_envs)]
        if self.infos_mode == "sb3":
            if done_np.any():
                idx = np.nonzero(done_np)[0]
                if self._pause_if_collision:   # no auto-reset in this mode (R:573-578): the row handed back IS the final observation
                    term = obs_np[idx]
                else:
                    tb = self._term_obs_buf if self._term_obs_buf.dim() == 2 else self._term_obs_buf[0]   # qr_step writes row [0][env]
                    term = tb[torch.as_tensor(idx, device=self.device)].cpu().numpy()   # rows written by the step kernel
                for j, i in enumerate(idx):
                    infos[i]["terminal_observation"] = term[j]
                    infos[i]["TimeLimit.truncated"] = bool(trunc_np[i])
            return infos
        if done_np.any():
            for i in np.nonzero(done_np)[0]:
                infos[i]["terminal_observation"] = obs_np[i]
                if trunc_np[i]:
                    infos[i]["TimeLimit.truncated"] = True
        return infos

    def seed(self, seed=None):
        """Upstream seed() is a no-op (R:600-601); here it keys the in-kernel Philox reset stream."""
        _lib.check(self._L.qr_seed(self._h, int(0 if seed is None else seed)))
        return [seed] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        raise AttributeError()  # R:603-604: makes SB3 >= 2.0 fall back to render_mode=None

    def set_attr(self, attr_name, value, indices=None):
        pass

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        pass

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def render(self, mode='human'):
        """Dict of per-env arrays consumed by quadcopter_animation (R:615-620)."""
        state_dict = dict(zip(self._RENDER_KEYS, self.world_states.T))
        a = self.actions.detach().cpu().numpy() if isinstance(self.actions, torch.Tensor) else np.array(self.actions)
        action_dict = dict(zip(['u1', 'u2', 'u3', 'u4'], (a.T + 1) / 2))
        return {**state_dict, **action_dict}

    # ------------------------------------------------------------------ device fast path (no host sync)
    def set_terminal_obs_buffer(self, buf):
        """Register (or, with None, remove) a float32 CUDA tensor that receives the TRUE terminal observation -- the
        gate-frame observation of an episode's final state, taken before the auto-reset -- of every env that finishes
        at a step: shape [N, L] for step_device, [K, N, L] for rollout_device / step_sequence_device /
        rollout_policy_device (row [k, env]).  Rows of envs that did not finish are left untouched.  This is what SB3
        bootstraps time-limit truncations from; the reference itself fills `terminal_observation` after reset_()
        (R:589-594), i.e. with the first observation of the NEXT episode."""
        rows = 0
        if buf is not None:
            assert buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous() and buf.shape[-1] == self.state_len
            assert buf.dim() in (2, 3) and buf.shape[-2] == self.num_envs
            rows = int(buf.shape[0]) if buf.dim() == 3 else 1   # K-step calls with K > rows are refused by the library
        self._term_obs_buf = buf   # keep it alive while the library holds the pointer
        _lib.check(self._L.qr_set_terminal_obs(self._h, _ptr(buf), rows))

    def probe_residual(self):
        """[N, 7] device tensor (vb[3], residual thrust, residual moment[3]) of the current states (E2E + residual only)."""
        out = torch.empty((self.num_envs, 7), dtype=torch.float32, device=self.device)
        _lib.check(self._L.qr_probe_residual(self._h, _ptr(out), self._stream()))
        return out

    def reset_device(self):
        _lib.check(self._L.qr_reset(self._h, None, _ptr(self._obs), self._stream()))
        self._last_obs = self._obs
        return self._obs

    def step_device(self, actions):
        """actions: float32 CUDA tensor [N,4].  Returns (obs, reward, done u8, trunc u8) device tensors; they are
        the env's own buffers and are overwritten by the next step (clone them to keep a copy)."""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        assert tuple(actions.shape) == (self.num_envs, 4)
        obs, rew, done, trunc = self._obs, self._rew, self._done, self._trunc
        _lib.check(self._L.qr_step(self._h, _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done), _ptr(trunc),
                                   self._stream()))
        if not self._pause:   # with pause set the kernel leaves the observation untouched (R:570-572)
            self._last_obs = obs
        return obs, rew, done, trunc

    ROLLOUT_FORMS = {"auto": 0, "multi_wave": 1, "general": 2, "general_multi_wave": 3, "one_wave": 4}

    def set_rollout_form(self, form):
        """Which family of fused kernels `rollout_device` may use: "auto" (by env count and mode), "multi_wave" (the forms built for more
        than one workgroup per CU, at any env count), "general" (the general kernels for every launch), "general_multi_wave" (both), "one_wave" (the one-wave forms at any env count).  All bit-identical: a test / A-B hook."""
        _lib.check(self._L.qr_set_rollout_form(self._h, self.ROLLOUT_FORMS[form]))
        return self

    def rollout_kernel_name(self):
        """Symbol of the device kernel rollout_device() launches on this env right now, as rocprofv3 prints it."""
        name = self._L.qr_rollout_kernel_name(self._h).decode()
        return "qr::%s<%d, %d>" % (name, int(self.VARIANT), self.gates_ahead)

    def rollout_device(self, actions, out=None):
        """K steps with pre-recorded actions [K,N,4] -> (obs[K,N,L], reward[K,N], done[K,N], trunc[K,N])."""
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        K = actions.shape[0]
        assert tuple(actions.shape) == (K, self.num_envs, 4)
        n, dev = self.num_envs, self.device
        if out is None:
            out = (torch.empty((K, n, self.state_len), dtype=torch.float32, device=dev),
                   torch.empty((K, n), dtype=torch.float32, device=dev),
                   torch.empty((K, n), dtype=torch.uint8, device=dev),
                   torch.empty((K, n), dtype=torch.uint8, device=dev))
        obs, rew, done, trunc = out
        _lib.check(self._L.qr_step_many(self._h, int(K), _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done), _ptr(trunc),
                                        self._stream()))
        if not self._pause:
            self._last_obs = obs[K - 1]
        return out

    def step_sequence_device(self, actions, out):
        """K per-step kernel launches (qr_step each) writing into rollout buffers `out` -- what a closed-loop
        caller does when the policy runs between steps.  Same results as rollout_device (fused kernel)."""
        K = actions.shape[0]
        obs, rew, done, trunc = out
        _lib.check(self._L.qr_step_launches(self._h, int(K), _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done),
                                            _ptr(trunc), self._stream()))
        if not self._pause:
            self._last_obs = obs[K - 1]
        return out

    def rollout_policy_device(self, policy, num_steps, log_std, noise_seed=0, first_step=0, deterministic=False, out=None,
                              precision="f16-operands"):
        """Closed-loop rollout in ONE kernel (qr_rollout_policy): K x [obs -> MFMA policy -> sample -> env.step].
        `policy` is an optimal_quad_control_rl_amd.policy.MfmaPolicy.  Returns device tensors
        (obs[K,N,L], actions[K,N,4] unclipped, logp[K,N], reward[K,N], done[K,N] u8, trunc[K,N] u8, last_obs[N,L]).
        precision="f32": the policy forward inside the kernel at the reference's precision (QR_ROLLOUT_F32CLASS)."""
        if precision not in ("f16-operands", "f32"):
            raise ValueError("precision must be 'f16-operands' or 'f32'")
        K, n, dev = int(num_steps), self.num_envs, self.device
        if out is None:
            out = (torch.empty((K, n, self.state_len), dtype=torch.float32, device=dev),
                   torch.empty((K, n, 4), dtype=torch.float32, device=dev),
                   torch.empty((K, n), dtype=torch.float32, device=dev),
                   torch.empty((K, n), dtype=torch.float32, device=dev),
                   torch.empty((K, n), dtype=torch.uint8, device=dev),
                   torch.empty((K, n), dtype=torch.uint8, device=dev))
        obs, act, logp, rew, done, trunc = out
        ls = np.ascontiguousarray(log_std.detach().cpu().numpy() if isinstance(log_std, torch.Tensor) else log_std,
                                  dtype=np.float32).reshape(4)
        _lib.check(self._L.qr_rollout_policy(self._h, policy._h, K, _f32p(ls), int(noise_seed), int(first_step),
                                             int(bool(deterministic)) | (2 if precision == "f32" else 0), _ptr(obs), _ptr(act), _ptr(logp), _ptr(rew),
                                             _ptr(done), _ptr(trunc), _ptr(self._obs), self._stream()))
        self._last_obs = self._obs
        return obs, act, logp, rew, done, trunc, self._obs

    def profile_rollout(self, actions, out):
        """Like rollout_device but every step kernel is bracketed by its own hipEvent pair on the launch stream.
        Returns (mean single-kernel duration in ms, whole-region ms).  Blocks."""
        K = actions.shape[0]
        obs, rew, done, trunc = out
        mean_ms, region_ms = C.c_float(), C.c_float()
        _lib.check(self._L.qr_profile_steps(self._h, int(K), _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done),
                                            _ptr(trunc), self._stream(), C.byref(mean_ms), C.byref(region_ms)))
        return float(mean_ms.value), float(region_ms.value)

    def set_timing(self, on):
        """hipEvent bracket around every K-step call (read back by last_rollout_ms()); off = two marker packets fewer per call"""
        _lib.check(self._L.qr_set_timing(self._h, int(bool(on))))

    def last_rollout_ms(self):
        ms = C.c_float()
        _lib.check(self._L.qr_last_step_many_ms(self._h, C.byref(ms)))
        return float(ms.value)


class Quadcopter3DGatesINDI(Quadcopter3DGates):
    """INDI inner-loop variant: actions are (p_cmd, q_cmd, r_cmd, T_cmd), 13-state model.  Mirrors I:142-410."""

    VARIANT = QR_VARIANT_INDI
    STATE_LEN = 13
    _RENDER_KEYS = ['x', 'y', 'z', 'vx', 'vy', 'vz', 'phi', 'theta', 'psi', 'p', 'q', 'r', 'T']

    def __init__(self, num_envs, gates_pos, gate_yaw, start_pos, gates_ahead=0, pause_if_collision=False, **kw):
        kw.pop("residual", None)
        super().__init__(num_envs, gates_pos, gate_yaw, start_pos, gates_ahead, pause_if_collision, residual=None, **kw)

    def get_attr(self, attr_name, indices=None):
        return None  # I:393-394
