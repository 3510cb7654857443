"""ctypes binding of libquadrace.so (the C ABI declared in include/quadrace.h).

This is the reference's own FFI style (ctypes.CDLL + POINTER(c_float), R:4395-4417).  The library is the
hand-written HIP implementation; there is no Python/NumPy fallback: if it cannot be built or loaded, or no
gfx950 GPU is visible, the caller gets an exception.
"""
import ctypes as C
import os

from . import build as _build

QR_OK, QR_E_INVALID, QR_E_NO_DEVICE, QR_E_HIP, QR_E_STATE = 0, -1, -2, -3, -4
QR_VARIANT_E2E, QR_VARIANT_INDI = 0, 1
QR_MAX_GATES, QR_MAX_GATES_AHEAD, QR_RESIDUAL_FLOATS = 32, 4, 740

_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p


class QrConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("num_envs", C.c_int32), ("gates_ahead", C.c_int32), ("device", C.c_int32),
                ("pause_if_collision", C.c_int32), ("reserved0", C.c_int32), ("env_id_base", C.c_uint64)]


# name -> (restype, argtypes); must list every symbol of include/quadrace.h and include/quad3d.h
SIGNATURES = {
    "qr_abi_version": (C.c_int, []),
    "qr_last_error": (C.c_char_p, []),
    "qr_create": (C.c_int, [C.POINTER(QrConfig), C.POINTER(_vp)]),
    "qr_destroy": (C.c_int, [_vp]),
    "qr_state_len": (C.c_int, [_vp]),
    "qr_obs_len": (C.c_int, [_vp]),
    "qr_num_envs": (C.c_int, [_vp]),
    "qr_set_track": (C.c_int, [_vp, _f32p, _f32p, C.c_int32, _f32p]),
    "qr_get_track_tables": (C.c_int, [_vp, _f32p, _f32p]),
    "qr_set_residual": (C.c_int, [_vp, _f32p, C.c_size_t]),
    "qr_set_disturbance": (C.c_int, [_vp, _f32p, C.c_float]),
    "qr_set_limits": (C.c_int, [_vp, C.c_int32, C.c_float]),
    "qr_set_pause": (C.c_int, [_vp, C.c_int32]),
    "qr_set_pause_if_collision": (C.c_int, [_vp, C.c_int32]),
    "qr_set_terminal_obs": (C.c_int, [_vp, _vp, C.c_int32]),
    "qr_set_timing": (C.c_int, [_vp, C.c_int32]),
    "qr_rollout_kernel_name": (C.c_char_p, [_vp]),
    "qr_set_rollout_form": (C.c_int, [_vp, C.c_int32]),
    "qr_seed": (C.c_int, [_vp, C.c_uint64]),
    "qr_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "qr_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qr_step_many": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qr_step_launches": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qr_observe": (C.c_int, [_vp, _vp, _vp]),
    "qr_probe_residual": (C.c_int, [_vp, _vp, _vp]),
    "qr_get_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qr_set_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qr_last_step_many_ms": (C.c_int, [_vp, _f32p]),
    "qr_policy_create": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "qr_policy_destroy": (C.c_int, [_vp]),
    "qr_policy_last_error": (C.c_char_p, []),
    "qr_policy_set_weights": (C.c_int, [_vp] + [_f32p] * 8),
    "qr_policy_forward": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "qr_policy_forward_f32class": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "qr_rollout_policy": (C.c_int, [_vp, _vp, C.c_int32, _f32p, C.c_uint64, C.c_uint64, C.c_int32, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp]),
    "qr_profile_steps": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _f32p, _f32p]),
    "qr_ppo_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "qr_ppo_create_ex": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "qr_ppo_destroy": (C.c_int, [_vp]),
    "qr_ppo_num_params": (C.c_int, [_vp]),
    "qr_ppo_pack": (C.c_int, [_vp, _vp, _vp]),
    "qr_ppo_grad": (C.c_int, [_vp] * 8 + [C.c_int32, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "qr_ppo_grad_f32class": (C.c_int, [_vp] * 8 + [C.c_int32, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "qr_ppo_minibatch": (C.c_int, [_vp] * 10 + [C.c_int32] + [C.c_float] * 8 + [C.c_int32, _vp, _vp]),
    "qr_ppo_forward": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "qr_ppo_gae": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float] + [_vp] * 7),
    "qr_ppo_apply": (C.c_int, [_vp] * 5 + [C.c_int32] + [C.c_float] * 5 + [C.c_int32, _vp, _vp]),
    "qr_ppo_epoch_begin": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "qr_ppo_epoch": (C.c_int, [_vp] * 10 + [C.c_int32] * 4 + [C.c_float] * 8 + [_vp, _vp]),
    "qr_ppo_shuffle_state": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int32, _vp]),
    "qr_ppo_adam_step": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32, _vp]),
    "qr_ppo_control": (C.c_int, [_vp, C.c_float, C.c_int32, _vp]),
    "qr_ppo_status": (C.c_int, [_vp, C.POINTER(C.c_int32), _vp]),
    # include/quad3d.h (predecessor environments of "3D quad.ipynb")
    "q3_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_vp)]),
    "q3_destroy": (C.c_int, [_vp]),
    "q3_num_envs": (C.c_int, [_vp]),
    "q3_elem_size": (C.c_int, [_vp]),
    "q3_set_track": (C.c_int, [_vp, _f32p, _f32p, C.c_int32, _f32p]),
    "q3_set_limits": (C.c_int, [_vp, C.c_int32, C.c_double]),
    "q3_set_thresholds": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double]),
    "q3_seed": (C.c_int, [_vp, C.c_uint64]),
    "q3_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "q3_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "q3_step_many": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "q3_rollout": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "q3_get_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "q3_set_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
}


ADDED_IN_ROUND_5 = ("qr_set_rollout_form", "qr_ppo_create_ex", "q3_rollout", "qr_policy_forward_f32class", "qr_ppo_grad_f32class")   # (and round 6)


class QuadraceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libquadrace error {code}: {msg}")
        self.code = code


_lib = None


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load libquadrace.so (building it with hipcc first if it is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and _build.needs_build():
        _build.build_native_locked()   # one builder at a time (torchrun starts every rank at once)
    if not os.path.exists(_build.LIB):
        raise RuntimeError(f"{_build.LIB} is missing: run `python -m optimal_quad_control_rl_amd.build`")
    L = C.CDLL(_build.LIB)
    for name, (rt, at) in SIGNATURES.items():
        if name in ADDED_IN_ROUND_5 and not hasattr(L, name) and os.environ.get("QR_PROBE_LIB"):
            continue   # a forensic build of an older source tree (tools/isa_patch.py): ABI 3 is additive, the older library lacks the entry
        fn = getattr(L, name)  # AttributeError here = ABI drift between header and library
        fn.restype, fn.argtypes = rt, at
    if L.qr_abi_version() != 3:
        raise RuntimeError("libquadrace ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc != QR_OK:
        raise QuadraceError(rc, load().qr_last_error().decode("utf-8", "replace"))
