"""Policy network on the MI355X matrix cores (`qr_policy_*` in include/quadrace.h).

The MLP the reference trains and deploys -- obs -> 120 -> 120 -> 120 -> 4, ReLU (SB3 `MlpPolicy` with
`net_arch pi=[120,120,120]`, R:783; generated C twin `c_code/neural_network.c:419-430`) -- evaluated by a hand-written
f16-operand / f32-accumulate MFMA kernel on device tensors: `model.predict(env.states, deterministic=True)` (R:801)
without torch's per-layer kernels.  No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class MfmaPolicy:
    def __init__(self, obs_len, device=None):
        self._L = _lib.load()
        self._h = None
        if not torch.cuda.is_available():
            raise RuntimeError("MfmaPolicy needs a gfx950 GPU: libquadrace has no CPU fallback")
        self.obs_len = int(obs_len)
        self._dev_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self._dev_index)
        h = C.c_void_p()
        rc = self._L.qr_policy_create(self.obs_len, self._dev_index, C.byref(h))
        if rc:
            raise _lib.QuadraceError(rc, self._L.qr_policy_last_error().decode())
        self._h = h

    def close(self):
        if self._h is not None:
            self._L.qr_policy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, layers):
        """layers = [(W1[120,L], b1[120]), (W2[120,120], b2), (W3[120,120], b3), (W4[4,120], b4)], torch Linear layout."""
        arrs = []
        shapes = [(120, self.obs_len), (120,), (120, 120), (120,), (120, 120), (120,), (4, 120), (4,)]
        for w, b in layers:
            for a in (w, b):
                if isinstance(a, torch.Tensor):
                    a = a.detach().cpu().numpy()
                arrs.append(np.ascontiguousarray(a, dtype=np.float32))
        assert [a.shape for a in arrs] == shapes, [a.shape for a in arrs]
        rc = self._L.qr_policy_set_weights(self._h, *[_f32p(a) for a in arrs])
        if rc:
            raise _lib.QuadraceError(rc, self._L.qr_policy_last_error().decode())
        return self

    def load_torch(self, sequential):
        """Take the weights of a torch `nn.Sequential(Linear, ReLU, Linear, ReLU, Linear, ReLU, Linear)`."""
        lin = [m for m in sequential if isinstance(m, torch.nn.Linear)]
        assert len(lin) == 4
        return self.set_weights([(m.weight, m.bias) for m in lin])

    def forward(self, obs, out=None, precision="f16-operands"):
        """obs: float32 CUDA tensor [n, obs_len] -> action means [n, 4] (not clipped).  Enqueued on the current stream.
        precision: "f16-operands" (the throughput kernel) or "f32" (qr_policy_forward_f32class: every operand as two f16 pieces,
        float32-class results -- the reference's precision, ~3 x the matrix work)."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.obs_len
        if precision not in ("f16-operands", "f32"):
            raise ValueError("precision must be 'f16-operands' or 'f32'")
        n = obs.shape[0]
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=obs.device)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        fn = self._L.qr_policy_forward if precision == "f16-operands" else self._L.qr_policy_forward_f32class
        rc = fn(self._h, n, C.c_void_p(obs.data_ptr()), C.c_void_p(out.data_ptr()), st)
        if rc:
            raise _lib.QuadraceError(rc, self._L.qr_policy_last_error().decode())
        return out
