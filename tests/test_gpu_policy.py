"""Policy network on the matrix cores vs (a) the reference's own generated C policy `nn_forward`
(c_code/neural_network.c, fixture F10 holds its baked weights and 512 input/output rows) and (b) torch float32.
Tolerance: the kernel feeds f16 operands to the MFMA (f32 accumulation), so agreement is at the f16 level:
|d mean| <= 4e-3 * max(1, |mean|_inf of the row set); the reference's policy outputs are O(0.1 - 1)."""
import numpy as np
import pytest
import torch

import parity as P

pytestmark = pytest.mark.gpu


def _layers(d):
    return [(d["w1"], d["b1"]), (d["w2"], d["b2"]), (d["w3"], d["b3"]), (d["w4"], d["b4"])]


def test_policy_matches_reference_nn_forward():
    from optimal_quad_control_rl_amd.policy import MfmaPolicy

    d = P.load("f10_policy")
    pol = MfmaPolicy(24).set_weights(_layers(d))
    out = pol.forward(torch.as_tensor(d["obs"]).cuda()).cpu().numpy()
    err = np.abs(out - d["mean"]).max()
    print("max |mean - nn_forward| =", err, " mean scale", np.abs(d["mean"]).max())
    assert err < 4e-3 * max(1.0, np.abs(d["mean"]).max())
    # ... and it is not trivially small: f32 torch agrees with nn_forward far better, the kernel at f16 level
    assert np.abs(out - d["mean"]).mean() < 1e-3


@pytest.mark.parametrize("L", [13, 17, 20, 24, 36])
@pytest.mark.parametrize("n", [1, 63, 64, 1000, 65536])
def test_policy_matches_torch(L, n):
    from optimal_quad_control_rl_amd.policy import MfmaPolicy
    from optimal_quad_control_rl_amd.ppo import ActorCritic

    torch.manual_seed(L * 1000 + n % 997)
    net = ActorCritic(L, 4).cuda()
    with torch.no_grad():  # make biases and the output head non-trivial
        for m in net.pi:
            if isinstance(m, torch.nn.Linear):
                m.bias.uniform_(-0.3, 0.3)
        net.pi[-1].weight.mul_(30.0)
    obs = (torch.randn(n, L, device="cuda") * 2.0).contiguous()
    pol = MfmaPolicy(L).load_torch(net.pi)
    out = pol.forward(obs)
    with torch.no_grad():
        ref = net.pi(obs)
    err = (out - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert out.shape == (n, 4)
    assert err < 5e-3 * scale, (err, scale)


def test_policy_errors():
    import ctypes as C
    from optimal_quad_control_rl_amd import _lib

    L = _lib.load()
    h = C.c_void_p()
    assert L.qr_policy_create(23, 0, C.byref(h)) == _lib.QR_E_INVALID
    assert L.qr_policy_create(24, 0, C.byref(h)) == 0
    assert L.qr_policy_forward(h, 4, C.c_void_p(8), C.c_void_p(8), None) == _lib.QR_E_STATE  # no weights yet
    assert L.qr_policy_destroy(h) == 0
