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


def test_f32class_policy_matches_reference_nn_forward_at_float32_level():
    """Round 6: the reference-precision forward (every operand as two f16 pieces, three matrix instructions per
    K-step, f32 accumulation) against the reference's own generated float32 `nn_forward` (F10) and against float64: the error is at the
    float32 rounding level of the network's outputs (nn_forward itself differs from float64 by about as much), 300 x below the
    f16-operand kernel's."""
    from optimal_quad_control_rl_amd.policy import MfmaPolicy

    d = P.load("f10_policy")
    pol = MfmaPolicy(24).set_weights(_layers(d))
    obs = torch.as_tensor(d["obs"]).cuda()
    out = pol.forward(obs, precision="f32").cpu().numpy()
    out16 = pol.forward(obs).cpu().numpy()
    x = d["obs"].astype(np.float64)
    for k, (w, b) in enumerate(_layers(d)):
        x = x @ np.asarray(w, np.float64).T + np.asarray(b, np.float64)
        if k < 3:
            x = np.maximum(x, 0.0)
    e_ref, e64, e16 = np.abs(out - d["mean"]).max(), np.abs(out - x).max(), np.abs(out16 - d["mean"]).max()
    ref64 = np.abs(d["mean"] - x).max()
    print("f32-class: max |mean - nn_forward| = %.3e, vs float64 %.3e (nn_forward vs float64 %.3e); f16-operand kernel %.3e" % (e_ref, e64, ref64, e16))
    assert e_ref <= 2e-6 * max(1.0, np.abs(d["mean"]).max())
    assert e64 <= 2e-6 * max(1.0, np.abs(x).max())
    assert e16 > 50 * e_ref                     # the two kernels really are different arithmetic


@pytest.mark.parametrize("L", [13, 17, 20, 24, 36])
@pytest.mark.parametrize("n", [1, 63, 1000, 65536])
def test_f32class_policy_matches_float64_torch(L, n):
    """... for every observation length, ragged and full launches, large inputs (rates up to 1000 rad/s) and a non-trivial output head:
    against the float64 evaluation of the same float32 parameters, error <= 4e-6 of the output scale (float32 torch: the same level)."""
    from optimal_quad_control_rl_amd.policy import MfmaPolicy
    from optimal_quad_control_rl_amd.ppo import ActorCritic

    torch.manual_seed(L * 1000 + n % 997)
    net = ActorCritic(L, 4).cuda()
    with torch.no_grad():
        for m in net.pi:
            if isinstance(m, torch.nn.Linear):
                m.bias.uniform_(-0.3, 0.3)
        net.pi[-1].weight.mul_(30.0)
    obs = (torch.randn(n, L, device="cuda") * 2.0).contiguous()
    obs[:, min(9, L - 1)] *= 300.0                      # a body-rate column near the 1000 rad/s guard
    pol = MfmaPolicy(L).load_torch(net.pi)
    out = pol.forward(obs, precision="f32")
    with torch.no_grad():
        ref64 = net.pi.double()(obs.double())
        ref32 = net.pi.float()(obs)
    scale = max(1.0, ref64.abs().max().item())
    err, err32 = (out.double() - ref64).abs().max().item(), (ref32.double() - ref64).abs().max().item()
    assert out.shape == (n, 4)
    assert err <= 4e-6 * scale, (err, err32, scale)


def test_policy_errors():
    import ctypes as C
    from optimal_quad_control_rl_amd import _lib

    L = _lib.load()
    h = C.c_void_p()
    assert L.qr_policy_create(23, 0, C.byref(h)) == _lib.QR_E_INVALID
    assert L.qr_policy_create(24, 0, C.byref(h)) == 0
    assert L.qr_policy_forward(h, 4, C.c_void_p(8), C.c_void_p(8), None) == _lib.QR_E_STATE  # no weights yet
    assert L.qr_policy_destroy(h) == 0


def _make(variant, n, seed=5):
    from optimal_quad_control_rl_amd import (Quadcopter3DGates, Quadcopter3DGatesINDI, TRAIN_DISTURBANCE_RANGES,
                                             square_track, zigzag_track)

    if variant == "e2e":
        env = Quadcopter3DGates(n, *zigzag_track(), gates_ahead=1, seed=seed, infos_mode="none")
        env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
    else:
        env = Quadcopter3DGatesINDI(n, *square_track(), gates_ahead=1, seed=seed, infos_mode="none")
    env.max_steps = 30  # auto-resets inside the window
    env.reset_device()
    return env


def _policy_for(env, gain=20.0):
    from optimal_quad_control_rl_amd.policy import MfmaPolicy
    from optimal_quad_control_rl_amd.ppo import ActorCritic

    torch.manual_seed(3)
    net = ActorCritic(env.state_len, 4).cuda()
    with torch.no_grad():
        net.pi[-1].weight.mul_(gain)  # a policy that actually moves the drone
    return net, MfmaPolicy(env.state_len).load_torch(net.pi)


@pytest.mark.parametrize("variant", ["e2e", "indi"])
@pytest.mark.parametrize("n", [65536, 1000])
@pytest.mark.parametrize("precision", ["f16-operands", "f32"])
def test_closed_loop_rollout_equals_policy_plus_step_launches(variant, n, precision):
    """deterministic closed-loop rollout kernel == K x [policy kernel, clip, step kernel], bit for bit -- with the f16-operand forward
    and (round 6, QR_ROLLOUT_F32CLASS) with the reference-precision forward inside the kernel."""
    K = 48
    net, pol = _policy_for(_make(variant, 8))
    a, b = _make(variant, n), _make(variant, n)
    obs, act, logp, rew, done, trunc, last = a.rollout_policy_device(pol, K, torch.zeros(4), deterministic=True, precision=precision)
    o = b.states_tensor.clone()
    for k in range(K):
        assert torch.equal(obs[k], o), k
        mean = pol.forward(o, precision=precision)
        assert torch.equal(act[k], mean), k
        o2, r2, d2, t2 = b.step_device(mean.clamp(-1, 1).contiguous())
        assert torch.equal(rew[k], r2) and torch.equal(done[k], d2) and torch.equal(trunc[k], t2), k
        o = o2.clone()
    assert torch.equal(last, o)
    for sa, sb in zip(a.get_state_tensors(), b.get_state_tensors()):
        assert sa is None or torch.equal(sa, sb)
    assert done.sum() >= n  # max_steps = 30 inside K = 48


def test_closed_loop_rollout_sampling_statistics():
    n, K = 65536, 8
    env = _make("indi", n)
    net, pol = _policy_for(env, gain=1.0)
    log_std = torch.tensor([0.0, -0.5, 0.3, -1.0])
    obs, act, logp, rew, done, trunc, last = env.rollout_policy_device(pol, K, log_std, noise_seed=11, first_step=1000)
    std = log_std.exp().cuda()
    mean = torch.stack([pol.forward(obs[k].contiguous()) for k in range(K)])
    eps = (act - mean) / std
    assert abs(eps.mean().item()) < 5e-3 and abs(eps.var().item() - 1.0) < 1e-2
    assert abs((eps ** 4).mean().item() - 3.0) < 0.1            # Gaussian kurtosis
    c = torch.corrcoef(eps.reshape(-1, 4).T)
    assert (c - torch.eye(4, device="cuda")).abs().max() < 1e-2  # independent components
    assert abs(torch.corrcoef(torch.stack([eps[0, :, 0], eps[1, :, 0]]))[0, 1].item()) < 2e-2  # and steps
    lp = (-0.5 * eps ** 2).sum(-1) - log_std.sum().item() - 2 * np.log(2 * np.pi)
    assert (lp - logp).abs().max().item() < 2e-3
    # same seed / step offset -> same noise; different offset -> different noise
    env2 = _make("indi", n)
    _, act2, *_ = env2.rollout_policy_device(pol, K, log_std, noise_seed=11, first_step=1000)
    assert torch.equal(act, act2)
    env3 = _make("indi", n)
    _, act3, *_ = env3.rollout_policy_device(pol, K, log_std, noise_seed=11, first_step=2000)
    assert not torch.equal(act[0], act3[0])


def test_ppo_fused_collect_matches_buffer_contract():
    """PPO with the closed-loop collect kernel: buffers are filled consistently (log-probs agree with the torch
    policy at f16 level, values are the torch value net's) and a few iterations run and improve survival."""
    from optimal_quad_control_rl_amd import Quadcopter3DGatesINDI, square_track
    from optimal_quad_control_rl_amd.ppo import PPO

    env = Quadcopter3DGatesINDI(8192, *square_track(), gates_ahead=1, infos_mode="none", seed=1)
    model = PPO(env, n_steps=64, n_epochs=8, batch_size=8192 * 64 // 16, learning_rate=1e-3, seed=0, fused_collect=True)
    model.collect()
    B = 64 * 8192
    with torch.no_grad():
        lp, _ = model.policy.log_prob_entropy(model.buf_obs.view(B, -1), model.buf_act.view(B, 4))
        v = model.policy.value(model.buf_obs.view(B, -1))
    assert (lp - model.buf_lp.view(B)).abs().max().item() < 0.05     # f16-operand policy vs f32 torch policy
    assert (lp - model.buf_lp.view(B)).abs().mean().item() < 2e-3
    assert torch.allclose(v, model.buf_val.view(B), atol=1e-5)
    first = dict(model.stats)
    model.train()
    model.learn(8192 * 64 * 40, log_every=0)
    assert model.stats["ep_len_mean"] > 1.5 * first["ep_len_mean"], (first, model.stats)


def test_ppo_native_update_learns():
    """PPO with the fused collect AND the matrix-core minibatch update (qr_ppo_minibatch): parameters stay finite, the
    torch modules alias the flat parameter vector the kernels update, and a short run improves survival and reward."""
    from optimal_quad_control_rl_amd import Quadcopter3DGatesINDI, square_track
    from optimal_quad_control_rl_amd.ppo import PPO

    env = Quadcopter3DGatesINDI(8192, *square_track(), gates_ahead=1, infos_mode="none", seed=1)
    model = PPO(env, n_steps=32, n_epochs=10, batch_size=8192 * 32 // 16, learning_rate=3e-4, gamma=0.99, seed=1,
                fused_collect=True, native_update=True)
    theta0 = model._updater.theta.clone()
    model.collect()
    model.train()                      # (with the native update, GAE and the episode statistics run here, on the device)
    first = dict(model.stats)
    assert model.stats["updates"] == 160 and model._updater.step == 160
    assert torch.isfinite(model._updater.theta).all() and not torch.equal(model._updater.theta, theta0)
    assert model.policy.pi[0].weight.data_ptr() == model._updater.theta.data_ptr()
    assert 0.0 < model.stats["clip_fraction"] < 0.5 and 0.0 < model.stats["approx_kl"] < 0.1
    model.learn(8192 * 32 * 150, log_every=0)
    # the untrained policy crashes within a few hundred steps (about -10 per ~400 steps); training must improve on that
    assert model.stats["reward_per_step"] > first["reward_per_step"] + 0.005, (first, model.stats)
    assert model.stats["ep_len_mean"] > 450, model.stats
    assert torch.isfinite(model._updater.theta).all()
    # the deterministic policy the kernels trained is what torch's predict() evaluates
    obs = env.reset_device()
    a = model.act_device(obs)
    assert a.shape == (8192, 4) and torch.isfinite(a).all() and float(a.abs().max()) <= 1.0


def test_f32_precision_trainer_collects_through_the_f32class_kernel():
    """precision='f32' of the SB3-shaped PPO (round 6): the collect phase's action means come from qr_policy_forward_f32class, everything
    else from torch float32.  Same seed, same env: the rollout buffers of that trainer and of the plain torch trainer agree at the
    float32 level over a whole rollout with auto-resets (identical done flags; actions within 2e-5; both use the same torch noise)."""
    from optimal_quad_control_rl_amd.ppo import PPO

    def run(policy_forward):
        env = _make("e2e", 2048, seed=11)
        env.max_steps = 10                          # every env is reset (and restarted) inside the 24-step rollout
        torch.manual_seed(123)
        t = PPO(env, n_steps=24, batch_size=2048 * 6, n_epochs=1, seed=3, policy_forward=policy_forward)
        with torch.no_grad():
            t.policy.pi[-1].weight.mul_(20.0)
        torch.manual_seed(777)                      # the sampling noise of both runs
        t.collect()
        out = (t.buf_obs.clone(), t.buf_act.clone(), t.buf_lp.clone(), t.buf_done.clone(), t.buf_rew.clone())
        used = t._mfma is not None
        env.close()
        return out, used

    (o1, a1, l1, d1, r1), used1 = run("torch")
    (o2, a2, l2, d2, r2), used2 = run("f32class")
    assert not used1 and used2
    assert float(d1.sum()) > 0                                                    # auto-resets inside the window
    assert torch.equal(o1[0], o2[0])                                              # same start
    assert 0 < float((a1[0] - a2[0]).abs().max()) <= 2e-5                         # different arithmetic, same precision class
    # a terminated env restarts from a fresh state, so an env whose termination falls on a threshold within float32 noise may part ways:
    # compare the envs whose done flags agree throughout (nearly all), over the whole rollout
    same = (d1 == d2).all(dim=0)
    assert float(same.float().mean()) > 0.995, float(same.float().mean())
    assert float((a1 - a2)[:, same].abs().max()) <= 1e-3 and float((o1 - o2)[:, same].abs().max()) <= 2e-2
    assert float((r1 - r2)[:, same].abs().max()) <= 2e-2 and float((l1 - l2)[:, same].abs().max()) <= 2e-2
