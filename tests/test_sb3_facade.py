"""CPU tests of the SB3-shaped model object (optimal_quad_control_rl_amd.sb3.PPO): the calls the reference's unchanged cells
make on a `stable_baselines3.PPO` -- the `model.predict` unpack of animate_policy (R:803), save / load (R:823, R:3985) and the
attribute walk of the policy -> C export cell (R:3985-3996) -- on a policy-only model (no env, no GPU: training has no CPU path)."""
import os
import zipfile

import numpy as np
import pytest
import torch
import torch.nn as nn

from optimal_quad_control_rl_amd import PPO, VecMonitor

REF_KW = dict(policy_kwargs=dict(activation_fn=torch.nn.ReLU, net_arch=[dict(pi=[120, 120, 120], vf=[120, 120, 120])], log_std_init=0),
              verbose=0, tensorboard_log="logs/E2E", n_steps=1000, batch_size=5000, n_epochs=10, gamma=0.999)   # R:783-795


def make(obs_dim=24, seed=3):
    return PPO("MlpPolicy", None, observation_dim=obs_dim, seed=seed, device="cpu", **REF_KW)


def test_predict_unpacks_like_sb3():
    model = make()
    states = np.random.default_rng(0).normal(size=(10, 24)).astype(np.float32)
    actions, _ = model.predict(states, deterministic=True)            # R:803
    assert isinstance(actions, np.ndarray) and actions.shape == (10, 4) and actions.dtype == np.float32
    assert _ is None and np.all(np.abs(actions) <= 1.0)
    a2, _ = model.predict(states, deterministic=True)
    np.testing.assert_array_equal(actions, a2)
    sampled, _ = model.predict(states, deterministic=False)           # SB3's default
    assert sampled.shape == (10, 4) and not np.array_equal(sampled, actions)
    single, _ = model.predict(states[0], deterministic=True)
    np.testing.assert_allclose(single, actions[0], rtol=1e-5, atol=1e-7)   # batch of 1 vs 10: different sgemm blocking
    assert model.n_steps == 1000 and model.num_timesteps == 0
    with pytest.raises(RuntimeError):
        model.learn(total_timesteps=10, reset_num_timesteps=False, tb_log_name="x")   # no env: no CPU training path


def test_save_load_and_the_export_cells_attribute_walk(tmp_path):
    model = make()
    with torch.no_grad():   # make the head non-trivial (SB3's init gives it gain 0.01)
        model.policy.action_net.weight.mul_(30.0)
        model.policy.log_std.copy_(torch.tensor([-0.5, -0.25, 0.0, 0.25]))
    path = model.save(str(tmp_path / "models" / "E2E" / "test1" / "3000000"))   # R:823: no extension -> '.zip' is appended
    assert path.endswith("3000000.zip") and zipfile.is_zipfile(path)
    with zipfile.ZipFile(path) as z:
        assert {"data", "policy.pth"} <= set(z.namelist())

    loaded = PPO.load(str(tmp_path / "models" / "E2E" / "test1" / "3000000.zip"))   # R:3985
    # R:3988-3996, verbatim apart from the prints
    network = list(loaded.policy.mlp_extractor.policy_net) + [loaded.policy.action_net]
    network = nn.Sequential(*network)
    assert [type(m) for m in network] == [nn.Linear, nn.ReLU, nn.Linear, nn.ReLU, nn.Linear, nn.ReLU, nn.Linear]
    assert [tuple(m.weight.shape) for m in network if isinstance(m, nn.Linear)] == [(120, 24), (120, 120), (120, 120), (4, 120)]
    assert "DiagGaussian" in repr(loaded.policy.action_dist)
    network_std = loaded.policy.log_std.exp().cpu().detach().numpy()
    np.testing.assert_allclose(network_std, np.exp([-0.5, -0.25, 0.0, 0.25]), rtol=1e-6)
    # the walked network IS the policy: same means as predict(deterministic=True) before clipping, on both models
    x = torch.randn(64, 24)
    with torch.no_grad():
        mean = network(x).numpy()
    np.testing.assert_array_equal(np.clip(mean, -1, 1), loaded.predict(x.numpy(), deterministic=True)[0])
    np.testing.assert_array_equal(loaded.predict(x.numpy(), deterministic=True)[0], model.predict(x.numpy(), deterministic=True)[0])
    # SB3's parameter names in policy.pth (what PPO.load of SB3 itself would look for)
    sd = loaded.policy.state_dict()
    for k in ("mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.4.bias", "mlp_extractor.value_net.2.weight",
              "action_net.weight", "value_net.bias", "log_std"):
        assert k in sd
    assert tuple(sd["value_net.weight"].shape) == (1, 120)
    assert loaded.n_steps == 1000 and loaded.batch_size == 5000 and loaded.n_epochs == 10 and loaded.gamma == 0.999
    assert "policy_net" in repr(loaded.policy)


def test_net_arch_forms_and_refusals():
    for arch in ([dict(pi=[120, 120, 120], vf=[120, 120, 120])], dict(pi=[120, 120, 120], vf=[120, 120, 120]), [120, 120, 120]):
        m = PPO("MlpPolicy", None, observation_dim=17, policy_kwargs=dict(activation_fn=nn.ReLU, net_arch=arch), device="cpu")
        assert m.net_arch == (120, 120, 120)
    with pytest.raises(ValueError):
        PPO("MlpPolicy", None, observation_dim=17, policy_kwargs=dict(activation_fn=nn.Tanh), device="cpu")
    with pytest.raises(ValueError):
        PPO("MlpPolicy", None, device="cpu")   # neither env nor observation_dim


def test_vecmonitor_passthrough_reaches_the_env():
    class Env:
        num_envs, state_len = 4, 24
        disturbance_ranges = None

    e = Env()
    w = VecMonitor(e)
    w.venv.disturbance_ranges = np.ones((6, 2))      # R:780
    assert e.disturbance_ranges is not None and w.num_envs == 4


# ---- the reference-precision mode (precision="f32"): on a GPU it runs on hand-written f32-class kernels (tests/test_gpu_policy.py,
# ---- tests/test_gpu_ppo_kernel.py compare them with float64); here, on the CPU, the float32 definition they are held to is pinned ----
def _f10_net():
    """the reference's own generated policy (c_code/neural_network.c weights + 512 rows of its nn_forward, fixture F10) in the
    trainer's network"""
    from optimal_quad_control_rl_amd.ppo import ActorCritic

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f10_policy.npz"))
    net = ActorCritic(24, 4)
    lin = [m for m in net.pi if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for m, k in zip(lin, "1234"):
            m.weight.copy_(torch.as_tensor(d["w" + k]))
            m.bias.copy_(torch.as_tensor(d["b" + k]))
    return net, d


def test_f32_mode_policy_forward_matches_the_reference_nn_forward():
    """<= 1e-5 against c_code/neural_network.c:397-430 nn_forward on F10 for the float32 policy forward in torch -- the
    level precision='f32' is held to (its f32-class matrix-core kernel sits at 7e-7 of nn_forward, the f16-operand kernel at 7e-4:
    tests/test_gpu_policy.py)"""
    net, d = _f10_net()
    with torch.no_grad():
        out = net.pi(torch.as_tensor(d["obs"])).numpy()
    assert np.abs(out - d["mean"]).max() <= 1e-5 * max(1.0, np.abs(d["mean"]).max())


def test_f32_mode_gradient_is_at_float32_level():
    """float32 autograd of the PPO loss agrees with the float64 evaluation of the same loss to <= 1e-4 of each tensor's scale: the
    level the f32-class gradient kernels of precision='f32' are held to against float64 on the GPU (tests/test_gpu_ppo_kernel.py:
    cosine >= 0.999999; the f16-operand kernel: 2-5 percent elementwise, cosine 0.9985-0.999)"""
    net, d = _f10_net()
    g = torch.Generator().manual_seed(0)
    B = 4096
    obs = torch.as_tensor(d["obs"])[torch.randint(0, 512, (B,), generator=g)] + 0.05 * torch.randn(B, 24, generator=g)
    with torch.no_grad():
        act = net.pi(obs) + net.log_std.exp() * torch.randn(B, 4, generator=g)
        old_lp, _ = net.log_prob_entropy(obs, act)
        old_lp = old_lp + 0.05 * torch.randn(B, generator=g)
        adv, ret = torch.randn(B, generator=g), torch.randn(B, generator=g)

    def grads(model, cast):
        o, a, lp0, ad, rt = (t.to(cast) for t in (obs, act, old_lp, adv, ret))
        ad = (ad - ad.mean()) / (ad.std() + 1e-8)
        lp, ent = model.log_prob_entropy(o, a)
        ratio = (lp - lp0).exp()
        pg = -torch.min(ad * ratio, ad * ratio.clamp(0.8, 1.2)).mean()
        loss = pg + 0.5 * torch.nn.functional.mse_loss(model.value(o), rt)
        model.zero_grad()
        loss.backward()
        return [p.grad.detach().double().clone() for p in model.parameters() if p.grad is not None]

    import copy
    g32 = grads(net, torch.float32)
    g64 = grads(copy.deepcopy(net).double(), torch.float64)
    for a32, a64 in zip(g32, g64):
        assert (a32 - a64).abs().max() <= 1e-4 * max(a64.abs().max().item(), 1e-12)


def test_precision_keyword(tmp_path):
    m = PPO("MlpPolicy", None, observation_dim=24, seed=0, device="cpu", precision="f32", **REF_KW)
    assert m.precision == "f32"
    path = m.save(str(tmp_path / "p"))                       # the checkpoint carries it; an explicit keyword wins
    assert PPO.load(path, device="cpu").precision == "f32"
    assert PPO.load(path, device="cpu", precision="f16-operands").precision == "f16-operands"
    with pytest.raises(ValueError):
        PPO("MlpPolicy", None, observation_dim=24, device="cpu", precision="bf16", **REF_KW)
