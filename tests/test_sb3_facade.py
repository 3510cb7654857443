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
