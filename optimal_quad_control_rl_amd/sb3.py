"""SB3-shaped model object around the on-device PPO (what the reference's training / export cells call).

The reference's notebook cells that stay as they are -- `animate_policy` (R:800-810), the training loop with a checkpoint
every ten rollouts (R:813-831) and the policy -> C export (R:3980-3996) -- talk to a `stable_baselines3.PPO` object:

    model = PPO("MlpPolicy", env, policy_kwargs=dict(activation_fn=torch.nn.ReLU,
                net_arch=[dict(pi=[120,120,120], vf=[120,120,120])], log_std_init=0),
                n_steps=1000, batch_size=5000, n_epochs=10, gamma=0.999)                 # R:783-795
    actions, _ = model.predict(env.states, deterministic=deterministic)                   # R:803
    model.learn(total_timesteps=TIMESTEPS, reset_num_timesteps=False, tb_log_name=...)   # R:820
    model.save(models_dir + '/' + log_name + '/' + str(model.num_timesteps))             # R:823
    model = PPO.load(path)                                                               # R:3985
    network = list(model.policy.mlp_extractor.policy_net) + [model.policy.action_net]    # R:3988
    model.policy.log_std.exp()                                                           # R:3994

`PPO` below offers exactly that surface on top of `ppo.PPO` (the trainer whose rollouts, GAE and minibatch updates run in
libquadrace's kernels): same constructor keywords with SB3's defaults, `predict() -> (numpy actions clipped to the Box,
None)`, `learn(total_timesteps, reset_num_timesteps=...)`, `save()` / `load()` (a zip like SB3's: `data` JSON + `policy.pth`
with SB3's parameter names + `policy.optimizer.pth`, plus this build's trainer state so that a run resumes bit for bit),
`n_steps`, `num_timesteps`, and a `policy` whose `mlp_extractor.policy_net`, `mlp_extractor.value_net`, `action_net`,
`value_net`, `log_std` are the live torch modules / parameters.  A model loaded without an env (or on a box without a GPU)
is a plain torch policy: `predict` and the attribute walk work, `learn` needs an env (there is no CPU training path).
"""
import io
import json
import math
import os
import zipfile

import numpy as np
import torch
from torch import nn

from .ppo import ActorCritic

FORMAT_VERSION = 1


class _MlpExtractor:
    """`model.policy.mlp_extractor`: the hidden stacks of both networks as `nn.Sequential`s that SHARE their Linear modules
    with the trainer's networks (SB3's `MlpExtractor.policy_net` / `.value_net`)."""

    def __init__(self, pi, vf):
        self.policy_net = nn.Sequential(*list(pi)[:-1])
        self.value_net = nn.Sequential(*list(vf)[:-1])

    def forward_actor(self, features):
        return self.policy_net(features)

    def forward_critic(self, features):
        return self.value_net(features)


class _DiagGaussian:
    """what `print(model.policy.action_dist)` shows in the reference cell (R:3992); carries the action dimension"""

    def __init__(self, action_dim):
        self.action_dim = int(action_dim)

    def __repr__(self):
        return f"DiagGaussianDistribution(action_dim={self.action_dim})"


class ActorCriticPolicy:
    """SB3-named view of an `ActorCritic` (same parameter tensors, nothing copied)."""

    def __init__(self, net: ActorCritic):
        self.net = net
        self.action_dist = _DiagGaussian(net.log_std.numel())

    # --- SB3 attribute names
    @property
    def mlp_extractor(self):
        return _MlpExtractor(self.net.pi, self.net.vf)

    @property
    def action_net(self):
        return self.net.pi[-1]

    @property
    def value_net(self):
        return self.net.vf[-1]

    @property
    def log_std(self):
        return self.net.log_std

    @property
    def device(self):
        return self.net.log_std.device

    def parameters(self):
        return self.net.parameters()

    def __repr__(self):
        ex = self.mlp_extractor
        return ("ActorCriticPolicy(\n  (mlp_extractor): MlpExtractor(\n    (policy_net): %r\n    (value_net): %r\n  )\n"
                "  (action_net): %r\n  (value_net): %r\n)" % (ex.policy_net, ex.value_net, self.action_net, self.value_net))

    # --- SB3's state_dict naming: mlp_extractor.policy_net.{0,2,4}.*, mlp_extractor.value_net.{0,2,4}.*, action_net.*,
    # value_net.*, log_std
    def state_dict(self):
        out = {"log_std": self.net.log_std.detach().clone()}
        for name, seq in (("policy_net", self.net.pi), ("value_net", self.net.vf)):
            mods = list(seq)
            for i, m in enumerate(mods[:-1]):
                if isinstance(m, nn.Linear):
                    out[f"mlp_extractor.{name}.{i}.weight"] = m.weight.detach().clone()
                    out[f"mlp_extractor.{name}.{i}.bias"] = m.bias.detach().clone()
            head = "action_net" if name == "policy_net" else "value_net"
            out[f"{head}.weight"] = mods[-1].weight.detach().clone()
            out[f"{head}.bias"] = mods[-1].bias.detach().clone()
        return out

    @torch.no_grad()
    def load_state_dict(self, sd):
        """In-place copies: with the matrix-core updater the parameters are views of ONE flat vector and must stay so."""
        self.net.log_std.copy_(sd["log_std"])
        for name, seq in (("policy_net", self.net.pi), ("value_net", self.net.vf)):
            mods = list(seq)
            for i, m in enumerate(mods[:-1]):
                if isinstance(m, nn.Linear):
                    m.weight.copy_(sd[f"mlp_extractor.{name}.{i}.weight"])
                    m.bias.copy_(sd[f"mlp_extractor.{name}.{i}.bias"])
            head = "action_net" if name == "policy_net" else "value_net"
            mods[-1].weight.copy_(sd[f"{head}.weight"])
            mods[-1].bias.copy_(sd[f"{head}.bias"])

    @torch.no_grad()
    def predict(self, observation, state=None, episode_start=None, deterministic=False):
        dev = self.device
        obs = torch.as_tensor(np.asarray(observation) if not isinstance(observation, torch.Tensor) else observation,
                              dtype=torch.float32, device=dev)
        single = obs.dim() == 1
        if single:
            obs = obs[None]
        mean = self.net.pi(obs)
        a = mean if deterministic else mean + self.net.log_std.exp() * torch.randn_like(mean)
        a = a.clamp(-1.0, 1.0).cpu().numpy()   # SB3 clips to the Box(-1, 1) action space (R:325)
        return (a[0] if single else a), state


def _net_arch(policy_kwargs):
    """pi / vf layer lists out of SB3's `net_arch` forms: [dict(pi=..., vf=...)] (SB3 <= 1.8, the reference's, R:784),
    dict(pi=..., vf=...) (SB3 2.x) or a plain list shared by both."""
    arch = (policy_kwargs or {}).get("net_arch", [dict(pi=[64, 64], vf=[64, 64])])
    if isinstance(arch, (list, tuple)) and len(arch) == 1 and isinstance(arch[0], dict):
        arch = arch[0]
    if isinstance(arch, dict):
        pi, vf = list(arch.get("pi", [])), list(arch.get("vf", []))
    else:
        pi = vf = list(arch)
    if pi != vf:
        raise ValueError("this build trains equal pi / vf architectures (the reference uses [120, 120, 120] for both)")
    return tuple(pi)


def _unwrap(env):
    """the race env inside a VecMonitor-like wrapper (`env.venv`, R:769,780)"""
    seen = 0
    while hasattr(env, "venv") and seen < 8:
        env, seen = env.venv, seen + 1
    return env


class VecMonitor:
    """Pass-through stand-in for `stable_baselines3.common.vec_env.VecMonitor` (R:769): the reference wraps its env in it for
    episode statistics and then reaches through `env.venv` (R:780).  Here the statistics are accumulated on the device by
    the trainer (`model.ep_info`), so the wrapper only forwards."""

    def __init__(self, venv, filename=None, info_keywords=()):
        self.venv = venv

    def __getattr__(self, name):
        return getattr(self.venv, name)


class PPO:
    """See the module docstring.  Keyword defaults are SB3 2.1's; keywords this build adds are marked (+)."""

    def __init__(self, policy="MlpPolicy", env=None, learning_rate=3e-4, n_steps=2048, batch_size=64, n_epochs=10,
                 gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5,
                 target_kl=None, tensorboard_log=None, policy_kwargs=None, verbose=0, seed=None, device="auto",
                 observation_dim=None,          # (+) size of the observation when there is no env (a policy-only model)
                 native_update="auto",          # (+) minibatch updates in libquadrace's matrix-core kernels when the shapes allow
                 fused_collect="auto",          # (+) rollouts as one closed-loop kernel
                 precision=None,                # (+) "f16-operands" (default) | "f32": see below
                 _init_trainer=True):
        # precision: the throughput kernels (closed-loop collection, PPO update) compute with f16 matrix-core operands and f32 accumulation
        # (policy mean within 7e-4 of the reference's float32 nn_forward, gradient cosine >= 0.9985 against float32 autograd).
        # precision="f32" is the REFERENCE-PRECISION mode, hand-written end to end since round 6: the collect phase is one closed-loop
        # kernel with the f32-class policy forward inside (7e-7 against nn_forward), every minibatch update runs in the f32-class
        # gradient kernels (three bf16 pieces per GEMM operand: cosine 1 - 1e-13 against float64 autograd) + the f32 apply kernel; the
        # value estimates of the collect phase come from the same f32-class forward kernel (no network is evaluated by torch).  Several times slower than the default path; for A/B runs that
        # ask whether an outcome is the recipe's or the arithmetic's.
        precision = precision or "f16-operands"
        # precision="f32-collect": the f32-class forward in the collect phase, the f16-operand matrix-core kernels for the update (an A/B leg
        # that isolates the precision of the COLLECTED actions / log-probabilities at full training speed)
        if precision not in ("f16-operands", "f32", "f32-collect"):
            raise ValueError("precision must be 'f16-operands', 'f32' or 'f32-collect'")
        self.precision = precision
        # precision="f32" (round 6): the collect phase stays ONE kernel with the f32-class forward inside, and the update runs in the
        # hand-written reference-precision gradient kernels (qr_ppo_grad_f32class) + the f32 apply kernel -- no torch in the loop
        if policy not in ("MlpPolicy", None):
            raise ValueError("only SB3's 'MlpPolicy' exists here")
        pk = dict(policy_kwargs or {})
        if pk.get("activation_fn", nn.ReLU) is not nn.ReLU:   # (SB3's own default is Tanh; the reference passes ReLU, R:784)
            raise ValueError("the networks of this build are ReLU MLPs (policy_kwargs['activation_fn'] = torch.nn.ReLU, R:784)")
        self.net_arch = _net_arch(pk) if "net_arch" in pk else (64, 64)
        self.log_std_init = float(pk.get("log_std_init", 0.0))
        self.hyper = dict(learning_rate=float(learning_rate), n_steps=int(n_steps), batch_size=int(batch_size),
                          n_epochs=int(n_epochs), gamma=float(gamma), gae_lambda=float(gae_lambda), clip_range=float(clip_range),
                          ent_coef=float(ent_coef), vf_coef=float(vf_coef), max_grad_norm=float(max_grad_norm),
                          target_kl=None if target_kl is None else float(target_kl))
        self.n_steps, self.batch_size, self.n_epochs, self.gamma = int(n_steps), int(batch_size), int(n_epochs), float(gamma)
        self.tensorboard_log, self.verbose = tensorboard_log, verbose
        self.seed = 0 if seed is None else int(seed)
        self.env = env
        self._trainer = None
        self._num_timesteps = 0
        core = _unwrap(env) if env is not None else None
        if core is not None and _init_trainer:
            from .ppo import PPO as Trainer

            rows = core.num_envs * self.n_steps
            # the matrix-core update: the reference's 3 x 120 networks, whole minibatches of >= 64 rows, and at most 4 096 minibatches
            # per epoch (qr_ppo_epoch's limit; all epochs of a train() are one graph of ~2 x minibatches x epochs kernel nodes, kept
            # below 32 768 nodes) -- anything else, e.g. SB3's own defaults batch_size=64 x n_steps=2048 on > 128 envs, takes the
            # torch update on the same device tensors instead of failing inside learn()
            minibatches = rows // self.batch_size if self.batch_size > 0 else 0
            shapes_ok = (tuple(self.net_arch) == (120, 120, 120) and self.batch_size >= (2 if precision == "f32" else 64) and rows % self.batch_size == 0
                         and minibatches <= 4096 and minibatches * self.n_epochs <= 16384)
            native = shapes_ok if native_update == "auto" else bool(native_update)
            fused = (tuple(self.net_arch) == (120, 120, 120)) if fused_collect == "auto" else bool(fused_collect)
            self._trainer = Trainer(core, n_steps=self.n_steps, batch_size=self.batch_size, n_epochs=self.n_epochs,
                                    gamma=self.gamma, gae_lambda=gae_lambda, clip_range=clip_range,
                                    learning_rate=learning_rate, vf_coef=vf_coef, ent_coef=ent_coef,
                                    max_grad_norm=max_grad_norm, net_arch=self.net_arch, log_std_init=self.log_std_init,
                                    seed=self.seed, target_kl=target_kl, fused_collect=fused, native_update=native,
                                    policy_forward="f32class" if (precision in ("f32", "f32-collect") and tuple(self.net_arch) == (120, 120, 120)) else "torch",
                                    update_precision="f32" if precision == "f32" else "f16-operands")
            self._net = self._trainer.policy
            self.observation_dim = int(core.state_len)
        else:
            if core is not None:
                observation_dim = int(core.state_len)
            if observation_dim is None:
                raise ValueError("a model without an env needs observation_dim")
            self.observation_dim = int(observation_dim)
            torch.manual_seed(self.seed)
            dev = torch.device("cuda") if (device in ("auto", "cuda") and torch.cuda.is_available()) else torch.device("cpu")
            self._net = ActorCritic(self.observation_dim, 4, self.net_arch, self.log_std_init).to(dev)
        self.policy = ActorCriticPolicy(self._net)

    # ------------------------------------------------------------------------------------------------ SB3 surface
    @property
    def num_timesteps(self):
        return self._trainer.num_timesteps if self._trainer is not None else self._num_timesteps

    @property
    def device(self):
        return self.policy.device

    @property
    def ep_info(self):
        """episode statistics of the last rollouts (what VecMonitor + `ep_info_buffer` give SB3's logger)"""
        return dict(self._trainer.stats) if self._trainer is not None else {}

    def get_env(self):
        return self.env

    def set_env(self, env):
        if self._trainer is not None:
            raise RuntimeError("this model already trains an env; load the checkpoint with the new env instead")
        fresh = PPO("MlpPolicy", env, policy_kwargs=dict(activation_fn=nn.ReLU, net_arch=dict(pi=list(self.net_arch), vf=list(self.net_arch)),
                                                         log_std_init=self.log_std_init), seed=self.seed, **self.hyper)
        fresh.policy.load_state_dict(self.policy.state_dict())
        fresh._trainer.sync_parameters()
        fresh._trainer.num_timesteps = self._num_timesteps
        self.__dict__.update(fresh.__dict__)

    def predict(self, observation, state=None, episode_start=None, deterministic=False):
        """-> (actions as a NumPy array clipped to the action Box, None), SB3's `BaseAlgorithm.predict` (R:803)."""
        return self.policy.predict(observation, state, episode_start, deterministic)

    def learn(self, total_timesteps, callback=None, log_interval=1, tb_log_name="PPO", reset_num_timesteps=True,
              progress_bar=False):
        """SB3 semantics: with reset_num_timesteps=False (R:820) `total_timesteps` MORE steps are taken; whole rollouts only."""
        if self._trainer is None:
            raise RuntimeError("learn() needs an env on a gfx950 GPU (PPO.load(path, env=...)); there is no CPU training path")
        t = self._trainer
        if reset_num_timesteps:
            t.num_timesteps = 0
        target = t.num_timesteps + int(total_timesteps)
        t.learn(target, log_every=(20 if self.verbose else 0), callback=callback)
        return self

    # ------------------------------------------------------------------------------------------------ checkpoints
    def _data(self):
        return dict(format_version=FORMAT_VERSION, algo="PPO", policy_class="MlpPolicy", observation_dim=self.observation_dim,
                    action_dim=4, net_arch=list(self.net_arch), activation_fn="ReLU", log_std_init=self.log_std_init,
                    num_timesteps=int(self.num_timesteps), seed=self.seed, precision=self.precision, **self.hyper)

    def save(self, path):
        """SB3 appends '.zip' to a path without an extension (R:823 passes none)."""
        path = str(path)
        if not os.path.splitext(path)[1]:
            path += ".zip"
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)

        def blob(obj):
            b = io.BytesIO()
            torch.save(obj, b)
            return b.getvalue()

        cpu = lambda sd: {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}  # noqa: E731
        with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
            z.writestr("data", json.dumps(self._data(), indent=1))
            z.writestr("policy.pth", blob(cpu(self.policy.state_dict())))
            if self._trainer is not None:
                ts = self._trainer.state_dict()
                z.writestr("policy.optimizer.pth", blob(cpu(ts.pop("optimizer"))))
                z.writestr("trainer_state.pth", blob(ts))
            z.writestr("_format", "optimal_quad_control_rl_amd PPO checkpoint v%d (SB3-shaped: data / policy.pth / policy.optimizer.pth)" % FORMAT_VERSION)
        return path

    @classmethod
    def load(cls, path, env=None, device="auto", **kwargs):
        path = str(path)
        if not os.path.exists(path) and os.path.exists(path + ".zip"):
            path += ".zip"
        with zipfile.ZipFile(path) as z:
            data = json.loads(z.read("data"))
            # everything is validated BEFORE a model (and, with an env, its GPU buffers) is built
            if data.get("format_version") != FORMAT_VERSION:
                raise ValueError("checkpoint format version %r, this build reads %d" % (data.get("format_version"), FORMAT_VERSION))
            if env is not None and _unwrap(env).state_len != data["observation_dim"]:
                raise ValueError("checkpoint and env disagree on the observation length")
            names = set(z.namelist())
            # the members hold tensors, dicts, tuples and scalars only: the restricted unpickler is enough, and a downloaded
            # checkpoint cannot run code
            rd = lambda n: torch.load(io.BytesIO(z.read(n)), map_location="cpu", weights_only=True)  # noqa: E731
            policy_sd = rd("policy.pth")
            opt = rd("policy.optimizer.pth") if "policy.optimizer.pth" in names else None
            trainer_state = rd("trainer_state.pth") if "trainer_state.pth" in names else None
        hyper = {k: data[k] for k in ("learning_rate", "n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range",
                                      "ent_coef", "vf_coef", "max_grad_norm", "target_kl")}
        kwargs = dict(kwargs)
        seed = kwargs.pop("seed", data["seed"])           # explicit keywords win over the checkpoint's, without colliding with it
        device = kwargs.pop("device", device)
        # a run saved in the reference-precision mode resumes in it (checkpoints written before round 6 carry no entry: the default path)
        kwargs.setdefault("precision", data.get("precision", "f16-operands"))
        hyper.update(kwargs)
        pk = dict(activation_fn=nn.ReLU, net_arch=dict(pi=data["net_arch"], vf=data["net_arch"]), log_std_init=data["log_std_init"])
        model = cls("MlpPolicy", env, policy_kwargs=pk, seed=seed, device=device,
                    observation_dim=data["observation_dim"], **hyper)
        model.policy.load_state_dict({k: v.to(model.device) for k, v in policy_sd.items()})
        model._num_timesteps = int(data["num_timesteps"])
        if model._trainer is not None:
            model._trainer.sync_parameters()
            model._trainer.num_timesteps = int(data["num_timesteps"])
            if trainer_state is not None and opt is not None:
                trainer_state["optimizer"] = opt
                model._trainer.load_state_dict(trainer_state)
        return model
