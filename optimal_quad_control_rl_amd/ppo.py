"""On-device PPO for the MI355X race environment (BASELINE config 5).

The reference trains with stable-baselines3's PPO (`R:783-795`, `I:521-533`): `MlpPolicy` with separate ReLU
networks pi = vf = [120, 120, 120], `log_std_init = 0`, `gamma = 0.999`, `n_steps = 1000`, `batch_size = 5000`,
`n_epochs = 10`, SB3 defaults otherwise (lr 3e-4, GAE lambda 0.95, clip 0.2, vf_coef 0.5, ent_coef 0,
max_grad_norm 0.5, advantage normalisation, orthogonal init, Adam eps 1e-5, actions clipped to the Box).

SB3 is not a dependency here and its NumPy rollout buffer would put a host round trip into every step; this is a
compact torch implementation of the same algorithm that consumes the env's device tensors directly
(`step_device`), so observations, actions, rewards and the rollout buffer never leave HBM.  With tens of thousands
of envs the rollout is short and the minibatches large (`n_steps`, `batch_size`, `n_epochs` are arguments; the
reference's values suit its 100 envs).

Time-limit truncations follow SB3's rule (`collect_rollouts`): the reward of a step that ended by the time limit gets
gamma * V(terminal observation) added, the episode still ends there.  The env kernels hand out the TRUE terminal
observation (final state before the auto-reset, `set_terminal_obs_buffer`); the reference fills
`infos[i]["terminal_observation"]` after `reset_()` (R:589-594), so its SB3 run bootstraps from the first observation of
the next episode -- that quirk is not reproduced (`truncation_bootstrap=False` gives the round-1 behaviour: truncation =
termination).  `target_kl` early stopping is SB3's: checked per minibatch before the optimiser step; with the matrix-core
updater the decision is taken on the device (no host round trip) and, data-parallel, on the all-reduced KL.
"""
import math
import time

import torch
import torch.distributed
from torch import nn


def _mlp(sizes, out_gain):
    layers = []
    for i in range(len(sizes) - 2):
        lin = nn.Linear(sizes[i], sizes[i + 1])
        nn.init.orthogonal_(lin.weight, gain=math.sqrt(2.0))
        nn.init.zeros_(lin.bias)
        layers += [lin, nn.ReLU()]
    head = nn.Linear(sizes[-2], sizes[-1])
    nn.init.orthogonal_(head.weight, gain=out_gain)
    nn.init.zeros_(head.bias)
    layers.append(head)
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    """SB3 `MlpPolicy` with `net_arch=dict(pi=[...], vf=[...])`: two separate MLPs and a state-independent log-std."""

    def __init__(self, obs_dim, act_dim, net_arch=(120, 120, 120), log_std_init=0.0):
        super().__init__()
        self.pi = _mlp([obs_dim, *net_arch, act_dim], 0.01)
        self.vf = _mlp([obs_dim, *net_arch, 1], 1.0)
        self.log_std = nn.Parameter(torch.full((act_dim,), float(log_std_init)))

    def value(self, obs):
        return self.vf(obs).squeeze(-1)

    def log_prob_entropy(self, obs, actions):
        mean = self.pi(obs)
        std = self.log_std.exp()
        lp = (-0.5 * ((actions - mean) / std) ** 2 - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        ent = (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std).sum().expand(obs.shape[0])
        return lp, ent

    @torch.no_grad()
    def act(self, obs, deterministic=False, mean=None):
        """`mean`: action means already evaluated elsewhere (the f32-class matrix-core forward, policy.MfmaPolicy.forward(precision="f32"))."""
        if mean is None:
            mean = self.pi(obs)
        if deterministic:
            return mean, None, None
        std = self.log_std.exp()
        actions = mean + std * torch.randn_like(mean)
        lp = (-0.5 * ((actions - mean) / std) ** 2 - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        return actions, lp, self.vf(obs).squeeze(-1)


def average_across_ranks(t):
    """In-place mean of a tensor over all ranks of the default process group (RCCL all-reduce on GPU tensors; the only
    collective of data-parallel training: one flat gradient of ~63 k floats per minibatch)."""
    torch.distributed.all_reduce(t)
    t /= torch.distributed.get_world_size()
    return t


class MfmaPpoUpdater:
    """PPO minibatch updates on the matrix cores (`qr_ppo_*`, csrc/quadrace_ppo.hip) for an `ActorCritic` with
    net_arch (120, 120, 120).  The module's parameters are re-pointed at views of ONE flat float32 vector (the layout
    the C ABI defines), so torch code that evaluates the networks keeps seeing the current weights."""

    force_data_parallel = False   # test hook: take the all-reduce path on a single rank

    def __init__(self, policy, obs_len, device, max_minibatch, betas=(0.9, 0.999), eps=1e-5, flags=0, precision="f16-operands"):
        if precision not in ("f16-operands", "f32"):
            raise ValueError("precision must be 'f16-operands' or 'f32'")
        self.precision = precision   # "f32": grad() / minibatch() run the reference-precision gradient kernels (qr_ppo_grad_f32class)
        import ctypes as C

        from . import _lib

        self._C, self._lib = C, _lib
        self._L = _lib.load()
        self._h = None
        self.device = device
        h = C.c_void_p()
        # (capacity in whole groups of 64 rows; a minibatch itself may be any size >= 64: the kernel masks a partial last group)
        # kernel forms: `flags` = QR_PPO_* of include/quadrace.h (1: f32 partials, 8: no epoch graph); nothing is read from the environment
        _lib.check(self._L.qr_ppo_create_ex(int(obs_len), int(device.index or 0), (int(max_minibatch) + 63) // 64 * 64, int(flags), C.byref(h)))
        self._h = h
        n = self._L.qr_ppo_num_params(self._h)
        self.theta = torch.zeros(n, dtype=torch.float32, device=device)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.stats = torch.zeros(4, dtype=torch.float32, device=device)
        self.betas, self.eps = betas, eps
        off = 0
        params = []
        for net in (policy.pi, policy.vf):
            for lin in [m for m in net if isinstance(m, nn.Linear)]:
                params += [lin.weight, lin.bias]
        params.append(policy.log_std)
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.theta[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.theta[off:off + k].view(p.shape)
                off += k
        assert off == n, (off, n)
        self.pack()

    def close(self):
        if self._h is not None:
            self._L.qr_ppo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _p(self, t):
        return self._C.c_void_p(t.data_ptr()) if t is not None else None

    @property
    def step(self):
        """Optimiser steps really taken (device-resident count: launches turned into no-ops by the early stop or a non-finite
        gradient norm do not advance it, like torch.optim.Adam under SB3).  Reading it synchronises."""
        v = self._C.c_int32(0)
        self._lib.check(self._L.qr_ppo_adam_step(self._h, self._C.byref(v), 0, self._stream()))
        return int(v.value)

    @step.setter
    def step(self, value):
        v = self._C.c_int32(int(value))
        self._lib.check(self._L.qr_ppo_adam_step(self._h, self._C.byref(v), 1, self._stream()))

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def pack(self):
        """Rebuild the f16 operand images (after theta was changed by anything but `minibatch`)."""
        self._lib.check(self._L.qr_ppo_pack(self._h, self._p(self.theta), self._stream()))

    @staticmethod
    def _check(obs, act, old_lp, adv, ret, idx):
        for t in (obs, act, old_lp, adv, ret):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()

    def grad(self, obs, act, old_lp, adv, ret, idx, clip=0.2, vf_coef=0.5, ent_coef=0.0, stats=False, precision=None, out=None):
        """Flat gradient of the PPO loss on the rows `idx` (no clipping, no optimiser step).  precision="f32": the reference-precision
        kernels (qr_ppo_grad_f32class: every GEMM operand as three bf16 pieces, replayed as one cached graph per distinct argument
        set -- pass the same `out` buffer from call to call to hit it); default = this updater's `precision`."""
        self._check(obs, act, old_lp, adv, ret, idx)
        g = out if out is not None else torch.empty(self.theta.numel() + 4, dtype=torch.float32, device=self.device)  # gradient + minibatch statistics
        fn = self._L.qr_ppo_grad_f32class if (precision or self.precision) == "f32" else self._L.qr_ppo_grad
        self._lib.check(fn(self._h, self._p(self.theta), self._p(obs), self._p(act), self._p(old_lp), self._p(adv),
                                            self._p(ret), self._p(idx), int(idx.numel()), clip, vf_coef, ent_coef, self._p(g),
                                            self._p(self.stats) if stats else None, self._stream()))
        return g

    @staticmethod
    def data_parallel():
        """True when minibatch updates must be synchronised across ranks (torch.distributed initialised with more than one
        rank; the class attribute `force_data_parallel` takes the same code path on a single rank, for testing)."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return False
        return torch.distributed.get_world_size() > 1 or MfmaPpoUpdater.force_data_parallel

    def broadcast_parameters(self, src=0):
        """Make every rank start from rank `src`'s parameters and optimiser state."""
        for t in (self.theta, self.m, self.v):
            torch.distributed.broadcast(t, src)
        self.pack()

    def forward(self, net, obs):
        """Forward pass on the matrix cores with the current operand images: net 0 -> action means [n, 4]; net 1 -> values [n]."""
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous()
        out = torch.empty((obs.shape[0], 4), dtype=torch.float32, device=obs.device)
        self._lib.check(self._L.qr_ppo_forward(self._h, int(net), int(obs.shape[0]), self._p(obs), self._p(out), self._stream()))
        return out if net == 0 else out[:, 0]

    def gae(self, rew, done, val, last_val, gamma, lam, ep_state=None, fin=None, term_val=None, out=None):
        """GAE(lambda) over a rollout [T, N] in one kernel -> (advantages, returns); `term_val` [T, N] = V(terminal obs) at
        time-limit truncations (0 elsewhere; SB3's bootstrap); `ep_state` = (ep_ret, ep_len, ep_gates) running per-env episode
        statistics (updated in place), sums over finished episodes accumulate into `fin` [4]."""
        T, N = rew.shape
        for t in (rew, done, val, last_val) + ((term_val,) if term_val is not None else ()):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        adv, ret = out if out is not None else (torch.empty_like(rew), torch.empty_like(rew))
        er, el, eg = ep_state if ep_state is not None else (None, None, None)
        self._lib.check(self._L.qr_ppo_gae(self._h, T, N, self._p(rew), self._p(done), self._p(val), self._p(last_val),
                                           self._p(term_val), gamma, lam, self._p(adv), self._p(ret), self._p(er), self._p(el),
                                           self._p(eg), self._p(fin), self._stream()))
        return adv, ret

    def control(self, target_kl=None, clear=True):
        """Arms SB3's target-KL early stop (None / <= 0: off) and, with `clear`, resets the sticky stop flag and the
        update counters -- call at the start of every train()."""
        self._lib.check(self._L.qr_ppo_control(self._h, float(target_kl or 0.0), int(bool(clear)), self._stream()))

    def status(self):
        """(stopped, optimiser steps taken, updates skipped for a non-finite gradient, barrier timeouts); synchronises."""
        out = (self._C.c_int32 * 4)()
        self._lib.check(self._L.qr_ppo_status(self._h, out, self._stream()))
        return bool(out[0]), int(out[1]), int(out[2]), int(out[3])

    def begin_epoch(self, adv, perm, B):
        """One launch for the advantage mean / std sums of every minibatch perm[k B:(k + 1) B] of the epoch."""
        assert perm.is_cuda and perm.dtype == torch.int32 and perm.is_contiguous() and perm.numel() % B == 0
        self._lib.check(self._L.qr_ppo_epoch_begin(self._h, self._p(adv), self._p(perm), int(B), perm.numel() // int(B),
                                                   self._stream()))

    def apply(self, grad, lr, B, max_grad_norm=0.5, stats=True):
        """Clip `grad` ([n + 4]: flat gradient + minibatch statistics, as returned by grad()) to the global norm and take one
        Adam step -- the second half of a data-parallel update: `g = up.grad(...); dist.all_reduce(g); g /= world;
        up.apply(g, lr, B)`.  The target-KL decision uses the (averaged) KL sum carried in g."""
        assert grad.is_cuda and grad.dtype == torch.float32 and grad.is_contiguous() and grad.numel() == self.theta.numel() + 4
        self._lib.check(self._L.qr_ppo_apply(self._h, self._p(self.theta), self._p(self.m), self._p(self.v), self._p(grad), int(B),
                                             max_grad_norm, lr, self.betas[0], self.betas[1], self.eps, 0,   # 0: device step count
                                             self._p(self.stats) if stats else None, self._stream()))

    def minibatch(self, obs, act, old_lp, adv, ret, idx, lr, clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5):
        if self.precision == "f32":   # reference-precision gradient kernels, then the (f32) apply kernel; averaged across ranks when data parallel
            if getattr(self, "_g32", None) is None:
                self._g32 = torch.empty(self.theta.numel() + 4, dtype=torch.float32, device=self.device)
            g = self.grad(obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef, stats=False, out=self._g32)
            return self.apply(average_across_ranks(g) if self.data_parallel() else g, lr, int(idx.numel()), max_grad_norm)
        if self.data_parallel():
            # data parallel: every rank holds the same parameters and its own envs; average the gradient (and the minibatch
            # statistics that ride behind it: every rank then takes the same target-KL decision)
            g = self.grad(obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef, stats=False)
            return self.apply(average_across_ranks(g), lr, int(idx.numel()), max_grad_norm)
        self._lib.check(self._L.qr_ppo_minibatch(self._h, self._p(self.theta), self._p(self.m), self._p(self.v), self._p(obs),
                                                 self._p(act), self._p(old_lp), self._p(adv), self._p(ret), self._p(idx),
                                                 int(idx.numel()), clip, vf_coef, ent_coef, max_grad_norm, lr, self.betas[0],
                                                 self.betas[1], self.eps, 0, self._p(self.stats), self._stream()))

    def epoch(self, obs, act, old_lp, adv, ret, perm, B, lr, clip=0.2, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, num_epochs=1,
              device_shuffle=False):
        """`num_epochs` whole epochs -- every minibatch perm[k B:(k + 1) B] in order -- as ONE replayed graph launch
        (qr_ppo_epoch).  `perm` must be the SAME int32 CUDA tensor from call to call: the graph's nodes address it.  Without
        `device_shuffle` the caller rewrites its content in place with a new permutation per epoch (num_epochs = 1); with it the
        library fills it with a fresh keyed permutation at the start of every epoch (set_shuffle / shuffle_state)."""
        self._check(obs, act, old_lp, adv, ret, perm)
        assert perm.numel() % int(B) == 0
        self._lib.check(self._L.qr_ppo_epoch(self._h, self._p(self.theta), self._p(self.m), self._p(self.v), self._p(obs), self._p(act),
                                             self._p(old_lp), self._p(adv), self._p(ret), self._p(perm), int(B), perm.numel() // int(B),
                                             int(num_epochs), int(bool(device_shuffle)), clip, vf_coef, ent_coef, max_grad_norm,
                                             float(lr), self.betas[0], self.betas[1], self.eps, self._p(self.stats), self._stream()))

    def shuffle_state(self):
        """(seed, epochs shuffled so far) of the on-device permutations; synchronises."""
        st = (self._C.c_uint64 * 2)()
        self._lib.check(self._L.qr_ppo_shuffle_state(self._h, st, 0, self._stream()))
        return int(st[0]), int(st[1])

    def set_shuffle(self, seed, count=0):
        st = (self._C.c_uint64 * 2)(int(seed) & (2 ** 64 - 1), int(count))
        self._lib.check(self._L.qr_ppo_shuffle_state(self._h, st, 1, self._stream()))


class PPO:
    def __init__(self, env, n_steps=32, batch_size=None, n_epochs=5, gamma=0.999, gae_lambda=0.95, clip_range=0.2,
                 learning_rate=3e-4, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5, net_arch=(120, 120, 120),
                 log_std_init=0.0, seed=0, target_kl=None, lr_final_frac=1.0, total_timesteps_hint=None,
                 fused_collect=False, native_update=False, truncation_bootstrap=True, policy_forward="torch", update_precision="f16-operands"):
        self.env = env
        self.n_envs, self.dev = env.num_envs, env.device
        self.n_steps, self.n_epochs = n_steps, n_epochs
        self.batch_size = batch_size or (self.n_envs * n_steps) // 4
        self.gamma, self.lam, self.clip = gamma, gae_lambda, clip_range
        self.vf_coef, self.ent_coef, self.max_grad_norm = vf_coef, ent_coef, max_grad_norm
        self.target_kl, self.lr0, self.lr_final_frac, self.total_hint = target_kl, learning_rate, lr_final_frac, total_timesteps_hint
        torch.manual_seed(seed)
        # minibatch permutations come from the trainer's OWN generator (checkpointed; independent of whatever else draws from
        # torch's global generators in the process)
        self._gen = torch.Generator(device=self.dev)
        self._gen.manual_seed(int(seed))
        obs_dim = env.state_len
        self.policy = ActorCritic(obs_dim, 4, net_arch, log_std_init).to(self.dev)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=learning_rate, eps=1e-5)
        T, N = n_steps, self.n_envs
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.buf_obs = torch.empty((T, N, obs_dim), **f32)
        self.buf_act = torch.empty((T, N, 4), **f32)
        self.buf_lp = torch.empty((T, N), **f32)
        self.buf_val = torch.empty((T, N), **f32)
        self.buf_rew = torch.empty((T, N), **f32)
        self.buf_done = torch.empty((T, N), **f32)
        # SB3's time-limit bootstrap: V(terminal observation) at truncated steps (0 elsewhere), added to the reward x gamma
        self.truncation_bootstrap = bool(truncation_bootstrap)
        self.buf_term_val = torch.zeros((T, N), **f32) if self.truncation_bootstrap else None
        self._term_obs = None
        self._done_u8 = torch.empty((T, N), dtype=torch.uint8, device=self.dev)
        self._trunc_u8 = torch.empty((T, N), dtype=torch.uint8, device=self.dev)
        self.num_timesteps = 0
        self._kl_first_trips, self._kl_lr_scale = 0, 1.0
        self.obs = env.reset_device().clone()
        # on-device episode statistics (what VecMonitor provides in the reference, R:769)
        self.ep_ret = torch.zeros(N, **f32)
        self.ep_len = torch.zeros(N, **f32)
        self.ep_gates = torch.zeros(N, **f32)
        self.stats = {}
        # fused_collect: the whole collect phase is ONE kernel (qr_rollout_policy): MFMA policy (f16 operands) +
        # Gaussian sampling + env step; values (and nothing else) are evaluated by torch afterwards in one batch.
        # native_update: every minibatch update (forward, loss, backward, grad-norm clip, Adam) runs in the hand-written
        # matrix-core kernels of csrc/quadrace_ppo.hip instead of torch autograd + torch.optim.Adam
        self.native_update = native_update
        self._updater = None
        if native_update:
            assert tuple(net_arch) == (120, 120, 120), "the matrix-core update is built for the reference's 3 x 120 networks"
            assert self.batch_size >= (2 if update_precision == "f32" else 64) and (T * N) % self.batch_size == 0
            # update_precision="f32": the reference-precision gradient kernels (qr_ppo_grad_f32class, three bf16 pieces per GEMM operand) + the
            # f32 apply kernel, one minibatch at a time (no epoch graph); the collect phase's values then come from the f32-class forward
            # kernel too (_value_f32class below) when policy_forward="f32class", from torch float32 otherwise
            self._updater = MfmaPpoUpdater(self.policy, obs_dim, self.dev, self.batch_size, precision=update_precision)
            self._updater.set_shuffle(0x5EED0000 + int(seed))   # on-device epoch permutations (single-process native update)
        self.fused_collect = fused_collect
        self.noise_seed = seed
        # position in the action-noise stream, in env steps: counts every step ever collected and is NOT reset with num_timesteps
        # (SB3's learn(reset_num_timesteps=True) resets its counter, not its RNG: a second learn() must not replay the first one's
        # noise -- ADVICE r03)
        self.noise_step = 0
        self._mfma = None
        self._mfma_vf = None
        # policy_forward="f32class" (per-step collection only): the action means of collect() come from the hand-written
        # reference-precision forward (qr_policy_forward_f32class: every operand as two f16 pieces, float32-class results) instead of
        # torch's per-layer kernels; sampling, log-probabilities and values stay torch float32.  The weights are re-packed once per rollout.
        if policy_forward not in ("torch", "f32class"):
            raise ValueError("policy_forward must be 'torch' or 'f32class'")
        self.policy_forward = policy_forward   # with fused_collect: the precision of the forward INSIDE the closed-loop kernel
        if fused_collect or self.policy_forward == "f32class":
            from .policy import MfmaPolicy

            assert tuple(net_arch) == (120, 120, 120), "the matrix-core policy kernels are built for the reference's 3 x 120 network"
            self._mfma = MfmaPolicy(obs_dim, self.dev.index)
            self._last_obs = None
            # reference precision end to end: the VALUE network of the fused collect phase through the same f32-class forward kernel (its one
            # output row padded to the kernel's four: the value is column 0), so that no network is evaluated by torch in that mode
            self._mfma_vf = (MfmaPolicy(obs_dim, self.dev.index)
                             if (self.policy_forward == "f32class" and fused_collect and self._updater is not None and self._updater.precision == "f32") else None)
        if self.truncation_bootstrap:  # the kernels write the pre-reset observation of finished envs here
            self._term_obs = torch.zeros((T, N, obs_dim) if fused_collect else (N, obs_dim), **f32)
            env.set_terminal_obs_buffer(self._term_obs)

    @torch.no_grad()
    def _value_f32class_loader(self):
        """Load the current value network into the second policy-kernel handle (last layer [1, 120] zero-padded to [4, 120]) and return
        obs [n, L] -> V [n] through qr_policy_forward_f32class (float32-class: 7e-7 of a float32 evaluation)."""
        lin = [m for m in self.policy.vf if isinstance(m, nn.Linear)]
        w4 = torch.zeros((4, 120), dtype=torch.float32, device=lin[3].weight.device); w4[0] = lin[3].weight[0]
        b4 = torch.zeros(4, dtype=torch.float32, device=lin[3].bias.device); b4[0] = lin[3].bias[0]
        self._mfma_vf.set_weights([(lin[0].weight, lin[0].bias), (lin[1].weight, lin[1].bias), (lin[2].weight, lin[2].bias), (w4, b4)])
        return lambda o: self._mfma_vf.forward(o.contiguous(), precision="f32")[:, 0].contiguous()

    @torch.no_grad()
    def _episode_stats(self, rew, done):
        fin = torch.zeros(4, dtype=torch.float32, device=self.dev)
        for t in range(rew.shape[0]):
            d = done[t]
            self.ep_ret += rew[t]
            self.ep_len += 1.0
            self.ep_gates += (rew[t] > 5.0).to(torch.float32)  # gate reward 10 - 10*d2g (R:537)
            fin += torch.stack([(self.ep_ret * d).sum(), (self.ep_len * d).sum(), (self.ep_gates * d).sum(), d.sum()])
            keep = 1.0 - d
            self.ep_ret *= keep
            self.ep_len *= keep
            self.ep_gates *= keep
        f = fin.tolist()
        if f[3] > 0:
            self.stats.update(ep_rew_mean=f[0] / f[3], ep_len_mean=f[1] / f[3], gates_per_episode=f[2] / f[3], episodes=f[3])
        self.stats["reward_per_step"] = float(rew.mean())

    @torch.no_grad()
    def collect_fused(self):
        self._mfma.load_torch(self.policy.pi)
        first_step = self.noise_step
        self.noise_step += self.n_steps
        obs, act, logp, rew, done, trunc, last_obs = self.env.rollout_policy_device(
            self._mfma, self.n_steps, self.policy.log_std, noise_seed=self.noise_seed, first_step=first_step,
            out=(self.buf_obs, self.buf_act, self.buf_lp, self.buf_rew, self._done_u8, self._trunc_u8),
            precision="f32" if self.policy_forward == "f32class" else "f16-operands")
        self.buf_done.copy_(done)
        T, N = self.n_steps, self.n_envs
        self.num_timesteps += T * N
        f16_values = self._updater is not None and self._updater.precision != "f32"
        if f16_values:
            value = lambda o: self._updater.forward(1, o.contiguous()).contiguous()
        elif self._mfma_vf is not None:
            value = self._value_f32class_loader()
        else:
            value = self.policy.value
        if self.truncation_bootstrap:
            # rows that ended by the time limit (rare: at most one per env per max_steps): V of their terminal observation
            self.buf_term_val.zero_()
            rows = trunc.view(-1).nonzero().squeeze(1)
            self.stats["truncations"] = int(rows.numel())
            if rows.numel():
                self.buf_term_val.view(-1)[rows] = value(self._term_obs.view(T * N, -1)[rows])
        if self._updater is not None:   # values on the matrix cores too; GAE and episode statistics follow in train()
            self.buf_val.copy_(value(self.buf_obs.view(T * N, -1)).view(T, N))
            self.last_val = value(last_obs)
            self._stats_pending = True
            return
        self.buf_val.copy_(value(self.buf_obs.view(T * N, -1)).view(T, N))
        self.last_val = value(last_obs)
        # the torch update evaluates log-probs with the f32 network: store the OLD log-probs from the same network, not the
        # f16 matrix-core ones of the rollout kernel, so that the ratio is exactly 1 at the first minibatch
        lp, _ = self.policy.log_prob_entropy(self.buf_obs.view(T * N, -1), self.buf_act.view(T * N, 4))
        self.buf_lp.copy_(lp.view(T, N))
        self._episode_stats(self.buf_rew, self.buf_done)

    @torch.no_grad()
    def collect(self):
        if self.fused_collect:
            return self.collect_fused()
        fin_ret = fin_len = fin_gates = fin_n = 0.0
        f32class = self.policy_forward == "f32class"
        if f32class:
            self._mfma.load_torch(self.policy.pi)
        for t in range(self.n_steps):
            actions, lp, val = self.policy.act(self.obs, mean=self._mfma.forward(self.obs.contiguous(), precision="f32") if f32class else None)
            self.buf_obs[t].copy_(self.obs)
            self.buf_act[t].copy_(actions)
            self.buf_lp[t].copy_(lp)
            self.buf_val[t].copy_(val)
            obs, rew, done, trunc = self.env.step_device(actions.clamp(-1.0, 1.0).contiguous())  # SB3 clips to the Box
            d = done.to(torch.float32)
            self.buf_rew[t].copy_(rew)
            self.buf_done[t].copy_(d)
            if self.truncation_bootstrap:  # rows of envs that did not finish hold stale data: masked by trunc
                self.buf_term_val[t].copy_(torch.where(trunc.bool(), torch.nan_to_num(self.policy.value(self._term_obs)), 0.0))
            self.ep_ret += rew
            self.ep_len += 1.0
            self.ep_gates += (rew > 5.0).to(torch.float32)  # gate reward 10 - 10*d2g (R:537)
            fin_ret = fin_ret + (self.ep_ret * d).sum()
            fin_len = fin_len + (self.ep_len * d).sum()
            fin_gates = fin_gates + (self.ep_gates * d).sum()
            fin_n = fin_n + d.sum()
            keep = 1.0 - d
            self.ep_ret *= keep
            self.ep_len *= keep
            self.ep_gates *= keep
            self.obs = obs.clone()
        self.last_val = self.policy.value(self.obs)
        self.num_timesteps += self.n_steps * self.n_envs
        n = float(fin_n)
        if n > 0:
            self.stats.update(ep_rew_mean=float(fin_ret) / n, ep_len_mean=float(fin_len) / n,
                              gates_per_episode=float(fin_gates) / n, episodes=n)
        self.stats["reward_per_step"] = float(self.buf_rew.mean())

    @torch.no_grad()
    def _gae(self):
        T = self.n_steps
        adv = torch.empty_like(self.buf_rew)
        last = torch.zeros(self.n_envs, dtype=torch.float32, device=self.dev)
        next_val = self.last_val
        for t in reversed(range(T)):
            nonterminal = 1.0 - self.buf_done[t]
            rew = self.buf_rew[t] if self.buf_term_val is None else self.buf_rew[t] + self.gamma * self.buf_term_val[t]
            delta = rew + self.gamma * next_val * nonterminal - self.buf_val[t]
            last = delta + self.gamma * self.lam * nonterminal * last
            adv[t] = last
            next_val = self.buf_val[t]
        return adv, adv + self.buf_val

    @torch.no_grad()
    def _sanitise_buffers(self):
        """An env whose state went NaN lives until max_steps (reference behaviour); its rows must not
        poison a minibatch: zero them (their advantage is zeroed in train())."""
        bad = ~(torch.isfinite(self.buf_obs).all(-1) & torch.isfinite(self.buf_act).all(-1) &
                torch.isfinite(self.buf_lp) & torch.isfinite(self.buf_rew) & torch.isfinite(self.buf_val))
        self._bad = bad
        self.stats["non_finite_rows"] = int(bad.sum())
        if self.stats["non_finite_rows"]:
            self.buf_obs[bad] = 0.0
            self.buf_act[bad] = 0.0
            self.buf_lp[bad] = 0.0
            self.buf_rew[bad] = 0.0
            self.buf_val[bad] = 0.0
            self.last_val = torch.nan_to_num(self.last_val)
            if self.buf_term_val is not None:
                self.buf_term_val.nan_to_num_()

    def _gae_native(self):
        fin = torch.zeros(4, dtype=torch.float32, device=self.dev)
        pending = getattr(self, "_stats_pending", False)
        if getattr(self, "_adv_ret", None) is None:   # persistent: the epoch graph's nodes address these buffers
            self._adv_ret = (torch.empty_like(self.buf_rew), torch.empty_like(self.buf_rew))
        adv, ret = self._updater.gae(self.buf_rew, self.buf_done, self.buf_val, self.last_val, self.gamma, self.lam,
                                     (self.ep_ret, self.ep_len, self.ep_gates) if pending else None, fin if pending else None,
                                     term_val=self.buf_term_val, out=self._adv_ret)
        if pending:
            self._stats_pending = False
            f = fin.tolist() + [float(self.buf_rew.mean())]
            if f[3] > 0:
                self.stats.update(ep_rew_mean=f[0] / f[3], ep_len_mean=f[1] / f[3], gates_per_episode=f[2] / f[3], episodes=f[3])
            self.stats["reward_per_step"] = f[4]
        return adv, ret

    def train(self):
        self._sanitise_buffers()
        adv, ret = self._gae_native() if self._updater is not None else self._gae()
        if self.stats["non_finite_rows"]:
            adv = adv.masked_fill(self._bad, 0.0)
            ret = torch.where(self._bad, self.buf_val, ret)
        B = self.n_steps * self.n_envs
        obs = self.buf_obs.view(B, -1)
        act = self.buf_act.view(B, 4)
        old_lp, adv, ret = self.buf_lp.view(B), adv.view(B), ret.view(B)
        losses = []
        # learning rate: SB3's `learning_rate` may be a schedule (linear decay when total_timesteps_hint is given); on top of it the
        # target-KL guard below may have halved it
        frac = min(1.0, self.num_timesteps / float(self.total_hint)) if self.total_hint else 0.0
        for g in self.opt.param_groups:
            g["lr"] = self.lr0 * (1.0 - (1.0 - self.lr_final_frac) * frac) * getattr(self, "_kl_lr_scale", 1.0)
        if self.native_update:
            return self._train_native(obs, act, old_lp.contiguous(), adv.contiguous(), ret.contiguous(), B)
        stop = False
        for _ in range(self.n_epochs):
            if stop:
                break
            perm = torch.randperm(B, device=self.dev, generator=self._gen)
            for s in range(0, B, self.batch_size):
                idx = perm[s:s + self.batch_size]
                a = adv[idx]
                a = (a - a.mean()) / (a.std() + 1e-8)
                lp, ent = self.policy.log_prob_entropy(obs[idx], act[idx])
                log_ratio = lp - old_lp[idx]
                ratio = log_ratio.exp()
                if self.target_kl is not None:  # SB3's early stopping on the approximate KL divergence
                    with torch.no_grad():
                        approx_kl = float(((ratio - 1.0) - log_ratio).mean())
                    if approx_kl > 1.5 * self.target_kl:
                        stop = True
                        break
                pg = -torch.min(a * ratio, a * ratio.clamp(1 - self.clip, 1 + self.clip)).mean()
                v = self.policy.value(obs[idx])
                vl = torch.nn.functional.mse_loss(v, ret[idx])
                loss = pg + self.vf_coef * vl - self.ent_coef * ent.mean()
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
                self.opt.step()
                losses.append(loss.detach())
        self._kl_guard(self.target_kl is not None, stop and len(losses) <= 1)
        if losses:
            self.stats["loss"] = float(torch.stack(losses).mean())
        self.stats["updates"] = self.stats.get("updates", 0) + len(losses)
        self.stats["std"] = float(self.policy.log_std.detach().exp().mean())

    def _kl_guard(self, armed, first_minibatch_tripped, patience=3):
        """SB3's target_kl rule checks the approximate KL BEFORE a minibatch's step, i.e. from the second minibatch on it measures
        what the steps already taken did.  If one step at the current learning rate moves the policy by more than 1.5 target_kl, the
        rule stops every rollout after its FIRST step and the run never recovers (seen once in round 2: profiles/
        r02_ppo_6e9_seed3_stall_trace.txt).  Guard: when a whole train() took at most ONE step because the rule tripped on its
        second minibatch -- or none at all -- `patience` rollouts in a row, the learning rate is halved (down to lr / 64) until
        steps fit under the limit again.  (The reference itself runs with target_kl = None; this only exists for the speed option.)"""
        if not armed:
            return
        self._kl_first_trips = self._kl_first_trips + 1 if first_minibatch_tripped else 0
        if self._kl_first_trips >= patience and getattr(self, "_kl_lr_scale", 1.0) > 1.0 / 64.0:
            self._kl_lr_scale = getattr(self, "_kl_lr_scale", 1.0) * 0.5
            self._kl_first_trips = 0
            self.stats["kl_lr_halvings"] = self.stats.get("kl_lr_halvings", 0) + 1

    def _train_native(self, obs, act, old_lp, adv, ret, B):
        up = self._updater
        lr = self.opt.param_groups[0]["lr"]
        kl_stop = self.target_kl is not None and self.target_kl < 1e8
        # SB3's target_kl rule runs on the device: before each optimiser step the update kernel compares the minibatch's
        # mean approx-KL (data-parallel: the all-reduced one) with 1.5 target_kl; a hit skips that step and turns every
        # later launch of this train() into a no-op -- no host synchronisation inside the loop, identical on every rank
        up.control(self.target_kl if kl_stop else None, clear=True)
        up.stats.zero_()
        # Single process: an epoch = ONE replayed graph (qr_ppo_epoch: statistics launch + gradient / apply kernels of every
        # minibatch); the permutation is rewritten in place in a buffer the graph's nodes address.  Data-parallel: the all-reduce
        # between gradient and apply keeps the launches on the stream.
        if getattr(self, "_perm_buf", None) is None or self._perm_buf.numel() != B:
            self._perm_buf = torch.empty(B, dtype=torch.int32, device=self.dev)
        perm = self._perm_buf
        if not up.data_parallel() and up.precision != "f32":
            # the permutations are drawn on the device too (a keyed bijection per epoch, no sort); without the early stop ALL
            # epochs of this train() are one graph launch, with it one launch per epoch and one status read in between
            hp = (self.batch_size, lr, self.clip, self.vf_coef, self.ent_coef, self.max_grad_norm)
            if not kl_stop:
                up.epoch(obs, act, old_lp, adv, ret, perm, *hp, num_epochs=self.n_epochs, device_shuffle=True)
            else:
                for _ in range(self.n_epochs):
                    up.epoch(obs, act, old_lp, adv, ret, perm, *hp, num_epochs=1, device_shuffle=True)
                    if up.status()[0]:   # after the stop every further launch is a no-op: one ~50 us read per epoch saves them
                        break
        else:
            for _ in range(self.n_epochs):
                perm.copy_(torch.randperm(B, device=self.dev, generator=self._gen))
                if up.precision != "f32":   # (the f32-class gradient kernels form each minibatch's advantage statistics themselves)
                    up.begin_epoch(adv, perm, self.batch_size)
                for s in range(0, B, self.batch_size):
                    up.minibatch(obs, act, old_lp, adv, ret, perm[s:s + self.batch_size], lr, self.clip, self.vf_coef, self.ent_coef,
                                 self.max_grad_norm)
                if kl_stop and up.status()[0]:   # identical on every rank: the KL sum travels with the all-reduced gradient
                    break
        stopped, applied, skipped, timeouts = up.status()
        assert timeouts == 0, "grid barrier of the update kernel timed out"
        st = up.stats.tolist()
        seen = max(1, (applied + skipped + (1 if stopped else 0)) * self.batch_size)
        self.stats["loss"] = (st[0] + self.vf_coef * st[1]) / seen
        self.stats["approx_kl"] = st[2] / seen
        self.stats["clip_fraction"] = st[3] / seen
        self.stats["updates"] = self.stats.get("updates", 0) + applied
        self.stats["early_stop"] = bool(stopped)
        self._kl_guard(kl_stop, bool(stopped) and applied <= 1)
        self.stats["skipped_nonfinite"] = self.stats.get("skipped_nonfinite", 0) + skipped
        self.stats["std"] = float(self.policy.log_std.detach().exp().mean())

    def learn(self, total_timesteps, log_every=10, callback=None):
        t0 = time.perf_counter()
        it = 0
        while self.num_timesteps < total_timesteps:
            self.collect()
            self.train()
            it += 1
            if log_every and it % log_every == 0:
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                s = self.stats
                print(f"it {it:4d}  steps {self.num_timesteps/1e6:8.1f}M  {self.num_timesteps/el/1e6:6.2f} Msteps/s  "
                      f"ep_rew {s.get('ep_rew_mean', float('nan')):8.2f}  ep_len {s.get('ep_len_mean', float('nan')):7.1f}  "
                      f"gates/ep {s.get('gates_per_episode', float('nan')):6.2f}  std {s.get('std', 0):.3f}", flush=True)
            if callback is not None and callback(self) is False:
                break
        return self

    @torch.no_grad()
    def act_device(self, obs, deterministic=True):
        """Device-tensor policy evaluation for evaluation loops that stay on the GPU: actions [n, 4] clipped to the Box, as a CUDA
        tensor.  (The SB3-shaped `predict() -> (numpy actions, None)` of the reference's cells lives on `sb3.PPO`.)"""
        if not isinstance(obs, torch.Tensor):
            obs = torch.as_tensor(obs, dtype=torch.float32, device=self.dev)
        a, _, _ = self.policy.act(obs, deterministic=deterministic)
        return a.clamp(-1.0, 1.0)

    # ------------------------------------------------------------------------------------------------ checkpoints
    def sync_parameters(self):
        """Call after the policy parameters were changed by anything but train() (e.g. a loaded checkpoint)."""
        if self._updater is not None:
            self._updater.pack()

    @torch.no_grad()
    def state_dict(self):
        """Everything beyond the policy parameters that the next collect() / train() depends on: optimiser state, step counters,
        episode accumulators, the env's state (positions in its reset streams included) and the generator of the minibatch
        permutations.  With the same env constructor arguments, load_state_dict() continues the run bit for bit."""
        if self._updater is not None:
            opt = dict(kind="mfma_adam", m=self._updater.m.clone(), v=self._updater.v.clone(), step=int(self._updater.step),
                       shuffle=self._updater.shuffle_state(),
                       betas=tuple(self._updater.betas), eps=float(self._updater.eps), lr=float(self.opt.param_groups[0]["lr"]))
        else:
            opt = dict(kind="torch_adam", state=self.opt.state_dict())
        world, dist, target, steps, episode = self.env.get_state_tensors()
        env = dict(world=world.cpu(), dist=None if dist is None else dist.cpu(), target=target.cpu(), steps=steps.cpu(),
                   episode=episode.cpu())
        return dict(optimizer=opt, num_timesteps=int(self.num_timesteps), ep_ret=self.ep_ret.cpu(), ep_len=self.ep_len.cpu(),
                    ep_gates=self.ep_gates.cpu(), stats=dict(self.stats), noise_seed=int(self.noise_seed), noise_step=int(self.noise_step),
                    lr0=float(self.lr0), lr_final_frac=float(self.lr_final_frac), total_hint=self.total_hint,
                    kl_trips=int(getattr(self, "_kl_first_trips", 0)), kl_lr_scale=float(getattr(self, "_kl_lr_scale", 1.0)), env=env, rng=self._gen.get_state())

    @torch.no_grad()
    def load_state_dict(self, sd):
        opt = sd["optimizer"]
        if self._updater is not None:
            assert opt["kind"] == "mfma_adam", "checkpoint was written by the torch optimiser path"
            self._updater.m.copy_(opt["m"])
            self._updater.v.copy_(opt["v"])
            self._updater.step = int(opt["step"])
            if "shuffle" in opt:
                self._updater.set_shuffle(*opt["shuffle"])
            for g in self.opt.param_groups:
                g["lr"] = float(opt["lr"])
            self._updater.pack()
        else:
            assert opt["kind"] == "torch_adam", "checkpoint was written by the matrix-core update path"
            self.opt.load_state_dict(opt["state"])
        self.num_timesteps = int(sd["num_timesteps"])
        self.ep_ret.copy_(sd["ep_ret"]); self.ep_len.copy_(sd["ep_len"]); self.ep_gates.copy_(sd["ep_gates"])
        self.stats = dict(sd["stats"])
        self.noise_seed = int(sd["noise_seed"])
        self.noise_step = int(sd.get("noise_step", self.num_timesteps // self.n_envs))   # older checkpoints: derived from the step count
        self.lr0, self.lr_final_frac, self.total_hint = float(sd["lr0"]), float(sd["lr_final_frac"]), sd["total_hint"]
        self._kl_first_trips = int(sd.get("kl_trips", 0))
        self._kl_lr_scale = float(sd.get("kl_lr_scale", 1.0))
        e = sd["env"]
        self.env.set_state_tensors(world=e["world"], dist=e["dist"], target=e["target"], steps=e["steps"], episode=e["episode"])
        self.env.update_states()
        self.obs = self.env.states_tensor.clone()
        self._gen.set_state(sd["rng"])
