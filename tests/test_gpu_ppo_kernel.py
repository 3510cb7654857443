"""GPU: the matrix-core PPO update (qr_ppo_*, csrc/quadrace_ppo.hip) against torch autograd / torch.optim.Adam on the
same minibatch.  Forward and backward GEMMs use f16 operands (f32 accumulation), so gradients agree to the f16 operand
level (a few 1e-3 relative per tensor); the optimiser arithmetic is f32 and is compared tightly."""
import copy
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(L, rows, seed=0, max_minibatch=4096, flags=0):
    from optimal_quad_control_rl_amd.ppo import ActorCritic, MfmaPpoUpdater

    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    pol = ActorCritic(L, 4).to(dev)
    with torch.no_grad():   # non-trivial biases / output layer / log_std so every gradient path is exercised
        for p in pol.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        pol.pi[-1].weight.mul_(30.0)
        pol.log_std.copy_(torch.tensor([-0.3, 0.1, -0.5, 0.2], device=dev))
    ref = copy.deepcopy(pol)
    up = MfmaPpoUpdater(pol, L, dev, max_minibatch=max_minibatch, flags=flags)   # flags: QR_PPO_* of include/quadrace.h (1 = f32 partials)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    obs = torch.randn((rows, L), device=dev, generator=g) * 1.5
    with torch.no_grad():
        mean = ref.pi(obs)
        act = mean + ref.log_std.exp() * torch.randn((rows, 4), device=dev, generator=g)
        lp_now, _ = ref.log_prob_entropy(obs, act)
        old_lp = lp_now + 0.15 * torch.randn(rows, device=dev, generator=g)     # ratios around 1, some clipped
        adv = torch.randn(rows, device=dev, generator=g) * 2.0 + 0.3
        ret = ref.value(obs) + torch.randn(rows, device=dev, generator=g) * 3.0
    return pol, ref, up, obs.contiguous(), act.contiguous(), old_lp.contiguous(), adv.contiguous(), ret.contiguous()


def _f16_operands(net, x):
    """The network with every GEMM operand (activations, weights, biases) rounded to f16 and f32 accumulation -- what the
    matrix-core kernels compute.  `.half().float()` is a straight-through rounding for autograd."""
    q = lambda t: t + (t.half().float() - t).detach()
    for m in net:
        x = torch.nn.functional.linear(q(x), q(m.weight), q(m.bias)) if isinstance(m, torch.nn.Linear) else m(x)
    return x


def _torch_loss(ref, obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef, f16_operands=False):
    i = idx.long()
    a = adv[i]
    a = (a - a.mean()) / (a.std() + 1e-8)
    if f16_operands:
        mean, v = _f16_operands(ref.pi, obs[i]), _f16_operands(ref.vf, obs[i]).squeeze(-1)
        lp = (-0.5 * ((act[i] - mean) / ref.log_std.exp()) ** 2 - ref.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        ent = (0.5 + 0.5 * math.log(2 * math.pi) + ref.log_std).sum().expand(i.shape[0])
    else:
        lp, ent = ref.log_prob_entropy(obs[i], act[i])
        v = ref.value(obs[i])
    ratio = (lp - old_lp[i]).exp()
    pg = -torch.min(a * ratio, a * ratio.clamp(1 - clip, 1 + clip)).mean()
    vl = torch.nn.functional.mse_loss(v, ret[i])
    return pg + vf_coef * vl - ent_coef * ent.mean(), pg, vl, ratio


def _flat_ref_grads(ref):
    out = []
    for net in (ref.pi, ref.vf):
        for lin in [m for m in net if isinstance(m, torch.nn.Linear)]:
            out += [lin.weight.grad.reshape(-1), lin.bias.grad.reshape(-1)]
    out.append(ref.log_std.grad.reshape(-1))
    return out


@pytest.mark.parametrize("L,B,clip", [(17, 1024, 0.2), (17, 1024, 50.0), (24, 512, 0.2), (36, 256, 50.0), (13, 64, 0.2),
                                      (17, 16384, 0.2), (24, 32768, 50.0), (17, 6400, 0.2),    # 6400: 100 groups -> 25 chunks (no XCD mapping)
                                      (24, 5000, 0.2), (24, 5000, 50.0), (13, 100, 50.0)])      # 5000 = the reference's batch_size (R:792): 78 groups + 8 rows
@pytest.mark.parametrize("partial", ["bf16", "f32"])
def test_gradient_matches_autograd(L, B, clip, partial):
    """Two references: (1) autograd through the same networks with f16-rounded GEMM operands -- what the kernels compute,
    so the comparison is tight; (2) plain f32 autograd -- there the f16 forward flips the ReLU state of units whose
    pre-activation is ~0 and (with clip = 0.2) the branch of ratios on the clip edge, a few-percent unbiased difference.
    clip = 50 switches the clipping off.  Both formats of the per-workgroup partial sums: bf16 (default; every partial carries a
    relative 2^-9 rounding, which a SCALAR parameter like the value head's bias sees undiluted: bound 1e-2 instead of 6e-3) and f32."""
    rows = max(3000, 4 * B)
    pol, ref, up, obs, act, old_lp, adv, ret = _setup(L, rows, seed=L, max_minibatch=max(4096, B), flags=1 if partial == "f32" else 0)
    idx = torch.randperm(rows, device=obs.device)[:B].to(torch.int32).contiguous()
    vf_coef, ent_coef = 0.5, 0.01
    up.stats.zero_()
    g = up.grad(obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef, stats=True)
    loss, pg, vl, ratio = _torch_loss(ref, obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef, f16_operands=True)
    loss.backward()
    refs16 = _flat_ref_grads(ref)
    for p in ref.parameters():
        p.grad = None
    loss, pg, vl, ratio = _torch_loss(ref, obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef)
    loss.backward()
    refs = _flat_ref_grads(ref)
    names = [f"{n}.{l}.{k}" for n in ("pi", "vf") for l in (1, 2, 3, 4) for k in ("w", "b")] + ["log_std"]
    off = 0
    report = []
    for name, r, r16 in zip(names, refs, refs16):
        mine = g[off:off + r.numel()]
        off += r.numel()
        err = float((mine - r).norm() / (r.norm() + 1e-12))
        err16 = float((mine - r16).norm() / (r16.norm() + 1e-12))
        cos = float(torch.dot(mine, r) / (mine.norm() * r.norm() + 1e-20))
        report.append((name, round(err16, 5), round(err, 5), round(cos, 6)))
    print(report)
    for name, err16, err, cos in report:
        if clip > 1 or name.startswith("vf"):
            # same operands: only the f16 rounding of the deltas remains (a 64-sample sum of +-deltas can cancel)
            assert err16 < ((1e-2 if partial == "bf16" else 6e-3) if B >= 256 else 3e-2), (name, err16, report)
        assert err < 1e-1 and cos > 0.995, (name, err, cos, report)
    assert off + 4 == g.numel()                              # the minibatch statistics ride behind the gradient
    st = up.stats.cpu().numpy()
    assert np.allclose(g[off:].cpu().numpy(), st, rtol=1e-6, atol=1e-6)
    assert abs(st[0] / B - float(pg)) < 5e-3 * max(1.0, abs(float(pg)))
    assert abs(st[1] / B - float(vl)) < 5e-3 * max(1.0, abs(float(vl)))
    clipped = float(((ratio - 1).abs() > clip).float().sum())
    assert abs(st[3] - clipped) <= max(3.0, 0.02 * B)      # f16 forward moves a few ratios across the clip edge
    if clip < 1:
        assert clipped > 0.05 * B                            # the clipped branch is exercised


def test_minibatch_update_matches_torch_adam():
    L, rows, B = 17, 4096, 1024
    pol, ref, up, obs, act, old_lp, adv, ret = _setup(L, rows, seed=5)
    opt = torch.optim.Adam(ref.parameters(), lr=3e-4, eps=1e-5)
    theta0 = up.theta.clone()
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    for k in range(4):
        idx = perm[k * B:(k + 1) * B].contiguous()
        up.minibatch(obs, act, old_lp, adv, ret, idx, lr=3e-4)
        loss, *_ = _torch_loss(ref, obs, act, old_lp, adv, ret, idx, 0.2, 0.5, 0.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
    flat_ref = torch.cat([p.detach().reshape(-1) for net in (ref.pi, ref.vf) for lin in net if isinstance(lin, torch.nn.Linear)
                          for p in (lin.weight, lin.bias)] + [ref.log_std.detach().reshape(-1)])
    d_mine, d_ref = up.theta - theta0, flat_ref - theta0
    assert float(d_ref.abs().max()) > 5e-4                       # four Adam steps of 3e-4 moved the parameters
    # Adam's first steps are ~ lr * sign(g): elements whose gradient is near zero may differ, the bulk must agree
    close = ((d_mine - d_ref).abs() < 0.15 * 3e-4 * 4).float().mean()
    assert float(close) > 0.97, float(close)
    assert float(torch.dot(d_mine, d_ref) / (d_mine.norm() * d_ref.norm())) > 0.98
    # the module the updater re-pointed sees the new weights (aliases of theta)
    assert torch.equal(pol.pi[0].weight.reshape(-1), up.theta[:120 * L])
    assert pol.log_std.data_ptr() == up.theta[-4:].data_ptr()


def test_full_epoch_tracks_torch():
    """One PPO epoch at the training size (16 minibatches of 16 384 rows, same permutation, same start): the policy and
    value functions after the native updates stay close to those after torch's updates."""
    L, rows, B = 17, 16 * 16384, 16384
    pol, ref, up, obs, act, old_lp, adv, ret = _setup(L, rows, seed=11, max_minibatch=B)
    opt = torch.optim.Adam(ref.parameters(), lr=3e-4, eps=1e-5)
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    with torch.no_grad():
        mean0, v0 = ref.pi(obs[:4096]).clone(), ref.value(obs[:4096]).clone()
    for k in range(rows // B):
        idx = perm[k * B:(k + 1) * B].contiguous()
        up.minibatch(obs, act, old_lp, adv, ret, idx, lr=3e-4)
        loss, *_ = _torch_loss(ref, obs, act, old_lp, adv, ret, idx, 0.2, 0.5, 0.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
    with torch.no_grad():
        dm_ref, dv_ref = ref.pi(obs[:4096]) - mean0, ref.value(obs[:4096]) - v0
        dm, dv = pol.pi(obs[:4096]) - mean0, pol.value(obs[:4096]) - v0
    cos_m = float((dm * dm_ref).sum() / (dm.norm() * dm_ref.norm()))
    cos_v = float((dv * dv_ref).sum() / (dv.norm() * dv_ref.norm()))
    print("epoch: change of policy mean / value vs torch: cos", cos_m, cos_v, "rel", float((dm - dm_ref).norm() / dm_ref.norm()),
          float((dv - dv_ref).norm() / dv_ref.norm()), "log_std", pol.log_std.tolist(), ref.log_std.tolist())
    assert cos_m > 0.97 and cos_v > 0.97
    assert float((dm - dm_ref).norm() / dm_ref.norm()) < 0.25 and float((dv - dv_ref).norm() / dv_ref.norm()) < 0.25
    assert torch.allclose(pol.log_std, ref.log_std, atol=2e-3)


def test_grad_then_apply_equals_minibatch():
    """The data-parallel split (qr_ppo_grad -> [all-reduce] -> qr_ppo_apply) takes the same step as qr_ppo_minibatch."""
    L, rows, B = 17, 8192, 2048
    pol_a, _, up_a, obs, act, old_lp, adv, ret = _setup(L, rows, seed=21)
    pol_b, _, up_b, *_ = _setup(L, rows, seed=21)
    assert torch.equal(up_a.theta, up_b.theta)
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    for k in range(3):
        idx = perm[k * B:(k + 1) * B].contiguous()
        up_a.minibatch(obs, act, old_lp, adv, ret, idx, lr=3e-4)
        g = up_b.grad(obs, act, old_lp, adv, ret, idx)
        up_b.apply(g, lr=3e-4, B=B)
    assert up_a.step == up_b.step == 3
    assert torch.allclose(up_a.theta, up_b.theta, rtol=0, atol=2e-7), float((up_a.theta - up_b.theta).abs().max())
    assert torch.allclose(up_a.m, up_b.m, rtol=1e-5, atol=1e-9)


def test_argument_validation():
    from optimal_quad_control_rl_amd import _lib
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    pol, ref, up, obs, act, old_lp, adv, ret = _setup(17, 256, seed=2)
    idx = torch.arange(40, device=obs.device, dtype=torch.int32)
    with pytest.raises(_lib.QuadraceError):
        up.grad(obs, act, old_lp, adv, ret, idx)                # fewer than 64 rows
    for removed in (2, 4, 16):                                  # bits 2 / 4 selected the round 1-2 kernel forms (gone); 16 never existed
        with pytest.raises(_lib.QuadraceError):
            MfmaPpoUpdater(pol, 17, obs.device, max_minibatch=4096, flags=removed)
    idx = torch.arange(8192, device=obs.device, dtype=torch.int32) % 256
    with pytest.raises(_lib.QuadraceError):
        up.grad(obs, act, old_lp, adv, ret, idx.contiguous())   # larger than max_minibatch


def test_gae_and_value_forward_match_torch():
    """qr_ppo_gae == the torch GAE / episode-statistics loops of ppo.py; qr_ppo_forward == the networks at f16-operand level."""
    from optimal_quad_control_rl_amd.ppo import PPO

    class FakeEnv:   # just enough of the env surface for PPO.__init__ and the torch reference loops
        num_envs, state_len, device = 4096, 17, torch.device("cuda", 0)

        def reset_device(self):
            return torch.zeros((self.num_envs, self.state_len), device=self.device)

    T, N = 32, 4096
    ref = PPO(FakeEnv(), n_steps=T, batch_size=N * T // 4, gamma=0.99, truncation_bootstrap=False)
    pol, _, up, obs, *_ = _setup(17, 8192, seed=3, max_minibatch=4096)
    g = torch.Generator(device=obs.device).manual_seed(5)
    rew = torch.randn((T, N), device=obs.device, generator=g) * 0.3
    rew[torch.rand((T, N), device=obs.device, generator=g) < 0.02] = 9.5          # gate passes
    done = (torch.rand((T, N), device=obs.device, generator=g) < 0.03).float()
    val = torch.randn((T, N), device=obs.device, generator=g)
    last_val = torch.randn(N, device=obs.device, generator=g)
    ref.buf_rew.copy_(rew); ref.buf_done.copy_(done); ref.buf_val.copy_(val); ref.last_val = last_val
    adv_ref, ret_ref = ref._gae()
    ref._episode_stats(rew, done)
    er, el, eg = (torch.zeros(N, device=obs.device) for _ in range(3))
    fin = torch.zeros(4, device=obs.device)
    adv, ret = up.gae(rew.contiguous(), done.contiguous(), val.contiguous(), last_val.contiguous(), 0.99, 0.95, (er, el, eg), fin)
    assert torch.allclose(adv, adv_ref, rtol=1e-5, atol=1e-5) and torch.allclose(ret, ret_ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(er, ref.ep_ret, atol=1e-4) and torch.equal(el, ref.ep_len) and torch.equal(eg, ref.ep_gates)
    f = fin.tolist()
    assert f[3] == float(done.sum()) and abs(f[1] / f[3] - ref.stats["ep_len_mean"]) < 1e-3
    assert abs(f[0] / f[3] - ref.stats["ep_rew_mean"]) < 1e-3 and abs(f[2] / f[3] - ref.stats["gates_per_episode"]) < 1e-4
    with torch.no_grad():
        v_ref, m_ref = pol.value(obs), pol.pi(obs)
    v, m = up.forward(1, obs), up.forward(0, obs)
    assert (v - v_ref).abs().max() < 2e-2 * max(1.0, float(v_ref.abs().max()))
    assert (m - m_ref).abs().max() < 2e-2 * max(1.0, float(m_ref.abs().max()))


def test_gae_with_truncation_bootstrap_matches_sb3_rule():
    """SB3 restated in torch: `collect_rollouts` adds gamma * V(terminal_observation) to the reward of every step that ended
    by the time limit (infos["TimeLimit.truncated"]) BEFORE the buffer stores it; `compute_returns_and_advantage` then treats
    every done as the end of the episode (next_non_terminal = 1 - episode_start).  qr_ppo_gae takes V(terminal obs) as
    term_val [T, N] (0 where the step was not truncated) and must give the same advantages / returns; the episode returns of
    the on-device VecMonitor stay those of the RAW rewards."""
    pol, _, up, obs, *_ = _setup(17, 4096, seed=9, max_minibatch=4096)
    dev = obs.device
    T, N, gamma, lam = 48, 2048, 0.999, 0.95
    g = torch.Generator(device=dev).manual_seed(17)
    rew = torch.randn((T, N), device=dev, generator=g) * 0.3
    done = (torch.rand((T, N), device=dev, generator=g) < 0.05)
    trunc = done & (torch.rand((T, N), device=dev, generator=g) < 0.5)            # half of the episode ends are time limits
    val = torch.randn((T, N), device=dev, generator=g) * 5.0
    last_val = torch.randn(N, device=dev, generator=g) * 5.0
    term_v = torch.randn((T, N), device=dev, generator=g) * 5.0                  # V(terminal_observation) of each row
    term_val = torch.where(trunc, term_v, torch.zeros_like(term_v)).contiguous()
    # --- SB3
    rewards = rew.clone()
    rewards[trunc] += gamma * term_v[trunc]                                       # collect_rollouts
    adv_ref = torch.zeros_like(rew)
    last_gae = torch.zeros(N, device=dev)
    for t in reversed(range(T)):                                                  # compute_returns_and_advantage
        if t == T - 1:
            next_non_terminal, next_values = 1.0 - done[t].float(), last_val      # SB3 passes `dones` of the last step
        else:
            next_non_terminal, next_values = 1.0 - done[t].float(), val[t + 1]    # episode_starts[t + 1] == dones[t]
        delta = rewards[t] + gamma * next_values * next_non_terminal - val[t]
        last_gae = delta + gamma * lam * next_non_terminal * last_gae
        adv_ref[t] = last_gae
    ret_ref = adv_ref + val
    er, el, eg = (torch.zeros(N, device=dev) for _ in range(3))
    fin = torch.zeros(4, device=dev)
    adv, ret = up.gae(rew.contiguous(), done.float().contiguous(), val.contiguous(), last_val.contiguous(), gamma, lam, (er, el, eg), fin,
                      term_val=term_val)
    assert torch.allclose(adv, adv_ref, rtol=1e-5, atol=2e-4), float((adv - adv_ref).abs().max())
    assert torch.allclose(ret, ret_ref, rtol=1e-5, atol=2e-4)
    # without term_val a truncation is a termination: the advantages at truncated rows differ by gamma * V(terminal obs)
    adv0, _ = up.gae(rew.contiguous(), done.float().contiguous(), val.contiguous(), last_val.contiguous(), gamma, lam)
    assert torch.allclose((adv - adv0)[trunc], gamma * term_v[trunc], rtol=1e-4, atol=1e-3)
    # VecMonitor sums use the raw rewards
    ep = torch.zeros(N, device=dev); tot = 0.0
    for t in range(T):
        ep += rew[t]; tot += float((ep * done[t].float()).sum()); ep *= 1.0 - done[t].float()
    assert abs(float(fin[0]) - tot) < 1e-2 * max(1.0, abs(tot)) and float(fin[3]) == float(done.sum())


def test_target_kl_early_stop_runs_on_the_device():
    """SB3's rule: approx_kl of a minibatch > 1.5 target_kl -> that optimiser step is not taken and training of this
    rollout stops.  Here the update kernel takes the decision; later launches are no-ops until control(clear=True)."""
    L, rows, B = 17, 8192, 2048
    pol, ref, up, obs, act, old_lp, adv, ret = _setup(L, rows, seed=33)
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    up.control(None, clear=True)
    up.begin_epoch(adv, perm, B)
    for k in range(2):
        up.minibatch(obs, act, old_lp, adv, ret, perm[k * B:(k + 1) * B], lr=3e-4)
    assert up.status() == (False, 2, 0, 0)
    theta2 = up.theta.clone()
    # the synthetic old log-probs are 0.15-sigma off: approx KL ~ 1e-2 per sample >> 1.5e-6
    with torch.no_grad():
        lp, _ = pol.log_prob_entropy(obs, act)
        kl = float(((lp - old_lp).exp() - 1 - (lp - old_lp)).mean())
    assert kl > 1e-3
    up.control(1e-6, clear=True)
    up.stats.zero_()
    up.minibatch(obs, act, old_lp, adv, ret, perm[2 * B:3 * B], lr=3e-4)      # exceeds the limit: no step, stop flag set
    up.minibatch(obs, act, old_lp, adv, ret, perm[3 * B:4 * B], lr=3e-4)      # skipped entirely
    assert up.status() == (True, 0, 0, 0)
    assert torch.equal(up.theta, theta2)
    assert abs(float(up.stats[2]) / B - kl) < 0.3 * kl                         # the breaking minibatch's KL was recorded (once)
    up.control(10.0 * kl, clear=True)                                           # generous limit: training resumes
    up.minibatch(obs, act, old_lp, adv, ret, perm[2 * B:3 * B], lr=3e-4)
    assert up.status() == (False, 1, 0, 0) and not torch.equal(up.theta, theta2)


def test_epoch_table_equals_per_minibatch_statistics_and_nonfinite_guard():
    L, rows, B = 17, 8192, 2048
    pol_a, _, up_a, obs, act, old_lp, adv, ret = _setup(L, rows, seed=41)
    pol_b, _, up_b, *_ = _setup(L, rows, seed=41)
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    up_a.begin_epoch(adv, perm, B)                      # one launch for the four minibatches' advantage sums
    for k in range(4):
        up_a.minibatch(obs, act, old_lp, adv, ret, perm[k * B:(k + 1) * B], lr=3e-4)
        up_b.minibatch(obs, act, old_lp, adv, ret, perm[k * B:(k + 1) * B].clone(), lr=3e-4)   # not announced: own launch
    assert torch.allclose(up_a.theta, up_b.theta, rtol=0, atol=2e-7)
    # image re-pack inside the update kernel == the gather pack of the same parameters
    v_scatter = up_a.forward(1, obs[:2048].contiguous()).clone()
    up_a.pack()
    assert torch.equal(v_scatter, up_a.forward(1, obs[:2048].contiguous()))
    m_scatter = up_b.forward(0, obs[:2048].contiguous()).clone()
    up_b.pack()
    assert torch.equal(m_scatter, up_b.forward(0, obs[:2048].contiguous()))
    # a non-finite gradient leaves parameters and Adam moments untouched and is counted
    before, m_before = up_a.theta.clone(), up_a.m.clone()
    bad_adv = adv.clone(); bad_adv[perm[0].long()] = float("nan")
    up_a.control(None, clear=True)
    up_a.minibatch(obs, act, old_lp, bad_adv, ret, perm[:B], lr=3e-4)
    assert up_a.status() == (False, 0, 1, 0)
    assert torch.equal(up_a.theta, before) and torch.equal(up_a.m, m_before)
    up_a.minibatch(obs, act, old_lp, adv, ret, perm[:B], lr=3e-4)               # and training continues afterwards
    assert up_a.status()[1] == 1 and torch.isfinite(up_a.theta).all()


@pytest.mark.parametrize("L,B", [(17, 1024), (24, 16384), (13, 192), (36, 40000 // 64 * 64), (17, 65536), (24, 16384 + 128), (24, 5000)])
def test_gradient_is_invariant_under_row_order(L, B):
    """The gradient of a minibatch is a SUM over its rows: presenting the same rows in another order puts every row into a different
    32-sample tile, workgroup and pass (B = 39 936: 312 pairs of sample groups over 128 workgroups, a ragged last pass; B = 65 536: four
    passes per workgroup; 5 000 / 16 512: a masked partial group) and must give the same sums to f32 summation noise -- a check of the
    tile / workgroup / pass bookkeeping that needs no second implementation (rounds 2-5 compared against the round-1/2 kernel forms,
    removed in round 6).  f32 partials: the arithmetic itself; bf16 partials (the default): deterministic, identical statistics and
    log_std gradient, and a difference bounded by the 2^-9 rounding of each per-workgroup partial."""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    pol, ref, up8, obs, act, old_lp, adv, ret = _setup(L, rows=max(B, 4096) * 2, seed=7, max_minibatch=B + 63, flags=1)
    up8h = MfmaPpoUpdater(pol, L, obs.device, max_minibatch=B + 63)   # the default: per-workgroup partials leave as bf16
    idx = torch.randperm(obs.shape[0], device=obs.device)[:B].to(torch.int32)
    idx2 = idx[torch.randperm(B, device=obs.device)].contiguous()
    idx3 = idx.flip(0).contiguous()
    G = lambda u, i: u.grad(obs, act, old_lp, adv, ret, i, clip=0.2, vf_coef=0.5, ent_coef=0.01, stats=True).clone()   # noqa: E731
    g8, g8b, g8p, g8r = G(up8, idx), G(up8, idx), G(up8, idx2), G(up8, idx3)
    torch.cuda.synchronize()
    n = g8.numel() - 4
    assert torch.isfinite(g8).all() and float(g8[:n].abs().max()) > 0
    assert torch.equal(g8, g8b)                                       # deterministic (no atomics)
    scale = float(g8[:n].abs().max())
    for other in (g8p, g8r):
        assert float((g8[:n] - other[:n]).abs().max()) <= 4e-6 * scale + 1e-7, (float((g8[:n] - other[:n]).abs().max()), scale)
        assert torch.allclose(g8[n:], other[n:], rtol=2e-5, atol=1e-6)   # the four minibatch statistics
    # 16-bit partials (the default): each of the <= 128 per-workgroup partial sums is rounded to bf16 (relative 2^-9) before the
    # apply kernel's fixed-order f32 sum -- deterministic, same statistics, and a gradient that differs from the f32-partial one by
    # far less than the f16 operands already cost against plain f32 (cosine >= 0.999 there)
    g8h, g8h2 = G(up8h, idx), G(up8h, idx)
    torch.cuda.synchronize()
    assert torch.equal(g8h, g8h2) and torch.equal(g8h[n:], g8[n:])
    assert torch.equal(g8h[n - 4:n], g8[n - 4:n])                      # log_std does not travel through the partials
    err = float((g8h[:n] - g8[:n]).abs().max())
    cos = float(torch.dot(g8h[:n].double(), g8[:n].double()) / (g8h[:n].double().norm() * g8[:n].double().norm()))
    assert 0 < err <= 8e-3 * scale and cos > 1 - 1e-5, (err, scale, cos)   # |error| <= 2^-9 sum_q |partial_q|: partials cancel
    up8.close(); up8h.close()


def test_fused_gradient_is_deterministic_and_stateless_across_minibatch_sizes():
    """No atomics anywhere in the update: the same minibatch gives the same bits twice; and a call does not see leftovers of an
    earlier, larger call in the partial / per-wave buffers (fewer workgroups write fewer partial rows than the previous launch)."""
    pol, ref, up, obs, act, old_lp, adv, ret = _setup(17, rows=40000, seed=5, max_minibatch=32768)
    dev = obs.device
    idx_big = torch.randperm(obs.shape[0], device=dev)[:32768].to(torch.int32)
    idx_small = idx_big[:192].contiguous()     # 3 sample groups: one full pair + a half-empty workgroup pass
    g1 = up.grad(obs, act, old_lp, adv, ret, idx_small).clone()
    gb = up.grad(obs, act, old_lp, adv, ret, idx_big).clone()
    g2 = up.grad(obs, act, old_lp, adv, ret, idx_small).clone()
    gb2 = up.grad(obs, act, old_lp, adv, ret, idx_big).clone()
    torch.cuda.synchronize()
    assert torch.equal(g1, g2) and torch.equal(gb, gb2)
    assert torch.isfinite(g1).all() and torch.isfinite(gb).all() and float(gb[:-4].abs().max()) > 0
    up.close()


@pytest.mark.parametrize("L,B,clip", [(17, 1024, 0.2), (24, 5000, 0.2), (24, 16384, 50.0), (36, 256, 0.2), (13, 100, 50.0), (24, 40000, 0.2)])
def test_f32class_gradient_matches_float64_autograd(L, B, clip):
    """Round 6: the reference-precision gradient kernels (qr_ppo_grad_f32class: every GEMM operand of the forward pass,
    the backward pass and the weight gradients as three bf16 pieces, f32 accumulation) against FLOAT64 autograd on the same rows: cosine
    >= 1 - 1e-6 over the whole gradient and per tensor, relative error per tensor at the float32 level (float32 torch autograd sits at the
    same level), identical minibatch statistics -- where the f16-operand kernel has cosine >= 0.9985.  With clip = 0.2 a sample whose ratio
    sits within float32 noise of the clip edge may take the other branch: the bound is looser there but still 100 x below the f16 kernel's."""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    rows = max(3000, 2 * B)
    pol, ref, up16, obs, act, old_lp, adv, ret = _setup(L, rows, seed=L + 1, max_minibatch=max(4096, B))
    up = MfmaPpoUpdater(pol, L, obs.device, max(4096, B), precision="f32")
    idx = torch.randperm(rows, device=obs.device)[:B].to(torch.int32).contiguous()
    vf_coef, ent_coef = 0.5, 0.01
    up.stats.zero_()
    g = up.grad(obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef, stats=True)
    g16 = up16.grad(obs, act, old_lp, adv, ret, idx, clip, vf_coef, ent_coef)
    ref64 = copy.deepcopy(ref).double()
    loss, pg, vl, ratio = _torch_loss(ref64, obs.double(), act.double(), old_lp.double(), adv.double(), ret.double(), idx, clip, vf_coef, ent_coef)
    loss.backward()
    want = torch.cat(_flat_ref_grads(ref64))
    n = want.numel()
    assert g.numel() == n + 4 and torch.isfinite(g).all()
    cos = float(torch.dot(g[:n].double(), want) / (g[:n].double().norm() * want.norm()))
    cos16 = float(torch.dot(g16[:n].double(), want) / (g16[:n].double().norm() * want.norm()))
    rel = float((g[:n].double() - want).norm() / want.norm())
    print("L=%d B=%d clip=%g: f32-class cosine 1 - %.2e, relative error %.2e | f16-operand kernel cosine 1 - %.2e" % (L, B, clip, 1 - cos, rel, 1 - cos16))
    assert cos >= 1 - 1e-6 and rel <= (2e-5 if clip > 1 else 1.5e-3), (cos, rel)
    assert (1 - cos) * 50 < (1 - cos16)                                   # and it really is another class of arithmetic
    names = [f"{nn}.{l}.{k}" for nn in ("pi", "vf") for l in (1, 2, 3, 4) for k in ("w", "b")] + ["log_std"]
    off = 0
    for name, r in zip(names, _flat_ref_grads(ref64)):
        mine = g[off:off + r.numel()].double()
        off += r.numel()
        err = float((mine - r).norm() / (r.norm() + 1e-30))
        assert err <= (5e-5 if clip > 1 or name.startswith("vf") else 5e-3), (name, err)
    st = up.stats.cpu().numpy()
    assert np.allclose(g[n:].cpu().numpy(), st, rtol=1e-6, atol=1e-6)
    assert abs(st[0] / B - float(pg)) < 1e-5 * max(1.0, abs(float(pg))) and abs(st[1] / B - float(vl)) < 1e-5 * max(1.0, abs(float(vl)))
    up.close()


def test_f32class_minibatch_update_tracks_float64_adam():
    """grad_f32class -> qr_ppo_apply: three minibatch updates of MfmaPpoUpdater(precision='f32') against torch.optim.Adam on float64
    autograd gradients with the same clipping: parameters agree to 2e-6 (the f16-operand update: 1e-4 level)."""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    L, B, rows = 24, 4096, 16384
    pol, ref, up16, obs, act, old_lp, adv, ret = _setup(L, rows, seed=9, max_minibatch=B)
    up = MfmaPpoUpdater(pol, L, obs.device, B, precision="f32")
    ref64 = copy.deepcopy(ref).double()
    opt = torch.optim.Adam(ref64.parameters(), lr=3e-4, eps=1e-5)
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    for k in range(3):
        idx = perm[k * B:(k + 1) * B].contiguous()
        up.minibatch(obs, act, old_lp, adv, ret, idx, lr=3e-4, clip=50.0, vf_coef=0.5, ent_coef=0.0, max_grad_norm=0.5)
        opt.zero_grad()
        loss, *_ = _torch_loss(ref64, obs.double(), act.double(), old_lp.double(), adv.double(), ret.double(), idx, 50.0, 0.5, 0.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref64.parameters(), 0.5)
        opt.step()
    want = torch.cat([p.detach().reshape(-1) for net in (ref64.pi, ref64.vf) for m in net if isinstance(m, torch.nn.Linear) for p in (m.weight, m.bias)] + [ref64.log_std.detach()])
    err = float((up.theta.double() - want).abs().max())
    print("max |theta - float64 Adam| after 3 updates: %.2e" % err)
    assert err <= 2e-6
    up.close()


def test_sb3_precision_f32_trains_on_the_hand_written_reference_precision_kernels():
    """precision='f32' of the SB3-shaped PPO (round 6): the reference's own recipe shape (100 envs x 1000 steps, batch_size 5000, 10 epochs)
    runs WITHOUT a torch-evaluated network in the loop -- one closed-loop collect kernel with the f32-class policy forward, the value
    estimates through the same f32-class forward kernel, every minibatch update in the f32-class gradient kernels + the f32 apply kernel -- and takes every one of its 2 x 10 x 20 optimiser steps with finite results;
    after the same two rollouts its parameters stay close to the f16-operand path's (same seeds, same noise stream: the two differ by
    arithmetic only)."""
    from optimal_quad_control_rl_amd import PPO, Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, square_track

    kw = dict(policy_kwargs=dict(activation_fn=torch.nn.ReLU, net_arch=[dict(pi=[120, 120, 120], vf=[120, 120, 120])]),
              n_steps=1000, batch_size=5000, n_epochs=10, gamma=0.999, seed=5)
    thetas = {}
    for precision in ("f32", "f16-operands"):
        env = Quadcopter3DGates(100, *square_track(), gates_ahead=1, infos_mode="none", seed=3)
        env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
        m = PPO("MlpPolicy", env, precision=precision, **kw)
        tr = m._trainer
        assert tr.native_update and tr.fused_collect
        assert tr._updater.precision == ("f32" if precision == "f32" else "f16-operands")
        assert tr.policy_forward == ("f32class" if precision == "f32" else "torch")
        theta0 = tr._updater.theta.clone()
        m.learn(total_timesteps=2 * 100 * 1000)
        st = tr.stats
        assert m.num_timesteps == 200000 and st["updates"] == 2 * 10 * 20 and st["skipped_nonfinite"] == 0 and np.isfinite(st["loss"])
        assert torch.isfinite(tr._updater.theta).all() and not torch.equal(tr._updater.theta, theta0)
        thetas[precision] = (theta0, tr._updater.theta.clone())
        if precision == "f32":
            # no network is evaluated by torch in this mode: the collect phase's value estimates come from the f32-class forward kernel
            # too (second policy-kernel handle, value head padded to four rows) and sit at float32 level of the torch evaluation
            assert tr._mfma_vf is not None
            tr.collect_fused()
            want = tr.policy.value(tr.buf_obs.view(-1, tr.buf_obs.shape[-1])).view_as(tr.buf_val)
            err = float((tr.buf_val - want).abs().max()) / max(1.0, float(want.abs().max()))
            print("collect-phase values, f32-class kernel vs torch float32: max relative difference %.2e" % err)
            assert err <= 3e-6
        else:
            assert tr._mfma_vf is None
        env.close()
    assert torch.equal(thetas["f32"][0], thetas["f16-operands"][0])                  # same initialisation
    moved = float((thetas["f32"][1] - thetas["f32"][0]).norm())
    apart = float((thetas["f32"][1] - thetas["f16-operands"][1]).norm())
    print("parameters moved %.3f, the two precisions ended %.3f apart" % (moved, apart))
    assert apart < moved                                                              # they went the same way


def test_f32class_graph_replay_is_bit_identical_to_plain_launches_and_survives_eviction():
    """qr_ppo_grad_f32class replays its ~20 launches as one cached hipGraph per distinct argument set (quadrace_ppo_f32.hip): a first call
    (capture), a second call with the same arguments (cache hit) and a handle created with QR_PPO_NO_EPOCH_GRAPH (plain launches) give the
    SAME bits; 300 distinct minibatch offsets push the 256-entry cache through its replacement path and the first offset, captured again
    afterwards, still gives the same bits."""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    L, B, rows = 24, 96, 300 * 8 + 96
    pol, ref, up16, obs, act, old_lp, adv, ret = _setup(L, rows, seed=11, max_minibatch=4096)
    graphs = MfmaPpoUpdater(pol, L, obs.device, 4096, precision="f32")
    plain = MfmaPpoUpdater(pol, L, obs.device, 4096, precision="f32", flags=8)
    perm = torch.randperm(rows, device=obs.device).to(torch.int32).contiguous()
    out = torch.empty(graphs.theta.numel() + 4, dtype=torch.float32, device=obs.device)
    out_plain = torch.empty_like(out)

    def run(up, buf, off):
        buf.fill_(float("nan"))
        return up.grad(obs, act, old_lp, adv, ret, perm[off:off + B], 0.2, 0.5, 0.01, out=buf).clone()

    first = run(graphs, out, 0)                      # capture
    again = run(graphs, out, 0)                      # cache hit
    want = run(plain, out_plain, 0)                  # no graph
    assert torch.isfinite(first).all()
    assert torch.equal(first, again) and torch.equal(first, want)
    for k in range(1, 300):                          # 299 more argument sets: the cache holds 256
        got = run(graphs, out, 8 * k)
        if k % 37 == 0:
            assert torch.equal(got, run(plain, out_plain, 8 * k)), k
    assert torch.equal(run(graphs, out, 0), want)    # offset 0 was replaced by then: captured again, same bits
    graphs.close(); plain.close()


@pytest.mark.parametrize("L,B", [(24, 2), (24, 3), (20, 31), (21, 33), (25, 63), (28, 65), (29, 255), (32, 257), (36, 513), (13, 4097), (17, 769), (24, 1025)])
def test_f32class_gradient_ragged_minibatches_and_every_observation_length(L, B):
    """Edge shapes of qr_ppo_grad_f32class against float64 autograd: the smallest minibatch the entry accepts (2 rows), sizes one row either
    side of the 32-row tile, the 64-row wave round and the 256-row weight-gradient slice (a slice of ONE row), and every observation length
    of the race envs (only 24 and 32 take the 16-byte load path of layer 1; the rest the 4-byte one).  clip = 50: no sample sits on a clip
    edge, so the bound is the float32-level one."""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    rows = max(3000, 2 * B)
    pol, ref, up16, obs, act, old_lp, adv, ret = _setup(L, rows, seed=100 + L + B, max_minibatch=max(4096, B))
    up = MfmaPpoUpdater(pol, L, obs.device, max(4096, B), precision="f32")
    idx = torch.randperm(rows, device=obs.device)[:B].to(torch.int32).contiguous()
    g = up.grad(obs, act, old_lp, adv, ret, idx, 50.0, 0.5, 0.01)
    ref64 = copy.deepcopy(ref).double()
    loss, pg, vl, ratio = _torch_loss(ref64, obs.double(), act.double(), old_lp.double(), adv.double(), ret.double(), idx, 50.0, 0.5, 0.01)
    loss.backward()
    want = torch.cat(_flat_ref_grads(ref64))
    n = want.numel()
    assert g.numel() == n + 4 and torch.isfinite(g).all()
    cos = float(torch.dot(g[:n].double(), want) / (g[:n].double().norm() * want.norm()))
    rel = float((g[:n].double() - want).norm() / want.norm())
    print("L=%d B=%d: cosine 1 - %.2e, relative error %.2e" % (L, B, 1 - cos, rel))
    assert cos >= 1 - 1e-6 and rel <= 2e-5, (L, B, cos, rel)
    up.close()


def test_f32class_gradient_rejects_bad_minibatch_sizes():
    """qr_ppo_grad_f32class: B < 2, B > max_minibatch and B beyond the 65 535 row tiles a launch can hold are QR_E_INVALID with a message, not
    a failed launch"""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    L, rows = 24, 4096
    pol, ref, up16, obs, act, old_lp, adv, ret = _setup(L, rows, seed=5, max_minibatch=4096)
    up = MfmaPpoUpdater(pol, L, obs.device, 4096, precision="f32")
    idx = torch.arange(rows, device=obs.device, dtype=torch.int32)
    with pytest.raises(Exception, match="minibatch size"):
        up.grad(obs, act, old_lp, adv, ret, idx[:1])
    up.close()
    big = MfmaPpoUpdater(pol, L, obs.device, 2097152 + 64, precision="f32")
    idx_big = torch.zeros(2097152 + 64, device=obs.device, dtype=torch.int32)
    with pytest.raises(Exception, match="2 097 120"):
        big.grad(obs, act, old_lp, adv, ret, idx_big)
    big.close()


def test_sb3_precision_f32_checkpoint_resumes_in_that_precision_bit_for_bit(tmp_path):
    """A model trained with precision='f32' and saved is, after PPO.load onto a fresh env, the SAME model in the SAME arithmetic: the
    checkpoint carries the precision (an explicit keyword still wins), and one more learn() on the original and on the reloaded copy gives
    identical parameters and Adam state -- the f32-class collect kernel, value kernel, gradient kernels (fixed summation order, cached graph
    or not) and the apply kernel are all deterministic."""
    from optimal_quad_control_rl_amd import PPO, Quadcopter3DGates, TRAIN_DISTURBANCE_RANGES, VecMonitor, square_track

    def make(**kw):
        env = VecMonitor(Quadcopter3DGates(1024, *square_track(), gates_ahead=1, infos_mode="none", seed=3))
        env.venv.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
        pk = dict(activation_fn=torch.nn.ReLU, net_arch=[dict(pi=[120, 120, 120], vf=[120, 120, 120])], log_std_init=0)
        return PPO("MlpPolicy", env, policy_kwargs=pk, verbose=0, n_steps=16, batch_size=4096, n_epochs=2, gamma=0.999, seed=5, **kw), env

    model, env = make(precision="f32")
    tr = model._trainer
    assert tr.native_update and tr.fused_collect and tr._updater.precision == "f32" and tr._mfma_vf is not None
    steps = model.n_steps * env.num_envs * 2
    model.learn(total_timesteps=steps, reset_num_timesteps=False)
    path = model.save(str(tmp_path / "f32" / str(model.num_timesteps)))
    _, env2 = make()                                                     # a fresh env built with the same arguments
    loaded = PPO.load(path, env=env2)
    assert loaded.precision == "f32" and loaded._trainer._updater.precision == "f32" and loaded._trainer._mfma_vf is not None
    assert PPO.load(path, precision="f16-operands").precision == "f16-operands"      # an explicit keyword wins
    model.learn(total_timesteps=steps, reset_num_timesteps=False)
    loaded.learn(total_timesteps=steps, reset_num_timesteps=False)
    assert loaded.num_timesteps == model.num_timesteps == 2 * steps
    sd_a, sd_b = model.policy.state_dict(), loaded.policy.state_dict()
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    assert torch.equal(tr._updater.m, loaded._trainer._updater.m) and torch.equal(tr._updater.v, loaded._trainer._updater.v)
    assert tr._updater.step == loaded._trainer._updater.step and tr.stats["skipped_nonfinite"] == 0


def test_f32class_update_honours_target_kl_and_the_nonfinite_guard():
    """precision='f32' reaches the device-side decisions of qr_ppo_apply through another call sequence (qr_ppo_grad_f32class, then apply, the
    minibatch statistics riding behind the gradient): SB3's target-KL rule -- no step, stop flag, later launches no-ops, the breaking
    minibatch's KL recorded once, training resumes after control(clear=True) -- and the non-finite guard (a NaN advantage leaves parameters
    and Adam moments untouched and is counted) behave as on the f16-operand path."""
    from optimal_quad_control_rl_amd.ppo import MfmaPpoUpdater

    L, rows, B = 17, 8192, 2048
    pol, ref, up16, obs, act, old_lp, adv, ret = _setup(L, rows, seed=33)
    up = MfmaPpoUpdater(pol, L, obs.device, 4096, precision="f32")
    perm = torch.randperm(rows, device=obs.device).to(torch.int32)
    up.control(None, clear=True)
    for k in range(2):
        up.minibatch(obs, act, old_lp, adv, ret, perm[k * B:(k + 1) * B], lr=3e-4)
    assert up.status() == (False, 2, 0, 0)
    theta2 = up.theta.clone()
    with torch.no_grad():
        lp, _ = pol.log_prob_entropy(obs, act)
        kl = float(((lp - old_lp).exp() - 1 - (lp - old_lp)).mean())
    assert kl > 1e-3
    up.control(1e-6, clear=True)
    up.stats.zero_()
    up.minibatch(obs, act, old_lp, adv, ret, perm[2 * B:3 * B], lr=3e-4)      # exceeds the limit: no step, stop flag set
    up.minibatch(obs, act, old_lp, adv, ret, perm[3 * B:4 * B], lr=3e-4)      # a no-op
    assert up.status() == (True, 0, 0, 0)
    assert torch.equal(up.theta, theta2)
    assert abs(float(up.stats[2]) / B - kl) < 0.3 * kl                         # recorded once
    up.control(10.0 * kl, clear=True)
    up.minibatch(obs, act, old_lp, adv, ret, perm[2 * B:3 * B], lr=3e-4)
    assert up.status() == (False, 1, 0, 0) and not torch.equal(up.theta, theta2)
    # non-finite guard
    before, m_before = up.theta.clone(), up.m.clone()
    bad_adv = adv.clone(); bad_adv[perm[0].long()] = float("nan")
    up.control(None, clear=True)
    up.minibatch(obs, act, old_lp, bad_adv, ret, perm[:B], lr=3e-4)
    assert up.status() == (False, 0, 1, 0)
    assert torch.equal(up.theta, before) and torch.equal(up.m, m_before)
    up.minibatch(obs, act, old_lp, adv, ret, perm[:B], lr=3e-4)               # and training continues afterwards
    assert up.status() == (False, 1, 1, 0) and torch.isfinite(up.theta).all()
    up.close()
