"""GPU suite for the predecessor environments (include/quad3d.h; Quadcopter3DVec / Quadcopter3DVecGates of
"3D quad.ipynb"): the HIP path against the committed reference fixtures, against the CPU oracle on
seeded inputs through auto-resets (resets must agree BIT-FOR-BIT: same Philox / Box-Muller specification), and
size-independent properties at 65 536 envs."""
import numpy as np
import pytest
import torch

import parity_quad3d as pq
from parity import rel_err

pytestmark = pytest.mark.gpu

FULL_N = 65536


class ProductImpl:
    def __init__(self, kind, n, track, **kw):
        from optimal_quad_control_rl_amd.quad3d import Quadcopter3DVec, Quadcopter3DVecGates

        self.env = Quadcopter3DVec(n, **kw) if kind == "hover" else Quadcopter3DVecGates(n, *track, **kw)

    def set_state(self, states, target, steps):
        self.env.set_state_tensors(states=np.asarray(states), target=None if target is None else np.asarray(target, np.int32),
                                   steps=np.asarray(steps, np.int32))

    def get_state(self):
        st, tg, sc = self.env.get_state_tensors()
        return st.cpu().numpy(), tg.cpu().numpy(), sc.cpu().numpy()

    def step(self, actions):
        a = torch.as_tensor(np.ascontiguousarray(actions, dtype=np.float32)).to(self.env.device)
        st, rew, done, trunc = self.env.step_device(a)
        return st.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool)

    def reset(self, mask=None):
        return self.env.reset_device(mask).cpu().numpy()


def make(kind, n, track):
    return ProductImpl(kind, n, track)


def oracle(kind, n, track, **kw):
    from oracle import quad3d as q3

    return q3.Quad3DOracle(q3.HOVER, n, **kw) if kind == "hover" else q3.Quad3DOracle(q3.GATES, n, *track, **kw)


# ---- reference fixtures --------------------------------------------------------------------------------------------
def test_hover_step_branches_vs_reference():
    pq.check_hover_step(make)


def test_hover_free_run_vs_reference():
    pq.check_hover_free_run(make)


def test_gates_step_branches_vs_reference():
    pq.check_gates_step(make)


def test_gates_free_run_vs_reference():
    pq.check_gates_free_run(make)


def test_reset_distributions_vs_reference():
    pq.check_reset_distribution(make)


# ---- product vs oracle on seeded inputs ------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["hover", "gates"])
def test_reset_bit_exact_vs_oracle(kind):
    trk = pq.gates_track()
    kw = dict(seed=0xDEADBEEFCAFE, env_id_base=(1 << 32) - 77)
    g, o = ProductImpl(kind, 5000, trk, **kw), oracle(kind, 5000, trk, **kw)
    sg, so = g.reset(), o.reset()
    np.testing.assert_array_equal(sg, so)
    np.testing.assert_array_equal(g.get_state()[1], o.target)
    mask = np.arange(5000) % 3 == 0
    sg2, so2 = g.reset(mask), o.reset(mask)
    np.testing.assert_array_equal(sg2, so2)
    np.testing.assert_array_equal(sg2[~mask], sg[~mask])
    assert (sg2[mask] != sg[mask]).any(axis=1).all()


@pytest.mark.parametrize("kind", ["hover", "gates"])
def test_lockstep_vs_oracle_through_resets(kind):
    """Teacher-forced lock-step, N = 4096, random actions, max_steps = 30 so every env is auto-reset in the window."""
    n, K = 4096, 45
    trk = pq.gates_track()
    g, o = ProductImpl(kind, n, trk, seed=7), oracle(kind, n, trk, seed=7)
    g.env.max_steps = 30
    o.set_limits(30)
    g.reset(); o.reset()
    tol_s, tol_r = (pq.TOL64_STEP, pq.TOL64_STEP) if kind == "hover" else (pq.TOL32_STEP_STATE, pq.TOL32_STEP_REWARD)
    rng = np.random.default_rng(3)
    total_done = 0
    desync = np.zeros(n, bool)   # envs whose reset streams diverged after a knife-edge done mismatch
    for k in range(K):
        g.set_state(o.states, o.target, o.steps)
        a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
        if k % 2:
            a = (0.2 * a).astype(np.float32)
        sg, rg, dg, tg = g.step(a)
        so, ro, do, to = o.step(a)
        mism = dg != do
        assert mism.sum() <= 1, f"step {k}: {mism.sum()} done mismatches"
        ok = ~mism
        np.testing.assert_array_equal(tg[ok], to[ok])
        assert np.abs(rg[ok].astype(np.float64) - ro[ok]).max() <= tol_r
        live, done = ok & ~do, ok & do
        assert rel_err(sg[live], so[live]).max() <= tol_s
        # episode counters are not injected: freshly reset lanes agree bit for bit as long as both sides have reset
        # that env equally often (always, except after a knife-edge mismatch)
        desync |= mism
        np.testing.assert_array_equal(sg[done & ~desync], so[done & ~desync])
        _, tgt, steps = g.get_state()
        np.testing.assert_array_equal(tgt[ok], o.target[ok])
        np.testing.assert_array_equal(steps[ok], o.steps[ok])
        total_done += int(do.sum())
    assert total_done >= n


# ---- size-independent properties at BASELINE size ----------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["hover", "gates"])
def test_determinism_sharding_and_rollout_equivalence(kind):
    """(a) same seed -> bitwise identical; (b) two half-size shards with env_id_base offsets == one full env;
    (c) q3_step_many == K x q3_step, bitwise."""
    trk = pq.gates_track()
    n, K = FULL_N, 40
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    acts = (torch.rand((K, n, 4), device=dev, generator=gen) * 2 - 1) * 0.3

    def run(num, base, a):
        e = ProductImpl(kind, num, trk, seed=2024, env_id_base=base).env
        e.max_steps = 25
        e.reset_device()
        rews, dones = [], []
        for k in range(K):
            st, r, d, _ = e.step_device(a[k].contiguous())
            rews.append(r.clone()); dones.append(d.clone())
        return torch.stack(rews), torch.stack(dones), st.clone(), e

    r1, d1, s1, _ = run(n, 0, acts)
    r2, d2, s2, _ = run(n, 0, acts)
    assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(s1, s2)
    assert int(d1.sum()) >= n  # every env finished at least once (max_steps = 25)
    h = n // 2
    ra, da, sa, _ = run(h, 0, acts[:, :h])
    rb, db, sb, _ = run(h, h, acts[:, h:])
    assert torch.equal(torch.cat([ra, rb], 1), r1) and torch.equal(torch.cat([da, db], 1), d1)
    assert torch.equal(torch.cat([sa, sb], 0), s1)
    e = ProductImpl(kind, n, trk, seed=2024).env
    e.max_steps = 25
    e.reset_device()
    rm, dm, sm = e.rollout_device(acts)
    assert torch.equal(rm, r1) and torch.equal(dm, d1) and torch.equal(sm, s1)
    assert torch.isfinite(s1).all(dim=1).float().mean() > 0.999   # (a NaN state persists until max_steps, as upstream)


@pytest.mark.parametrize("kind", ["hover", "gates"])
@pytest.mark.parametrize("n", [1000, 4096])
def test_rollout_with_per_step_rows_equals_k_steps(kind, n):
    """q3_rollout (round 6): the fused K-step kernel WITH what a trainer consumes -- env.states after every step [K][N][16], rewards,
    dones and truncation flags -- equals K x q3_step bit for bit, through auto-resets, for a ragged env count (1 000: part-filled last
    workgroup and wave) and a full one; and leaves the handle in the same state (the next step agrees too)."""
    trk = pq.gates_track()
    K = 40
    dev = torch.device("cuda", 0)
    acts = (torch.rand((K, n, 4), device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 2 - 1) * 0.3
    e = ProductImpl(kind, n, trk, seed=77).env
    e.max_steps = 25
    e.reset_device()
    rows = []
    for k in range(K):
        rows.append([t.clone() for t in e.step_device(acts[k].contiguous())])
    nxt = [t.clone() for t in e.step_device(acts[0].contiguous())]
    f = ProductImpl(kind, n, trk, seed=77).env
    f.max_steps = 25
    f.reset_device()
    st, rew, done, trunc = f.rollout_states_device(acts)
    for k in range(K):
        for got, want in zip((st[k], rew[k], done[k], trunc[k]), rows[k]):
            assert torch.equal(got, want), (kind, n, k)
    assert int(done.sum()) >= n and int(trunc.sum()) > 0          # every env finished at least once; time limits were hit
    for got, want in zip(f.step_device(acts[0].contiguous()), nxt):
        assert torch.equal(got, want)


def test_vecenv_surface_matches_reference_classes():
    from optimal_quad_control_rl_amd import Quadcopter3DVec, Quadcopter3DVecGates

    gp, gy, sp = pq.gates_track()
    env = Quadcopter3DVecGates(64, gp, gy, sp)
    obs = env.reset()
    assert obs.shape == (64, 16) and obs.dtype == np.float32
    assert env.num_gates == 8 and env.max_steps == 1000 and env.dt == 0.01
    assert env.target_gates.dtype.kind == "i" and env.step_counts.dtype == np.float32
    env.max_steps = 3
    infos = None
    for k in range(3):
        obs, rew, done, infos = env.step(np.zeros((64, 4), np.float32))
    assert done.all() and rew.dtype == np.float32 and done.dtype == bool
    assert infos[0] is infos[63] and infos[0]["TimeLimit.truncated"] is True     # the shared dict of `[{}] * N`
    assert infos[0]["terminal_observation"].shape == (16,)
    r = env.render()
    assert set(r) == {'x', 'y', 'z', 'vx', 'vy', 'vz', 'phi', 'theta', 'psi', 'p', 'q', 'r', 'w1', 'w2', 'w3', 'w4',
                      'u1', 'u2', 'u3', 'u4'} and np.allclose(r['u1'], 0.5)
    assert env.env_is_wrapped(None) == [False] * 64 and env.get_attr("x") is None
    st = env.states
    st[:, 2] = 0.5                      # z > 0: ground collision is evaluated on the PRE-step state
    env.states = st
    _, rew, done, _ = env.step(np.zeros((64, 4), np.float32))
    assert done.all() and np.all(rew == -10)
    hov = Quadcopter3DVec(32)
    obs = hov.reset()
    assert obs.dtype == np.float64 and obs.shape == (32, 16) and hov.step_counts.dtype == np.float64
    assert hov.pos_threshold == 0.3 and abs(hov.ang_threshold - 10 * np.pi / 180) < 1e-15
    hov.states = np.zeros((32, 16))
    hov.pos_threshold = 0.0             # the goal can no longer be reached: reward is the small negative shaping term
    _, rew, done, _ = hov.step(np.zeros((32, 4), np.float32))
    assert rew.dtype == np.float64 and not done.any() and np.all(rew < 0) and np.all(rew > -0.01)
    hov.pos_threshold = 0.3
    hov.states = np.zeros((32, 16))
    _, rew, done, _ = hov.step(np.zeros((32, 4), np.float32))
    assert done.all() and np.all(rew == 100)
    with pytest.raises(ValueError):
        hov.step_device(torch.zeros((31, 4), device=hov.device))
    with pytest.raises(Exception):
        Quadcopter3DVecGates(4, np.zeros((40, 3)), np.zeros(40), np.zeros(3))   # more than 32 gates
