"""HIP product vs the CPU oracle on identical seeded inputs, and size-independent properties at BASELINE sizes.

Because product and oracle implement the same Philox4x32-10 reset specification, freshly reset lanes must agree
BIT-FOR-BIT; integrated lanes agree within the float32 step tolerance (different FMA contraction / sincos)."""
import numpy as np
import pytest
import torch

import parity as P

pytestmark = pytest.mark.gpu

E2E, INDI = 0, 1
FULL_N = 65536  # BASELINE.json configs 2/3


@pytest.fixture(scope="module")
def PA():
    assert torch.cuda.is_available()
    from product_adapter import ProductAdapter

    return ProductAdapter


@pytest.fixture(scope="module")
def OA():
    from oracle_adapter import OracleAdapter

    return OracleAdapter


def _pair(PA, OA, variant, n, tname, ga, blob, seed=11, env_id_base=0, ranges=P.TRAIN_DIST_RANGES):
    trk = P.tracks()[tname]
    kw = dict(gates_ahead=ga, residual=blob if variant == E2E else None, dist_ranges=ranges if variant == E2E else None,
              seed=seed, env_id_base=env_id_base)
    return PA(variant, n, trk, **kw), OA(variant, n, trk, **kw)


@pytest.mark.parametrize("variant", [E2E, INDI])
def test_reset_bit_exact_vs_oracle(PA, OA, variant, residual_blob):
    g, o = _pair(PA, OA, variant, 5000, "zigzag", 1, residual_blob, seed=0xDEADBEEFCAFE, env_id_base=(1 << 32) - 77)
    og, oo = g.reset(), o.reset()
    wg, dg, tg, sg = g.get_state()
    wo, do, to, so = o.get_state()
    np.testing.assert_array_equal(wg, wo)
    if variant == E2E:
        np.testing.assert_array_equal(dg, do)
    assert P.obs_err(og, oo).max() < P.TOL_STEP_OBS
    # masked reset: only masked envs change, and they draw the NEXT episode of their own stream
    mask = (np.arange(5000) % 3 == 0)
    g.reset(mask.astype(np.uint8)); o.reset(mask.astype(np.uint8))
    wg2, wo2 = g.get_state()[0], o.get_state()[0]
    np.testing.assert_array_equal(wg2, wo2)
    np.testing.assert_array_equal(wg2[~mask], wg[~mask])
    assert (wg2[mask] != wg[mask]).any(axis=1).all()


@pytest.mark.parametrize("variant,tname,ga", [(E2E, "zigzag", 1), (E2E, "square", 2), (INDI, "square", 1), (INDI, "zigzag", 0)])
def test_lockstep_vs_oracle(PA, OA, variant, tname, ga, residual_blob):
    """Teacher-forced lock-step at N=4096 with random actions, through auto-resets."""
    n, K = 4096, 60
    g, o = _pair(PA, OA, variant, n, tname, ga, residual_blob)
    g.env.max_steps = 40            # every env is truncated (and auto-reset) inside the window
    o.env.set_limits(40, 0.01)
    g.reset(); o.reset()
    rng = np.random.default_rng(5)
    tot_done = 0
    for k in range(K):
        wo, do, to, so = o.get_state()
        g.set_state(wo, do if variant == E2E else None, to, so)
        g.env.set_state_tensors(episode=o.env.episode.astype(np.int64))
        a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
        if k % 2:  # bias half of the steps towards flying (fewer resets, more gate logic)
            a = (0.124 + 0.3 * a).astype(np.float32) if variant == E2E else (0.2 * a + [0, 0, 0, 0.22]).astype(np.float32)
        og, rg, dng, trg = g.step(a)
        oo, ro, dno, tro = o.step(a)
        mism = dng != dno
        assert mism.sum() <= 1, f"step {k}: {mism.sum()} done mismatches"  # knife-edge threshold cases only
        ok = ~mism
        wg, dg, tg, sg = g.get_state()
        wo2, do2, to2, so2 = o.get_state()
        np.testing.assert_array_equal(tg[ok], to2[ok])
        np.testing.assert_array_equal(sg[ok], so2[ok])
        np.testing.assert_array_equal(trg, tro)
        assert np.abs(rg[ok] - ro[ok]).max() < P.TOL_STEP_REWARD
        done = dno & ok
        live = ~dno & ok
        np.testing.assert_array_equal(wg[done], wo2[done])  # freshly reset lanes: bit exact
        if variant == E2E:
            np.testing.assert_array_equal(dg[done], do2[done])
        if live.any():
            assert P.rel_err(wg[live], wo2[live]).max() < P.TOL_STEP_STATE
        assert P.obs_err(og[ok], oo[ok], wo2[ok]).max() < P.TOL_STEP_OBS
        tot_done += int(dno.sum())
    assert tot_done >= n


@pytest.mark.parametrize("variant", [E2E, INDI])
def test_full_size_one_step_vs_oracle(PA, OA, variant, residual_blob):
    """BASELINE size N=65536: one reset + three steps, product vs oracle."""
    n = FULL_N
    g, o = _pair(PA, OA, variant, n, "zigzag" if variant == E2E else "square", 1, residual_blob, seed=3)
    g.reset(); o.reset()
    np.testing.assert_array_equal(g.get_state()[0], o.get_state()[0])
    rng = np.random.default_rng(9)
    for k in range(3):
        a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
        og, rg, dng, _ = g.step(a)
        oo, ro, dno, _ = o.step(a)
        ok = dng == dno
        assert (~ok).sum() <= 2
        wg, wo = g.get_state()[0], o.get_state()[0]
        live = ok & ~dno
        # free-running (no re-injection): error compounds over the three steps
        assert P.rel_err(wg[live], wo[live]).max() < 4 * P.TOL_STEP_STATE
        assert np.abs(rg[ok] - ro[ok]).max() < 4 * P.TOL_STEP_REWARD


# ---- size-independent properties at full size ------------------------------------------------------------------------
def _run(env, actions):
    outs = []
    for k in range(actions.shape[0]):
        o, r, d, t = env.step_device(actions[k])
        outs.append((o.clone(), r.clone(), d.clone(), t.clone()))
    return outs


@pytest.mark.parametrize("variant", [E2E, INDI])
def test_determinism_sharding_and_rollout_equivalence(variant):
    """(a) same seed -> bitwise identical; (b) two half-size shards with env_id_base offsets == one full env
    (the multi-GPU partition); (c) qr_step_many == K x qr_step (bitwise)."""
    from optimal_quad_control_rl_amd import Quadcopter3DGates, Quadcopter3DGatesINDI, square_track, zigzag_track
    from optimal_quad_control_rl_amd import TRAIN_DISTURBANCE_RANGES

    cls = Quadcopter3DGates if variant == E2E else Quadcopter3DGatesINDI
    trk = zigzag_track() if variant == E2E else square_track()
    n, K = FULL_N, 40

    def make(num, base):
        e = cls(num, *trk, gates_ahead=1, seed=2024, env_id_base=base, infos_mode="none")
        if variant == E2E:
            e.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
        e.max_steps = 25  # force truncation resets inside the window
        e.reset_device()
        return e

    gen = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.rand((K, n, 4), device="cuda", generator=gen) * 2 - 1
    full_a, full_b, full_c = make(n, 0), make(n, 0), make(n, 0)
    ra, rb = _run(full_a, acts), full_b.rollout_device(acts)  # K x qr_step  vs  fused qr_step_many
    rc = full_c.step_sequence_device(acts, tuple(torch.empty_like(t) for t in rb))  # qr_step_launches
    for k in range(K):
        for j in range(4):
            assert torch.equal(ra[k][j], rb[j][k]), (k, j)
            assert torch.equal(ra[k][j], rc[j][k]), (k, j)
    for sa, sb in zip(full_a.get_state_tensors(), full_b.get_state_tensors()):  # final internal state as well
        assert sa is None or torch.equal(sa, sb)
    lo, hi = make(n // 2, 0), make(n // 2, n // 2)
    rl, rh = _run(lo, acts[:, : n // 2].contiguous()), _run(hi, acts[:, n // 2:].contiguous())
    for k in range(K):
        for j in range(4):
            assert torch.equal(ra[k][j][: n // 2], rl[k][j]), (k, j)
            assert torch.equal(ra[k][j][n // 2:], rh[k][j]), (k, j)
    dones = torch.stack([r[2] for r in ra]).sum().item()
    assert dones >= n  # every env was reset at least once (max_steps = 25)


@pytest.mark.parametrize("variant", [E2E, INDI])
def test_invariants_full_size(variant):
    from optimal_quad_control_rl_amd import Quadcopter3DGates, Quadcopter3DGatesINDI, zigzag_track
    from optimal_quad_control_rl_amd import TRAIN_DISTURBANCE_RANGES

    cls = Quadcopter3DGates if variant == E2E else Quadcopter3DGatesINDI
    gp, gy, sp = zigzag_track()
    n = FULL_N
    env = cls(n, gp, gy, sp, gates_ahead=1, seed=1, infos_mode="none")
    if variant == E2E:
        env.disturbance_ranges = TRAIN_DISTURBANCE_RANGES
    env.reset_device()
    gen = torch.Generator(device="cuda").manual_seed(1)
    prev_steps = env.get_state_tensors()[3].clone()
    n_done = 0
    for k in range(150):
        a = torch.rand((n, 4), device="cuda", generator=gen) * 2 - 1
        obs, rew, done, trunc = env.step_device(a)
        w, dist, tgt, steps, ep = env.get_state_tensors()
        d = done.bool()
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert (steps[d] == 0).all() and (tgt[d] == 0).all()
        assert (steps[~d] == prev_steps[~d] + 1).all()
        assert ((tgt >= 0) & (tgt < env.num_gates)).all()
        assert (rew <= 10.0).all() and (rew >= -10.0).all()
        # freshly reset envs sit inside the reset box around start_pos (R:455-457)
        if d.any():
            box = (w[d][:, 0:3] - torch.tensor(sp, device="cuda", dtype=torch.float32)).abs().max()
            assert box <= 0.5 + 1e-6
            assert (ep[d] >= 2).all()
        # obs is the gate-frame transform of the stored state: z' = z - gate_z (R:383), rates/rpms copied
        gz = torch.tensor(gp[:, 2], device="cuda", dtype=torch.float32)[tgt.long()]
        assert torch.equal(obs[:, 2], w[:, 2] - gz)
        assert torch.equal(obs[:, 9:env.STATE_LEN], w[:, 9:])
        prev_steps = steps.clone()
        n_done += int(d.sum())
    assert n_done > n // 4  # random actions crash often (INDI is tamer than the motor-level model)


@pytest.mark.parametrize("n", [1, 63, 255, 257, 1000])
@pytest.mark.parametrize("variant", [E2E, INDI])
def test_ragged_sizes_vs_oracle(PA, OA, variant, n, residual_blob):
    g, o = _pair(PA, OA, variant, n, "square", 1, residual_blob, seed=n)
    g.reset(); o.reset()
    np.testing.assert_array_equal(g.get_state()[0], o.get_state()[0])
    rng = np.random.default_rng(n)
    a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
    og, rg, dng, _ = g.step(a)
    oo, ro, dno, _ = o.step(a)
    np.testing.assert_array_equal(dng, dno)
    assert P.obs_err(og, oo).max() < P.TOL_STEP_OBS
    assert og.shape == (n, g.env.state_len)


@pytest.mark.parametrize("ga", [0, 1, 2, 3, 4])
def test_gates_ahead_and_single_gate_track(PA, OA, ga, residual_blob):
    """gates_ahead larger than the track (wrap-around indexing, R:406-412) incl. a 1-gate track."""
    for G in (1, 2, 7):
        gp, gy, sp = P.tracks()["zigzag"]
        trk = (gp[:G], gy[:G], sp)
        g, o = PA(E2E, 300, trk, gates_ahead=ga, residual=residual_blob, seed=1), \
            OA(E2E, 300, trk, gates_ahead=ga, residual=residual_blob, seed=1)
        g.reset(); o.reset()
        a = np.full((300, 4), 0.124, np.float32)
        for _ in range(3):
            og, rg, dng, _ = g.step(a)
            oo, ro, dno, _ = o.step(a)
        np.testing.assert_array_equal(dng, dno)
        assert og.shape == oo.shape == (300, 20 + 4 * ga)
        assert P.obs_err(og, oo).max() < 4 * P.TOL_STEP_OBS


def test_large_env_count_memory_and_speed():
    """1 Mi envs in one handle (HBM-resident state, ~300 MB incl. outputs): smoke for the 288 GB sizing."""
    from optimal_quad_control_rl_amd import Quadcopter3DGates, zigzag_track

    n = 1 << 20
    env = Quadcopter3DGates(n, *zigzag_track(), gates_ahead=1, infos_mode="none")
    env.reset_device()
    a = torch.zeros((n, 4), device="cuda")
    obs, rew, done, trunc = env.step_device(a)
    torch.cuda.synchronize()
    assert obs.shape == (n, 24) and torch.isfinite(obs).all()


def test_error_reporting(PA):
    from optimal_quad_control_rl_amd import _lib
    import ctypes as C

    L = _lib.load()
    cfg = _lib.QrConfig(0, 0, 1, 0, 0, 0, 0)
    h = C.c_void_p()
    assert L.qr_create(C.byref(cfg), C.byref(h)) == _lib.QR_E_INVALID and b"num_envs" in L.qr_last_error()
    cfg = _lib.QrConfig(0, 8, 9, 0, 0, 0, 0)
    assert L.qr_create(C.byref(cfg), C.byref(h)) == _lib.QR_E_INVALID
    cfg = _lib.QrConfig(0, 8, 1, 0, 0, 0, 0)
    assert L.qr_create(C.byref(cfg), C.byref(h)) == 0
    # stepping before a track is set is a state error, not a crash
    assert L.qr_step(h, None, None, None, None, None, None) == _lib.QR_E_STATE
    assert L.qr_set_residual(h, None, 5) == 0
    blob = (C.c_float * 10)()
    assert L.qr_set_residual(h, blob, 10) == _lib.QR_E_INVALID
    assert L.qr_destroy(h) == 0


def test_sb3_facing_step_wait_and_infos():
    """The NumPy / SB3-facing surface: shapes, dtypes, auto-reset semantics and both infos modes (R:589-595)."""
    from optimal_quad_control_rl_amd import Quadcopter3DGates, zigzag_track

    gp, gy, sp = zigzag_track()
    n = 300
    for mode in ("reference", "per_env"):
        env = Quadcopter3DGates(n, gp, gy, sp, gates_ahead=1, seed=3, infos_mode=mode)
        env.max_steps = 7
        obs = env.reset()
        assert obs.shape == (n, 24) and obs.dtype == np.float32
        assert env.observation_space.shape == (24,) and env.action_space.shape == (4,)
        for k in range(7):
            a = np.random.default_rng(k).uniform(-1, 1, size=(n, 4)).astype(np.float32)
            obs, rew, done, infos = env.step(a)
        assert obs.dtype == np.float32 and rew.dtype == np.float32 and done.dtype == bool and len(infos) == n
        assert done.all()  # max_steps = 7: everything truncated at the 7th step, already reset
        assert (env.step_counts == 0).all() and np.allclose(env.states, obs)
        if mode == "reference":
            assert infos[0] is infos[-1] and infos[0]["TimeLimit.truncated"] is True
            np.testing.assert_array_equal(infos[0]["terminal_observation"], obs[-1])
        else:
            assert infos[0] is not infos[1] and all(i["TimeLimit.truncated"] for i in infos)
        r = env.render()
        assert set(r) == {'x', 'y', 'z', 'vx', 'vy', 'vz', 'phi', 'theta', 'psi', 'p', 'q', 'r', 'w1', 'w2', 'w3', 'w4',
                          'u1', 'u2', 'u3', 'u4'} and r['x'].shape == (n,)
        with pytest.raises(AttributeError):
            env.get_attr("render_mode")
        assert env.env_is_wrapped(None) == [False] * n
        assert env.update_states_world().shape == (n, 16)
        env.close()


def test_ppo_consumes_device_tensors_and_improves():
    """BASELINE config 5 smoke: the on-device PPO (reference hyper-parameters' algorithm) runs on the env's device tensors
    and learns within a few iterations -- the first thing PPO learns here is not to crash (episodes get longer)."""
    from optimal_quad_control_rl_amd import Quadcopter3DGatesINDI, square_track
    from optimal_quad_control_rl_amd.ppo import PPO

    env = Quadcopter3DGatesINDI(8192, *square_track(), gates_ahead=1, infos_mode="none", seed=1)
    model = PPO(env, n_steps=64, n_epochs=8, batch_size=8192 * 64 // 16, learning_rate=1e-3, seed=0)
    model.collect()
    first = dict(model.stats)
    model.train()
    model.learn(8192 * 64 * 40, log_every=0)
    last = model.stats
    assert model.num_timesteps >= 8192 * 64 * 40
    assert last["ep_len_mean"] > 1.5 * first["ep_len_mean"], (first, last)
    assert last["reward_per_step"] > first["reward_per_step"]
    a = model.act_device(env.states_tensor)
    assert a.shape == (8192, 4) and float(a.abs().max()) <= 1.0


@pytest.mark.parametrize("variant", [E2E, INDI])
def test_limits_setters_vs_oracle(PA, OA, variant, residual_blob):
    """env.dt / env.max_steps writes (I:648) reach the kernels: compare against the oracle with the same limits."""
    n = 512
    g, o = _pair(PA, OA, variant, n, "square", 1, residual_blob, seed=21)
    g.env.dt = 0.004
    g.env.max_steps = 5
    o.env.set_limits(5, 0.004)
    g.reset(); o.reset()
    rng = np.random.default_rng(3)
    total_done = 0
    for k in range(6):
        a = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
        og, rg, dng, trg = g.step(a)
        oo, ro, dno, tro = o.step(a)
        np.testing.assert_array_equal(dng, dno)
        np.testing.assert_array_equal(trg, tro)
        assert P.obs_err(og, oo).max() < 4 * P.TOL_STEP_OBS
        total_done += int(dno.sum())
    assert total_done == n and float(g.env.dt) == np.float32(0.004) and g.env.max_steps == 5


def test_non_finite_actions_do_not_crash_and_stay_contained():
    """A NaN action poisons only its own env (no cross-lane effects through the wave-wide MLP / reset paths), and the
    reference's behaviour is kept: NaN comparisons are False, so the env lives until max_steps."""
    from optimal_quad_control_rl_amd import Quadcopter3DGates, zigzag_track

    n = 256
    env = Quadcopter3DGates(n, *zigzag_track(), gates_ahead=1, seed=2, infos_mode="none")
    ref = Quadcopter3DGates(n, *zigzag_track(), gates_ahead=1, seed=2, infos_mode="none")
    env.max_steps = 12
    ref.max_steps = 12
    env.reset_device(); ref.reset_device()
    a = torch.full((n, 4), 0.1, device="cuda")
    bad = a.clone()
    bad[37] = float("nan")
    for k in range(12):
        o1, r1, d1, t1 = env.step_device(bad)
        o2, r2, d2, t2 = ref.step_device(a)
        keep = torch.ones(n, dtype=torch.bool, device="cuda")
        keep[37] = False
        assert torch.equal(o1[keep], o2[keep]) and torch.equal(r1[keep], r2[keep]) and torch.equal(d1[keep], d2[keep])
        if k < 11:
            assert not bool(d1[37]) and torch.isnan(o1[37]).any()
    assert bool(d1[37]) and bool(t1[37])  # ended by the time limit, then reset to a finite state
    assert torch.isfinite(o1[37]).all()
