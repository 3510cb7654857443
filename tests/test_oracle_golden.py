"""Pins the TEST-ONLY CPU oracle (oracle/quadrace_oracle.c) against the reference.

Every fixture under tests/golden/ was produced by importing the real reference notebooks
(tools/gen_golden.py).  In addition the reference's own artefacts are used directly:
  * residual-MLP known answer stored in the notebook output (R:184+),
  * relative gate tables baked into c_code/nn_controller.c:40-60,
  * the reference's generated C for the MLPs, compiled by oracle/Makefile into oracle/_ref/.
"""
import numpy as np
import pytest

import parity as P
from oracle import oracle as O
from oracle_adapter import OracleAdapter


# ---- RNG spec ---------------------------------------------------------------------------------------
def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [hex(x) for x in O.philox([0, 0, 0, 0], [0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in O.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


# ---- F1: residual MLPs ------------------------------------------------------------------------------
def test_residual_known_answer_from_notebook_output(residual_blob):
    """R:248-267 prints thrust/moment for state row [0,1,...,15] (phi=theta=psi=0); stored output R:184+."""
    s = np.array([[0, 1, 2, 3, 4, 5, 0, 0, 0, 9, 10, 11, 12, 13, 14, 15]], np.float32)
    np.testing.assert_allclose(O.body_velocity(s)[0], [3, 4, 5], rtol=0, atol=0)
    thrust, moment = O.residual(residual_blob, s)
    np.testing.assert_allclose(thrust[0, 0], 36.098232, rtol=3e-7)
    np.testing.assert_allclose(moment[0], [0.2847767, -0.22512697, -0.05896095], rtol=0, atol=3e-7)


def test_residual_matches_fixture(residual_blob):
    d = P.load("f1_residual")
    thrust, moment = O.residual(residual_blob, d["states"])
    np.testing.assert_allclose(O.body_velocity(d["states"]), d["vb"], rtol=0, atol=4e-6)
    assert P.rel_err(thrust, d["thrust"]).max() < 2e-6
    assert P.rel_err(moment, d["moment"]).max() < 2e-6


def test_residual_matches_reference_generated_c(residual_blob):
    """oracle/_ref: the reference's own nn_thrust.c / nn_moment.c compiled from /root/reference/c_code."""
    import ctypes as C

    R = O.ref_residual_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    d = P.load("f1_residual")
    s = d["states"]
    vb = O.body_velocity(s)
    thrust, moment = O.residual(residual_blob, s)
    f32p = C.POINTER(C.c_float)
    for i in range(0, s.shape[0], 4):
        x = np.concatenate([s[i, 12:16], vb[i], s[i, 9:12]]).astype(np.float32)
        t = np.zeros(1, np.float32)
        m = np.zeros(3, np.float32)
        R.nn_thrust_forward(x[:7].ctypes.data_as(f32p), t.ctypes.data_as(f32p))
        R.nn_moment_forward(x.ctypes.data_as(f32p), m.ctypes.data_as(f32p))
        assert abs(t[0] - thrust[i, 0]) <= 2e-6 * max(1, abs(t[0]))
        assert np.abs(m - moment[i]).max() <= 2e-6


# ---- F2/F3: equations of motion ---------------------------------------------------------------------
def test_f_func_e2e():
    d = P.load("f2_ffunc_e2e")
    out = O.f_e2e(d["state"], d["control"], d["disturbance"])
    err = P.rel_err(out, d["dstate"])
    # columns 9,10 (p,q accelerations) carry ~1e-5 relative rounding noise in the reference itself
    assert err[:, [0, 1, 2, 12, 13, 14, 15]].max() == 0.0
    assert err[:, 3:9].max() < 4e-6
    assert err[:, 9:12].max() < 3e-5


def test_f_func_indi():
    d = P.load("f3_ffunc_indi")
    out = O.f_indi(d["state"], d["control"])
    err = P.rel_err(out, d["dstate"])
    assert err[:, [0, 1, 2, 9, 10, 11, 12]].max() == 0.0
    assert err.max() < 4e-6


# ---- track tables -----------------------------------------------------------------------------------
NN_CONTROLLER_POS_REL = np.array([  # c_code/nn_controller.c:40-49
    [2.8284265995025635, 2.82842755317688, 0.0], [2.1213202476501465, 2.1213202476501465, 0.0],
    [2.8284270763397217, 2.8284270763397217, 0.0], [2.1213202476501465, 2.1213204860687256, 0.0]] * 2, np.float32)
NN_CONTROLLER_YAW_REL = np.array([-4.71238899230957, 1.570796251296997, 1.570796251296997, 1.570796251296997] * 2,
                                 np.float32)  # c_code/nn_controller.c:51-60


def test_track_tables_pinned_by_reference_c_code():
    gp, gy, sp = P.tracks()["square"]
    env = OracleAdapter(O.INDI, 1, (gp, gy, sp)).env
    pr, yr = env.track_tables()
    np.testing.assert_allclose(pr, NN_CONTROLLER_POS_REL, rtol=0, atol=5e-7)
    np.testing.assert_allclose(yr, NN_CONTROLLER_YAW_REL, rtol=0, atol=5e-7)


@pytest.mark.parametrize("tname", ["zigzag", "square"])
def test_track_tables_fixture(tname):
    d = P.load("tracks")
    env = OracleAdapter(O.E2E, 1, P.tracks()[tname]).env
    pr, yr = env.track_tables()
    np.testing.assert_allclose(pr, d[tname + "_gate_pos_rel"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(yr, d[tname + "_gate_yaw_rel"], rtol=0, atol=5e-7)


# ---- F4: observation transform ----------------------------------------------------------------------
@pytest.mark.parametrize("tname", ["zigzag", "square"])
@pytest.mark.parametrize("ga", [0, 1, 2])
def test_obs_transform(tname, ga):
    d = P.load("f4_obs")
    trk = P.tracks()[tname]
    key = f"{tname}_e2e_ga{ga}"
    n = d[key + "_world"].shape[0]
    for rname, ranges in (("zero", np.zeros((6, 2), np.float32)), ("train", P.TRAIN_DIST_RANGES)):
        a = OracleAdapter(O.E2E, n, trk, gates_ahead=ga, dist_ranges=ranges)
        a.set_state(d[key + "_world"], d[key + f"_dist_{rname}"], d[key + "_target"], np.zeros(n, np.int32))
        obs = a.observe()
        assert obs.shape == d[key + f"_obs_{rname}"].shape
        assert P.obs_err(obs, d[key + f"_obs_{rname}"]).max() < 2e-6
    key = f"{tname}_indi_ga{ga}"
    a = OracleAdapter(O.INDI, n, trk, gates_ahead=ga)
    a.set_state(d[key + "_world"], None, d[key + "_target"], np.zeros(n, np.int32))
    obs = a.observe()
    assert obs.shape == d[key + "_obs"].shape
    assert P.obs_err(obs, d[key + "_obs"]).max() < 2e-6


# ---- F5: BASELINE config 1 (1 env, E2E, no residual, fixed action sequence) ---------------------------
@pytest.mark.parametrize("ga", [0, 1])
@pytest.mark.parametrize("tag", ["ctrl", "hover", "random"])
def test_config1_teacher_forced(ga, tag):
    traj = P.load("f5_traj_e2e_noresidual")
    a = OracleAdapter(O.E2E, 1, P.tracks()["zigzag"], gates_ahead=ga, residual=None)
    rep = P.teacher_forced(a, traj, f"ga{ga}_{tag}_", has_dist=True)
    print(tag, rep)
    assert rep.steps == traj[f"ga{ga}_{tag}_actions"].shape[0]


@pytest.mark.parametrize("tag", ["ctrl", "hover"])
def test_config1_free_run_100(tag):
    traj = P.load("f5_traj_e2e_noresidual")
    a = OracleAdapter(O.E2E, 1, P.tracks()["zigzag"], gates_ahead=1, residual=None)
    rep = P.free_run(a, traj, f"ga1_{tag}_", has_dist=True, horizon=100)
    print(tag, rep)


# ---- F6: E2E + residual + disturbances ----------------------------------------------------------------
@pytest.mark.parametrize("tname", ["zigzag", "square"])
def test_e2e_residual_teacher_forced(tname, residual_blob):
    traj = P.load("f6_traj_e2e_residual")
    n = traj[tname + "_world0"].shape[0]
    a = OracleAdapter(O.E2E, n, P.tracks()[tname], gates_ahead=1, residual=residual_blob,
                      dist_ranges=P.TRAIN_DIST_RANGES)
    rep = P.teacher_forced(a, traj, tname + "_", has_dist=True)
    print(tname, rep)
    assert rep.dones > 0 and rep.passes > 0


# ---- F7: branch known-answers -------------------------------------------------------------------------
@pytest.mark.parametrize("variant,vname", [(O.E2E, "e2e"), (O.INDI, "indi")])
def test_branches(variant, vname, residual_blob):
    d = P.load("f7_branches")
    n = d[vname + "_world0"].shape[0]
    a = OracleAdapter(variant, n, P.tracks()["zigzag"], gates_ahead=1, residual=residual_blob)
    a.set_state(d[vname + "_world0"], np.zeros((n, 6), np.float32), d[vname + "_target0"], d[vname + "_steps0"])
    obs, rew, done, trunc = a.step(d[vname + "_actions"])
    names = list(d[vname + "_names"])
    np.testing.assert_array_equal(done, d[vname + "_done"].astype(bool), err_msg=str(names))
    np.testing.assert_allclose(rew, d[vname + "_reward"], rtol=0, atol=2e-5)
    w, _, t, s = a.get_state()
    np.testing.assert_array_equal(t, d[vname + "_target"])
    np.testing.assert_array_equal(s, d[vname + "_steps"])
    assert trunc[names.index("max_steps")] and trunc.sum() == 1
    live = ~done
    assert P.rel_err(w[live], d[vname + "_world"][live]).max() < P.TOL_STEP_STATE
    assert P.obs_err(obs[live], d[vname + "_obs"][live]).max() < P.TOL_STEP_OBS
    # the reference's own numbers for the clean pass: reward 10 - 10*|0.005| = 9.95, target 0 -> 1
    i = names.index("pass_clean")
    assert abs(rew[i] - 9.95) < 1e-4 and t[i] == 1


# ---- F11: edge states (NaN / inf components, rates at the 1000 rad/s guard, theta at +-pi/2, |psi| ~ 1e4) ----------
@pytest.mark.parametrize("variant,vname", [(O.E2E, "e2e"), (O.INDI, "indi")])
def test_edge_states(variant, vname, residual_blob):
    print(P.check_edges(OracleAdapter, variant, vname, residual_blob))


# ---- F8: INDI trajectories ----------------------------------------------------------------------------
@pytest.mark.parametrize("key", ["zigzag", "square", "single"])
def test_indi_teacher_forced(key):
    traj = P.load("f8_traj_indi")
    n = traj[key + "_world0"].shape[0]
    trk = P.tracks()["square" if key == "single" else key]
    a = OracleAdapter(O.INDI, n, trk, gates_ahead=1)
    rep = P.teacher_forced(a, traj, key + "_", has_dist=False)
    print(key, rep)
    assert rep.passes > 0


def test_indi_free_run_100():
    traj = P.load("f8_traj_indi")
    a = OracleAdapter(O.INDI, 1, P.tracks()["square"], gates_ahead=1)
    rep = P.free_run(a, traj, "single_", has_dist=False, horizon=100)
    print(rep)


# ---- F9: pause_if_collision / pause modes ---------------------------------------------------------------
@pytest.mark.parametrize("variant,vname", [(O.E2E, "e2e"), (O.INDI, "indi")])
def test_modes(variant, vname, residual_blob):
    d = P.load("f9_modes")
    acts = d[vname + "_actions"]
    n = acts.shape[1]
    # NB: the E2E reference run used the real residual model
    a = OracleAdapter(variant, n, P.tracks()["zigzag"], gates_ahead=1, residual=residual_blob,
                      pause_if_collision=True)
    a.set_state(d[vname + "_world0"], d[vname + "_dist0"] if variant == O.E2E else None, d[vname + "_target0"],
                d[vname + "_steps0"])
    a.observe()
    pause_from = int(d[vname + "_pause_from_step"])
    for k in range(acts.shape[0]):
        if k == pause_from:
            a.set_pause(True)
        obs, rew, done, trunc = a.step(acts[k])
        w, _, t, s = a.get_state()
        np.testing.assert_array_equal(done, d[vname + "_done"][k].astype(bool), err_msg=f"step {k}")
        np.testing.assert_array_equal(t, d[vname + "_target"][k])
        np.testing.assert_array_equal(s, d[vname + "_steps"][k])
        assert np.abs(rew - d[vname + "_reward"][k]).max() < 1e-4, k
        # frozen envs keep their state, flying envs drift apart slowly (free run, unstable open loop)
        assert P.rel_err(w, d[vname + "_world"][k]).max() < 1e-4, k
        assert P.obs_err(obs, d[vname + "_obs"][k]).max() < 1e-4, k
    assert d[vname + "_done"][:pause_from].any() and not d[vname + "_done"][pause_from:].any()


# ---- reset distribution ---------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,vname", [(O.E2E, "e2e"), (O.INDI, "indi")])
def test_reset_distribution(variant, vname):
    """Reset ranges are the reference's (R:455-489 / I:270-296); the stream is this build's Philox spec."""
    stats = P.load("reset_stats")
    n = 20000
    a = OracleAdapter(variant, n, P.tracks()["zigzag"], gates_ahead=1, dist_ranges=P.TRAIN_DIST_RANGES, seed=123)
    a.reset()
    w, dist, t, s = a.get_state()
    assert (t == 0).all() and (s == 0).all()
    lo, hi = stats[vname + "_world_min"], stats[vname + "_world_max"]
    span = hi - lo
    assert (w.min(0) >= lo - 0.01 * span).all() and (w.max(0) <= hi + 0.01 * span).all()
    np.testing.assert_allclose(w.mean(0), stats[vname + "_world_mean"], atol=0.03 * span.max(), rtol=0)
    np.testing.assert_allclose(w.std(0), stats[vname + "_world_std"], rtol=0.03)
    if variant == O.E2E:
        assert (np.abs(dist[:, 3:5]) == 0).all()
        np.testing.assert_allclose(dist.std(0)[[0, 1, 2, 5]], stats["e2e_dist_std"][[0, 1, 2, 5]], rtol=0.03)
        assert (dist.min(0) >= P.TRAIN_DIST_RANGES[:, 0]).all() and (dist.max(0) <= P.TRAIN_DIST_RANGES[:, 1]).all()
    # every env gets its own stream, and a second reset draws fresh values
    assert len(np.unique(w[:, 0])) > 0.99 * n
    w2 = a.reset()
    assert not np.array_equal(a.get_state()[0], w)
