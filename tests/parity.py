"""Shared parity machinery: replays the committed reference trajectories (tests/golden, generated from the
real reference by tools/gen_golden.py) on any implementation of the env -- the CPU oracle in the
`not gpu` suite and the HIP product in the `gpu` suite -- so both are held to the same bar.

An implementation is wrapped as an object with:
    set_state(world[N,S], dist[N,6] or None, target[N], steps[N]);  get_state() -> (world, dist, target, steps)
    step(actions[N,4]) -> (obs[N,L], reward[N], done[N] bool, trunc[N] bool);  observe() -> obs
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAIN_DIST_RANGES = np.array([[-0.03, 0.03], [-0.03, 0.03], [-0.01, 0.01], [0, 0], [0, 0], [-0.5, 0.5]], np.float32)

# Tolerances (float32 path; BASELINE.json north_star: 1e-5 relative on state vectors).
#   one teacher-forced step: the reference's own lambdified expression has ~1e-5 absolute rounding noise
#   in the angular accelerations (cancellation of +-100-sized terms), i.e. ~1e-7 on a state after *dt.
TOL_STEP_STATE = 2e-6   # |d state| / max(1,|state|) after ONE step from identical state
TOL_STEP_OBS = 8e-6   # gate-frame positions reach ~10 m: a rotation (2 products + add) is a few ulp(10)=9.5e-7
TOL_STEP_REWARD = 2e-5  # absolute; rewards are differences of O(1..10) distances
TOL_FREE_RUN = 1e-5     # north_star: free-running 100 steps, max |d state| / max(1,|state|)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def tracks():
    d = load("tracks")
    return {t: (d[t + "_gate_pos"], d[t + "_gate_yaw"], d[t + "_start_pos"]) for t in ("zigzag", "square")}


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


def obs_err(a, b, world=None):
    """Observation error.  Columns 0..5 are 2-D rotations of (pos - gate, vel) into the gate frame: a small
    component can come from cancellation of large operands, so its rounding noise scales with the length of the
    rotated vector (|obs[0:2]|, |obs[3:5]| are rotation invariant), not with the component itself.  Column 8 is
    the world yaw wrapped into (-pi, pi]: it inherits the ABSOLUTE rounding of the unwrapped yaw (pass `world`
    when the unwrapped yaw can be large), and is compared on the circle."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    err[..., 8] = np.abs((a[..., 8] - b[..., 8] + np.pi) % (2 * np.pi) - np.pi)
    scale = np.maximum(1.0, np.abs(b))
    scale[..., 0:2] = np.maximum(1.0, np.linalg.norm(b[..., 0:2], axis=-1, keepdims=True))
    scale[..., 3:5] = np.maximum(1.0, np.linalg.norm(b[..., 3:5], axis=-1, keepdims=True))
    if world is not None:
        scale[..., 8] = np.maximum(1.0, np.abs(np.asarray(world, np.float64)[..., 8]))
    return err / scale


class TrajectoryReport:
    def __init__(self):
        self.max_state = 0.0
        self.max_obs = 0.0
        self.max_reward = 0.0
        self.steps = 0
        self.dones = 0
        self.passes = 0

    def __repr__(self):
        return (f"steps={self.steps} dones={self.dones} passes={self.passes} max_state={self.max_state:.3g} "
                f"max_obs={self.max_obs:.3g} max_reward={self.max_reward:.3g}")


def teacher_forced(env, traj, prefix, has_dist, tol_state=TOL_STEP_STATE, tol_obs=TOL_STEP_OBS,
                   tol_rew=TOL_STEP_REWARD, stride=1):
    """Per step: inject the reference's state k, step with action k, compare with the reference's state k+1."""
    g = lambda k: traj[prefix + k]
    acts = g("actions")
    H = acts.shape[0]
    world, target, steps = g("world0"), g("target0"), g("steps0")
    dist = g("dist0") if has_dist else None
    rep = TrajectoryReport()
    for k in range(0, H):
        if k % stride == 0:
            env.set_state(world, dist, target, steps)
            obs, rew, done, trunc = env.step(acts[k])
            w_new, d_new, t_new, s_new = env.get_state()
            ref_done = g("done")[k].astype(bool)
            ref_rew = g("reward")[k]
            np.testing.assert_array_equal(done, ref_done, err_msg=f"done mismatch at step {k}")
            rep.max_reward = max(rep.max_reward, float(np.abs(rew - ref_rew).max()))
            assert np.abs(rew - ref_rew).max() <= tol_rew, f"reward mismatch at step {k}: {rew} vs {ref_rew}"
            np.testing.assert_array_equal(t_new, g("target")[k], err_msg=f"target mismatch at step {k}")
            np.testing.assert_array_equal(s_new, g("steps")[k], err_msg=f"step_counts mismatch at step {k}")
            np.testing.assert_array_equal(trunc, (steps + 1) >= 1200, err_msg=f"trunc mismatch at step {k}")  # R:553
            live = ~ref_done
            if live.any():
                es = rel_err(w_new[live], g("world")[k][live]).max()
                eo = obs_err(obs[live], g("obs")[k][live]).max()
                rep.max_state, rep.max_obs = max(rep.max_state, float(es)), max(rep.max_obs, float(eo))
                assert es <= tol_state, f"state mismatch at step {k}: {es}"
                assert eo <= tol_obs, f"obs mismatch at step {k}: {eo}"
            rep.steps += 1
            rep.dones += int(ref_done.sum())
            rep.passes += int((ref_rew > 1).sum())
        world, target, steps = g("world")[k], g("target")[k], g("steps")[k]
        if has_dist:
            dist = g("dist")[k]
    return rep


def free_run(env, traj, prefix, has_dist, horizon=100, tol=TOL_FREE_RUN):
    """Inject the initial state once, then replay the recorded action sequence without correction."""
    g = lambda k: traj[prefix + k]
    acts = g("actions")
    assert not g("done")[:horizon].any(), "free-run window must not contain a reference reset"
    env.set_state(g("world0"), g("dist0") if has_dist else None, g("target0"), g("steps0"))
    rep = TrajectoryReport()
    for k in range(horizon):
        obs, rew, done, trunc = env.step(acts[k])
        w_new, _, t_new, s_new = env.get_state()
        es = rel_err(w_new, g("world")[k]).max()
        rep.max_state = max(rep.max_state, float(es))
        rep.max_obs = max(rep.max_obs, float(obs_err(obs, g("obs")[k]).max()))
        rep.max_reward = max(rep.max_reward, float(np.abs(rew - g("reward")[k]).max()))
        assert not done.any(), f"unexpected termination at free-run step {k}"
        np.testing.assert_array_equal(t_new, g("target")[k])
        rep.steps += 1
        rep.passes += int((g("reward")[k] > 1).sum())
    assert rep.max_state <= tol, f"free-run drift {rep.max_state} > {tol} over {horizon} steps"
    return rep



def residual_rate_allowance(world_row, blob, dt=0.01):
    """Upper bound of |d new body rate| (absolute, [3]) that the ROUNDING of the moment network can cause in one E2E step from state
    `world_row` (float64 evaluation): ours (both layer-1 operands as two f16 pieces: 2^-21 sum |w||x| per hidden pre-activation,
    quadrace_device.hpp residual_mlp; layer 2 and the bias adds are f32 fmaf chains) plus the reference's (float32 sgemm / addmm of an
    11-term and a 33-term dot product in an unspecified order: gamma_n = n 2^-24 / (1 - n 2^-24) of sum |w||x|), pushed through
    |W2| and the angular-acceleration gains of R:144-146."""
    b = np.asarray(blob, np.float64)
    W1, b1, W2, b2 = b[289:609].reshape(32, 10), b[609:641], b[641:737].reshape(3, 32), b[737:740]
    s = np.asarray(world_row, np.float64)
    sph, cph, sth, cth, sps, cps = np.sin(s[6]), np.cos(s[6]), np.sin(s[7]), np.cos(s[7]), np.sin(s[8]), np.cos(s[8])
    R = np.array([[cps * cth, cps * sth * sph - sps * cph, cps * sth * cph + sps * sph],
                  [sps * cth, sps * sth * sph + cps * cph, sps * sth * cph - cps * sph],
                  [-sth, cth * sph, cth * cph]])
    x = np.concatenate([s[12:16], R.T @ s[3:6], s[9:12]])
    mag1 = np.abs(W1) @ np.abs(x) + np.abs(b1)                       # sum |w||x| of every hidden pre-activation
    h = np.maximum(W1 @ x + b1, 0.0)
    u = 2.0 ** -24
    per_hidden = (2.0 ** -21 + 2 * 12 * u) * mag1                    # ours + the reference's layer 1
    out_err = np.abs(W2) @ per_hidden + 2 * 34 * u * (np.abs(W2) @ h + np.abs(b2))   # + both layer-2 chains
    gains = np.array([1103.7527593819, 805.152979066023, 486.854917234664])
    return dt * gains * out_err


def check_edges(env_factory, variant_id, vname, residual_blob):
    """F11 (tools/gen_golden.py gen_f11_edges): one step of the reference from states at the edges the other fixtures miss.
      * NaN / inf components: the NaN / inf PATTERN of the new state, the observation and the reward is the reference's, the finite
        components agree at the usual one-step tolerance, `done` is the reference's (a NaN env stays alive until max_steps);
      * body rates bracketing the 1000 rad/s guard: the rows one rad/s inside / outside must agree exactly; the two rows on
        consecutive float32 values around the reference's own flip may differ only if the new rate is within 1e-3 of the guard;
      * theta within 1e-3 of +-pi/2: the Euler kinematics multiply by 1/cos(theta) ~ 1e3-1e4, so phi and psi inherit the rounding of
        cos(theta) (7e-8 absolute for every float32 cosine, the reference's NumPy one included) amplified by that factor: they are
        compared with the tolerance scaled by 1/|cos theta|, everything else at the usual one;
      * |psi| ~ 1e4: usual tolerances (argument reduction of sin / cos, yaw wrap of the observation).
    Rows with a ~1000 rad/s rate feed the residual MLPs inputs a thousand times their usual size: the rounding of the moment network's
    output (this build's split layer: <= 2^-21 sum |w||x| per hidden unit; the reference's float32 sgemm: the standard gamma_n bound of
    its own chains, in ANOTHER summation order) reaches the angular accelerations through 1 / I = 1104 / 805 / 487 and a rate after
    dt.  The three rates (and their observation columns) of those rows get that bound, evaluated per row (residual_rate_allowance),
    ON TOP of the usual tolerance -- round 5's blanket 10 x is gone.  INDI has no residual network: usual tolerance."""
    d = load("f11_edges")
    names = [str(x) for x in d[vname + "_names"]]
    n = len(names)
    a = env_factory(variant_id, n, tracks()["zigzag"], gates_ahead=1, residual=residual_blob)
    w0 = d[vname + "_world0"]
    a.set_state(w0, np.zeros((n, 6), np.float32), d[vname + "_target0"], d[vname + "_steps0"])
    with np.errstate(all="ignore"):
        obs, rew, done, trunc = a.step(d[vname + "_actions"])
    w, _, t, s = a.get_state()
    ref_done = d[vname + "_done"].astype(bool)
    knife = np.array([nm.endswith("last_alive") or nm.endswith("first_oob") for nm in names])
    mism = done != ref_done
    assert not (mism & ~knife).any(), [names[i] for i in np.nonzero(mism & ~knife)[0]]
    for i in np.nonzero(mism)[0]:   # a knife-edge row: justified only by a new rate within 1e-3 of the guard
        rates = np.abs(np.asarray(d[vname + "_world"][i] if not ref_done[i] else w[i], np.float64)[9:12])
        assert np.abs(rates - 1000.0).min() < 1e-3, (names[i], rates)
    ok = ~mism
    np.testing.assert_array_equal(t[ok], d[vname + "_target"][ok])
    np.testing.assert_array_equal(s[ok], d[vname + "_steps"][ok])
    assert trunc[names.index("nan_x_at_max_steps")] and done[names.index("nan_x_at_max_steps")]
    # rewards: NaN where the reference's is NaN, equal otherwise
    ref_rew = d[vname + "_reward"]
    np.testing.assert_array_equal(np.isnan(rew[ok]), np.isnan(ref_rew[ok]))
    fin = ok & ~np.isnan(ref_rew)
    assert np.abs(rew[fin] - ref_rew[fin]).max() < TOL_STEP_REWARD
    live = ok & ~ref_done            # rows the reference did not reset
    ref_w, ref_o = d[vname + "_world"], d[vname + "_obs"]
    np.testing.assert_array_equal(np.isnan(w[live]), np.isnan(ref_w[live]))
    np.testing.assert_array_equal(np.isinf(w[live]), np.isinf(ref_w[live]))
    np.testing.assert_array_equal(np.isnan(obs[live]), np.isnan(ref_o[live]))
    tol_w = np.full(ref_w.shape, TOL_STEP_STATE)
    tol_o = np.full(ref_o.shape, TOL_STEP_OBS)
    for i, nm in enumerate(names):
        if nm.startswith("theta_"):
            amp = 1.0 / abs(np.cos(np.float64(w0[i, 7])))
            tol_w[i, 6] *= amp; tol_w[i, 8] *= amp
            tol_o[i, 6] *= amp; tol_o[i, 8] *= amp
        if nm.startswith("rate_") and residual_blob is not None and ref_w.shape[1] == 16:
            # body rates at the 1000 rad/s guard feed the moment network inputs a thousand times their usual size: no blanket
            # multiplier -- the PROVEN bound of the row (residual_rate_allowance), added to the usual tolerance
            extra = residual_rate_allowance(w0[i], residual_blob) / np.maximum(1.0, np.abs(np.asarray(ref_w[i, 9:12], np.float64)))
            tol_w[i, 9:12] += extra
            tol_o[i, 9:12] += extra
    fw = np.isfinite(ref_w) & live[:, None]
    assert (rel_err(np.where(fw, w, 0), np.where(fw, ref_w, 0)) <= tol_w).all(), \
        [(names[i], j) for i, j in zip(*np.nonzero(rel_err(np.where(fw, w, 0), np.where(fw, ref_w, 0)) > tol_w))]
    fo = np.isfinite(ref_o) & live[:, None]
    eo = obs_err(np.where(fo, obs, 0), np.where(fo, ref_o, 0), world=np.where(np.isfinite(ref_w), ref_w, 0))
    assert (eo <= tol_o).all(), [(names[i], j, eo[i, j]) for i, j in zip(*np.nonzero(eo > tol_o))]
    return dict(rows=n, knife_edge_mismatches=int(mism.sum()))
