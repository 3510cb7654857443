"""Shared parity checks for the predecessor environments of "3D quad.ipynb" (Quadcopter3DVec, Quadcopter3DVecGates):
replays tests/golden/q3_*.npz (generated from the real notebook by tools/gen_golden_q3.py) on any implementation --
the CPU oracle in the `not gpu` suite, the HIP product in the `gpu` suite.

An implementation is a factory  make(kind, num_envs, track)  returning an object with
    set_state(states[N,16], target[N] or None, steps[N]);  step(actions[N,4]) -> (states, reward, done, trunc)
    get_state() -> (states, target, steps)
kind: "hover" (float64) or "gates" (float32); track = (gate_pos, gate_yaw, start_pos) for "gates".
"""
import numpy as np

from parity import load, rel_err

# float64 hover env: libm / device sin-cos-tan agree to an ulp or two; one Euler step keeps that at 1e-15 relative
TOL64_STEP = 1e-12
TOL64_FREE_RUN = 1e-9      # ~120 steps of an open-loop-unstable attitude amplify rounding noise
# float32 gates env: same budget as the race env (tests/parity.py)
TOL32_STEP_STATE = 2e-6
TOL32_STEP_REWARD = 2e-5
TOL32_FREE_RUN = 1e-4      # ~90 open-loop steps; the k_pv * v_y moment coupling makes the attitude diverge e-fold per ~0.2 s


def gates_track():
    d = load("q3_gates")
    return d["gate_pos"], d["gate_yaw"], d["start_pos"]


def check_hover_step(make):
    d = load("q3_hover")
    n = d["step_state0"].shape[0]
    env = make("hover", n, None)
    env.set_state(d["step_state0"], None, d["step_steps0"])
    st, rew, done, trunc = env.step(d["step_actions"])
    assert st.dtype == np.float64 and rew.dtype == np.float64
    ref_done = d["step_done"].astype(bool)
    assert np.array_equal(done, ref_done)
    live = ~ref_done
    assert rel_err(st[live], d["step_state1"][live]).max() <= TOL64_STEP
    assert np.abs(rew - d["step_reward"]).max() <= TOL64_STEP
    assert bool(trunc.any()) == bool(d["step_any_truncated"])
    _, _, steps = env.get_state()
    assert np.array_equal(steps[live], d["step_steps1"][live])
    assert np.all(steps[ref_done] == 0)
    # every branch is present in the fixture
    assert (d["step_reward"] == 100).sum() >= 20 and (d["step_reward"] == -1).sum() >= 10 and ref_done.sum() >= 40
    return env


def check_hover_free_run(make):
    d = load("q3_hover")
    H, n = d["traj_actions"].shape[:2]
    env = make("hover", n, None)
    env.set_state(d["traj_state0"], None, np.zeros(n, np.int32))
    worst = 0.0
    for h in range(H):
        st, rew, done, _ = env.step(d["traj_actions"][h])
        assert not done.any()
        worst = max(worst, rel_err(st, d["traj_states"][h]).max())
        assert np.abs(rew - d["traj_rewards"][h]).max() <= 1e-9
    assert worst <= TOL64_FREE_RUN, worst
    return worst


def check_gates_step(make):
    d = load("q3_gates")
    n = d["step_state0"].shape[0]
    env = make("gates", n, gates_track())
    env.set_state(d["step_state0"], d["step_target0"], d["step_steps0"])
    st, rew, done, trunc = env.step(d["step_actions"])
    assert st.dtype == np.float32 and rew.dtype == np.float32
    ref_done = d["step_done"].astype(bool)
    assert np.array_equal(done, ref_done)
    live = ~ref_done
    assert rel_err(st[live], d["step_state1"][live]).max() <= TOL32_STEP_STATE
    assert np.abs(rew.astype(np.float64) - d["step_reward"]).max() <= TOL32_STEP_REWARD
    assert bool(trunc.any()) == bool(d["step_any_truncated"])
    _, target, steps = env.get_state()
    assert np.array_equal(target[live], d["step_target1"][live])
    assert np.array_equal(steps[live], d["step_steps1"][live])
    assert (d["step_reward"] == 10).sum() >= 2 and (d["step_reward"] == -10).sum() >= 20
    assert (d["step_target1"][live] != d["step_target0"][live]).sum() >= 10    # gate passes that are not the final one
    return env


def check_gates_free_run(make):
    d = load("q3_gates")
    H, n = d["traj_actions"].shape[:2]
    env = make("gates", n, gates_track())
    env.set_state(d["traj_state0"], np.zeros(n, np.int32), np.zeros(n, np.int32))
    worst = 0.0
    for h in range(H):
        st, rew, done, _ = env.step(d["traj_actions"][h])
        assert not done.any()
        worst = max(worst, rel_err(st, d["traj_states"][h]).max())
        assert np.abs(rew.astype(np.float64) - d["traj_rewards"][h]).max() <= 1e-4
        assert np.array_equal(env.get_state()[1], d["traj_targets"][h])
    assert worst <= TOL32_FREE_RUN, worst
    return worst


def check_reset_distribution(make, n=200000):
    """The reset distributions are the reference's (moments from 200 000 reference resets, q3_reset_stats)."""
    r = load("q3_reset_stats")
    env = make("hover", n, None)
    st = env.reset()
    se = 4.0 / np.sqrt(n)
    assert np.all(np.abs(st.mean(0) - r["hover_mean"]) <= se * np.maximum(1, r["hover_std"]) * 2)
    assert np.allclose(st.std(0), r["hover_std"], rtol=0.01)
    assert np.all(st.min(0) >= np.floor(r["hover_min"])) and np.all(st.max(0) <= np.ceil(r["hover_max"]))
    assert np.all(env.get_state()[2] == 0)
    gp, gy, sp = gates_track()
    env = make("gates", n, (gp, gy, sp))
    st = env.reset().astype(np.float64)
    _, seg, steps = env.get_state()
    assert seg.min() == 0 and seg.max() == gp.shape[0] - 1 and np.all(steps == 0)
    assert np.allclose(np.bincount(seg, minlength=gp.shape[0]) / n, r["gates_segment_hist"], atol=0.004)
    pts = np.concatenate([sp[None], gp]).astype(np.float32)
    dev = st.copy()
    dev[:, 0:3] -= ((pts[seg] + pts[seg + 1]) / 2).astype(np.float64)
    assert np.all(np.abs(dev.mean(0) - r["gates_dev_mean"]) <= 2 * se * np.maximum(r["gates_dev_std"], 0.05))
    assert np.allclose(dev.std(0), r["gates_dev_std"], rtol=0.01)
    kurt = ((dev - dev.mean(0)) ** 4).mean(0) / dev.var(0) ** 2
    assert np.allclose(kurt[:12], 3.0, atol=0.08) and np.allclose(kurt[12:], 1.8, atol=0.03)   # normal / uniform
    assert np.all(np.abs(st[:, 12:]) <= 1.0)
