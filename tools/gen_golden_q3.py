#!/usr/bin/env python3
"""Generate tests/golden/q3_*.npz from the REAL predecessor notebook ("3D quad.ipynb": f_func, Quadcopter3DVec,
Quadcopter3DVecGates), SURVEY.md section 8(f) #4.

Run in the build container (needs /root/reference + sympy):   python tools/gen_golden_q3.py

Only numeric input/output vectors are written.  Parity is defined on INJECTED states (the reference resets from
NumPy's global generator): every fixture stores the pre-step state, the actions, and what the reference returned.
Rows whose env finished carry a random post-reset state in the reference; they are flagged by `done` and the tests
compare only reward / done for them.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SCALE = np.array([5, 5, 5, 10, 10, 10, 1.2, 1.2, np.pi, 10, 10, 10, 1, 1, 1, 1])


def q3_track():
    """Track of Q3 cell 16 (inputs)."""
    gate_pos = np.array([[-1.5, -2, -1.5], [1.5, 2, -1.5], [1.5, -2, -1.5], [-1.5, 2, -1.5]] * 2, dtype=np.float64)
    gate_yaw = np.array([0, 0, np.pi, np.pi] * 2)
    start_pos = np.array([-4, -2, -1.5])
    return gate_pos, gate_yaw, start_pos


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)} B)")


def quiet_step(env, actions):
    with contextlib.redirect_stdout(io.StringIO()):
        env.step_async(actions)
        return env.step_wait()


def gen_ffunc(ns, rng):
    n = 1024
    s = rng.uniform(-1, 1, size=(n, 16)) * SCALE
    u = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
    d64 = ns["f_func"](s.T, u.T).T
    s32 = s.astype(np.float32)
    d32 = ns["f_func"](s32.T, u.T).T
    assert d64.dtype == np.float64 and d32.dtype == np.float32
    save("q3_ffunc", state64=s, state32=s32, control=u, dstate64=d64, dstate32=d32)


def gen_hover(ns, rng):
    out = {}
    # ---- one-step batch with every branch of Q3 cell 6 step_wait
    n = 512
    env = ns["Quadcopter3DVec"](n)
    s = rng.uniform(-1, 1, size=(n, 16)) * SCALE
    s[:, 0:3] *= 1.5                      # some rows start beyond |pos| > 10?  no: 7.5 max; out-of-bounds rows below
    steps = rng.integers(0, 900, size=n).astype(np.float64)
    u = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
    k = 0
    for j in range(24):                   # goal rows: everything small (thresholds 0.3 / 0.3 / 10 deg / 10 deg)
        s[k] = rng.uniform(-1, 1, 16) * np.array([.1, .1, .1, .1, .1, .1, .1, .1, .1, .1, .1, .1, .03, .03, .03, .03])
        u[k] = s[k, 12:16].astype(np.float32)
        k += 1
    for j in range(8):                    # near-goal rows failing exactly one criterion
        s[k] = 0
        s[k, [0, 3, 6, 9, 8, 11, 1, 4][j]] = [0.35, 0.35, 0.2, 0.2, 0.2, 0.2, -0.31, -0.31][j]
        u[k] = 0
        k += 1
    for j in range(16):                   # out of bounds: |pos| > 10 or |phi|,|theta| > pi (psi is NOT checked)
        s[k, [0, 1, 2, 6, 7][j % 5]] = [10.5, -10.5, 10.2, 3.3, -3.3][j % 5]
        k += 1
    for j in range(4):                    # |psi| > pi alone is in bounds
        s[k, 8] = [3.5, -3.5, 6.0, -6.0][j]
        k += 1
    for j in range(8):                    # max_steps (>= 1000 after the increment)
        steps[k] = [999, 1000, 998, 999, 1500, 999, 999, 997][j]
        k += 1
    env.states = s.copy()
    env.step_counts = steps.copy()
    obs, rew, done, infos = quiet_step(env, u)
    assert obs.dtype == np.float64 and rew.dtype == np.float64
    out.update(step_state0=s, step_steps0=steps.astype(np.int32), step_actions=u, step_state1=obs.copy(),
               step_reward=rew.copy(), step_done=done.astype(np.uint8), step_steps1=env.step_counts.astype(np.int32),
               step_any_truncated=np.array("TimeLimit.truncated" in infos[0]))
    print("   hover branches: done", int(done.sum()), "goal", int((rew == 100).sum()), "oob", int((rew == -1).sum()))
    # ---- free-running trajectories (no termination): small initial states, actions near the hover command
    n, H = 8, 400
    env = ns["Quadcopter3DVec"](n)
    s0 = rng.uniform(-1, 1, size=(n, 16)) * np.array([2, 2, 2, .5, .5, .5, .05, .05, 1, .02, .02, .02, .02, .02, .02, .02])
    s0[:, 0] += 3.0                       # keep away from the goal region
    env.states = s0.copy()
    t = np.arange(H)[:, None, None] * 0.01
    acts = (0.02 * np.sin(2 * np.pi * rng.uniform(0.5, 3, (1, n, 4)) * t + rng.uniform(0, 6.28, (1, n, 4)))).astype(np.float32)
    traj, rews = [], []
    for h in range(H):
        obs, rew, done, _ = quiet_step(env, acts[h])
        if done.any() or np.abs(obs[:, 6:8]).max() > 1.0:   # stop before the Euler-angle singularity amplifies rounding
            H = h
            break
        traj.append(obs.copy()); rews.append(rew.copy())
    print("   hover free run length", H)
    out.update(traj_state0=s0, traj_actions=acts[:H], traj_states=np.stack(traj), traj_rewards=np.stack(rews))
    save("q3_hover", **out)


def gen_gates(ns, rng):
    out = {}
    gp, gy, sp = q3_track()
    G = gp.shape[0]
    out.update(gate_pos=gp, gate_yaw=gy, start_pos=sp)
    n = 640
    env = ns["Quadcopter3DVecGates"](n, gp, gy, sp)
    s = (rng.uniform(-1, 1, size=(n, 16)) * SCALE).astype(np.float32)
    s[:, 2] = -np.abs(s[:, 2]) - 0.2       # above ground unless crafted
    tgt = rng.integers(0, G, size=n)
    steps = rng.integers(0, 900, size=n).astype(np.float32)
    u = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
    k = 0

    def put(target, off_along, off_lat, off_z, speed):
        """state just before gate `target`'s plane: `off_along` m before it along the normal, moving at `speed`"""
        nonlocal k
        c, sn = np.cos(gy[target]), np.sin(gy[target])
        s[k] = 0
        s[k, 0] = gp[target, 0] + off_along * c - off_lat * sn
        s[k, 1] = gp[target, 1] + off_along * sn + off_lat * c
        s[k, 2] = gp[target, 2] + off_z
        s[k, 3], s[k, 4] = speed * c, speed * sn
        tgt[k] = target
        steps[k] = 10
        u[k] = 0
        k += 1

    for target in range(G):
        put(target, -0.004, 0.0, 0.0, 1.0)        # clean pass (final gate for target = G-1 -> reward 10, done)
        put(target, -0.004, 0.3, -0.2, 1.0)       # pass off-centre
        put(target, -0.004, 0.7, 0.0, 1.0)        # collision (lateral)
        put(target, -0.004, 0.0, 0.6, 1.0)        # collision (vertical)
        put(target, -0.004, 0.0, 0.0, -1.0)       # moving away: no crossing
        put(target, 0.004, 0.0, 0.0, 1.0)         # already behind the plane: no crossing
    for j in range(8):                            # ground collision is tested on the PRE-step z
        s[k, 2] = [0.01, 0.5, -0.0001, 1e-6, 0.0, 2.0, -0.001, 0.2][j]
        k += 1
    for j in range(10):                           # out of bounds on the PRE-step state (no reward override)
        s[k, [0, 1, 9, 10, 11][j % 5]] = [10.5, -10.5, 1001.0, -1001.0, 1500.0][j % 5] * (1 if j < 5 else -1)
        k += 1
    for j in range(8):
        steps[k] = [999, 1000, 998, 999, 1500, 999, 999, 997][j]
        k += 1
    env.states = s.copy()
    env.step_counts = steps.copy()
    env.target_gates = tgt.copy()
    obs, rew, done, infos = quiet_step(env, u)
    assert obs.dtype == np.float32 and rew.dtype == np.float32, (obs.dtype, rew.dtype)
    out.update(step_state0=s, step_target0=tgt.astype(np.int32), step_steps0=steps.astype(np.int32), step_actions=u,
               step_state1=obs.copy(), step_reward=rew.copy(), step_done=done.astype(np.uint8),
               step_target1=env.target_gates.astype(np.int32), step_steps1=env.step_counts.astype(np.int32),
               step_any_truncated=np.array("TimeLimit.truncated" in infos[0]))
    print("   gates branches: done", int(done.sum()), "r=10", int((rew == 10).sum()), "r=-10", int((rew == -10).sum()))
    # ---- free-running trajectories
    n, H = 8, 300
    env = ns["Quadcopter3DVecGates"](n, gp, gy, sp)
    s0 = (rng.uniform(-1, 1, size=(n, 16)) * np.array([.3, .3, .3, .05, .05, .05, .05, .05, .3, .02, .02, .02, .02, .02, .02, .02])).astype(np.float32)
    s0[:, 0:3] += np.array([-3.0, -2.0, -1.5], np.float32)   # between the start and gate 0
    env.states = s0.copy()
    env.target_gates[:] = 0
    t = np.arange(H)[:, None, None] * 0.01
    acts = (0.02 * np.sin(2 * np.pi * rng.uniform(0.5, 3, (1, n, 4)) * t + rng.uniform(0, 6.28, (1, n, 4)))).astype(np.float32)
    traj, rews, tg = [], [], []
    for h in range(H):
        obs, rew, done, _ = quiet_step(env, acts[h])
        if done.any() or np.abs(obs[:, 6:8]).max() > 1.0:
            H = h
            break
        traj.append(obs.copy()); rews.append(rew.copy()); tg.append(env.target_gates.astype(np.int32).copy())
    print("   gates free run length", H, "targets at end", tg[-1])
    out.update(traj_state0=s0, traj_actions=acts[:H], traj_states=np.stack(traj), traj_rewards=np.stack(rews),
               traj_targets=np.stack(tg))
    save("q3_gates", **out)


def gen_reset_stats(ns):
    """Moments of the reference's reset distributions (the product's Philox reset is checked against these)."""
    out = {}
    n = 200000
    np.random.seed(7)
    env = ns["Quadcopter3DVec"](n)
    st = env.reset()
    out.update(hover_mean=st.mean(0), hover_std=st.std(0), hover_min=st.min(0), hover_max=st.max(0))
    gp, gy, sp = q3_track()
    env = ns["Quadcopter3DVecGates"](n, gp, gy, sp)
    st = env.reset().astype(np.float64)
    seg = env.target_gates.copy()
    pts = np.concatenate([sp[None], gp]).astype(np.float32)
    mid = ((pts[seg] + pts[seg + 1]) / 2).astype(np.float64)
    dev = st.copy()
    dev[:, 0:3] -= mid
    out.update(gates_dev_mean=dev.mean(0), gates_dev_std=dev.std(0), gates_dev_kurt=((dev - dev.mean(0)) ** 4).mean(0) / dev.var(0) ** 2,
               gates_segment_hist=np.bincount(seg, minlength=gp.shape[0]) / n, gates_w_min=st[:, 12:].min(0), gates_w_max=st[:, 12:].max(0))
    save("q3_reset_stats", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_import.load_q3()
    rng = np.random.default_rng(20240603)
    gen_ffunc(ns, rng)
    gen_hover(ns, rng)
    gen_gates(ns, rng)
    gen_reset_stats(ns)


if __name__ == "__main__":
    main()
