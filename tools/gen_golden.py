#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container (needs /root/reference + sympy + torch):

    python tools/gen_golden.py

The reference notebooks are imported through tools/ref_import.py (stubs for the
missing gymnasium / stable_baselines3 packages; nothing else is altered) and
driven with seeded inputs.  Only numeric input/output vectors are written --
never reference source.  The residual-MLP weights (NNDroneModel/*.pt, data) are
exported as a flat float32 blob in the nn_thrust.c / nn_moment.c order:
    thrust: W1[32][7] b1[32] W2[1][32] b2[1]   (289 floats)
    moment: W1[32][10] b1[32] W2[3][32] b2[3]  (451 floats)

Fixture ids follow SURVEY.md section 8(c): F1..F9 (+ track tables).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "optimal_quad_control_rl_amd", "data")

# ----------------------------------------------------------------------------
# Track / randomisation constants (inputs; values as in R:640-661, I:438-460, R:772-779)
# ----------------------------------------------------------------------------
def zigzag_track():
    l = 1
    gate_pos = np.array([[-3 * l, 0, -1.5], [-1 * l, 0, -1.5], [1 * l, 0, -1.5], [3 * l, 0, -1.5],
                         [1 * l, 0, -1.5], [-1 * l, 0, -1.5], [-3 * l, 0, -1.5]], dtype=np.float64)
    gate_yaw = np.array([np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2])
    start_pos = gate_pos[0] + np.array([0, -1.0, 0])
    return gate_pos, gate_yaw, start_pos


def square_track():
    gate_pos = np.array([[2, -1.5, -1.5], [2, 1.5, -1.5], [-2, 1.5, -1.5], [-2, -1.5, -1.5]] * 2, dtype=np.float64)
    gate_yaw = np.array([np.pi / 4, 3 * np.pi / 4, 5 * np.pi / 4, 7 * np.pi / 4] * 2)
    start_pos = gate_pos[3]
    return gate_pos, gate_yaw, start_pos


TRAIN_DIST_RANGES = np.array([[-0.03, 0.03], [-0.03, 0.03], [-0.01, 0.01], [0, 0], [0, 0], [-0.5, 0.5]])

TRACKS = {"zigzag": zigzag_track, "square": square_track}



# ----------------------------------------------------------------------------
# A crude waypoint controller (this build's own code, NOT from the reference): it only exists to
# produce action sequences that keep the reference simulator flying through gates, so that the
# committed trajectories exercise gate passes / long horizons.  The recorded action arrays are
# what the parity tests replay; the controller itself is never used by tests.
# ----------------------------------------------------------------------------
K_W=4.36301076e-08; K_P=1.4119331e-09; K_Q=1.21601884e-09; K_R1=2.57035545e-06
IXX,IYY,IZZ=0.000906,0.001242,0.002054

def desired_rates_thrust(ws, target_pt, psi_des=0.0, vmax=3.0):
    pos, vel = ws[:,0:3].astype(np.float64), ws[:,3:6].astype(np.float64)
    phi,theta,psi = ws[:,6].astype(np.float64), ws[:,7].astype(np.float64), ws[:,8].astype(np.float64)
    e = target_pt - pos
    vdes = 1.6*e
    n = np.linalg.norm(vdes,axis=1,keepdims=True); vdes = vdes*np.minimum(1, vmax/np.maximum(n,1e-9))
    a = 2.5*(vdes - vel)
    a = np.clip(a,-6,6)
    f = a - np.array([0,0,9.81])           # specific force to realise (world), points up (-z)
    T = np.linalg.norm(f,axis=1)
    zb = -f/T[:,None]                        # body z axis (down) in world
    c,s = np.cos(psi),np.sin(psi)
    zx =  c*zb[:,0] + s*zb[:,1]; zy = -s*zb[:,0] + c*zb[:,1]; zz = zb[:,2]
    phi_des = -np.arcsin(np.clip(zy,-0.6,0.6)); theta_des = np.arctan2(zx,zz)
    theta_des = np.clip(theta_des,-0.6,0.6)
    p_des = 7*(phi_des-phi); q_des = 7*(theta_des-theta)
    dpsi = (psi_des-psi+np.pi)%(2*np.pi)-np.pi
    r_des = 2*dpsi
    return p_des,q_des,r_des,T

def ctrl_indi(ws, tp):
    p,q,r,T = desired_rates_thrust(ws,tp)
    a = np.stack([p/3,q/3,r/2,T/8-1],axis=1)
    return np.clip(a,-1,1).astype(np.float32)

def ctrl_e2e(ws, tp):
    p_des,q_des,r_des,T = desired_rates_thrust(ws,tp)
    p,q,r = ws[:,9].astype(np.float64),ws[:,10].astype(np.float64),ws[:,11].astype(np.float64)
    Mx = IXX*14*(p_des-p); My = IYY*14*(q_des-q); Mz = IZZ*6*(r_des-r)
    base = T/(4*K_W)
    sr = np.array([1,-1,-1,1.]); sp = np.array([1,1,-1,-1.]); sy = np.array([-1,1,-1,1.])
    W2 = base[:,None] + sr[None]*Mx[:,None]/(4*K_P) + sp[None]*My[:,None]/(4*K_Q)
    W = np.sqrt(np.clip(W2,3000**2,11000**2)) + sy[None]*Mz[:,None]/(4*K_R1)
    u = (W-7000)/4000
    return np.clip(u,-1,1).astype(np.float32)

def waypoint(env, lead=0.6):
    g = env.target_gates % env.num_gates
    gp = env.gate_pos[g].astype(np.float64); gy = env.gate_yaw[g].astype(np.float64)
    return gp + lead*np.stack([np.cos(gy),np.sin(gy),0*gy],axis=1)



def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)} B)")


# ----------------------------------------------------------------------------
def export_weights(ns):
    blobs = []
    for model in (ns["thrust_model"], ns["moment_model"]):
        sd = model.state_dict()
        keys = list(sd.keys())
        assert len(keys) == 4, keys
        for k in keys:  # W1, b1, W2, b2 (torch Linear: weight[out][in])
            blobs.append(sd[k].detach().cpu().numpy().astype(np.float32).ravel())
    blob = np.concatenate(blobs)
    assert blob.size == 740, blob.size
    os.makedirs(DATA, exist_ok=True)
    path = os.path.join(DATA, "residual_mlp_f32.bin")
    blob.tofile(path)
    print(f"  wrote {os.path.relpath(path, ROOT)} ({blob.size} f32)")
    return blob


def gen_f1_residual(ns, rng):
    """F1: residual MLP known-answer (R:248-267 stored output) + 256 random rows."""
    states = np.zeros((257, 16), dtype=np.float32)
    states[0] = [0, 1, 2, 3, 4, 5, 0, 0, 0, 9, 10, 11, 12, 13, 14, 15]
    r = rng.uniform(-1, 1, size=(256, 16))
    scale = np.array([5, 5, 5, 10, 10, 10, 1.2, 1.2, np.pi, 10, 10, 10, 1, 1, 1, 1])
    states[1:] = (r * scale).astype(np.float32)
    vb = ns["get_body_velocity"](states.T).T
    thrust, moment = ns["thrust_moment_model_world_states"](states)
    save("f1_residual", states=states, vb=vb.astype(np.float32), thrust=thrust, moment=moment)


def rand_state16(rng, n):
    scale = np.array([5, 5, 5, 10, 10, 10, 1.2, 1.2, np.pi, 10, 10, 10, 1, 1, 1, 1])
    return (rng.uniform(-1, 1, size=(n, 16)) * scale).astype(np.float32)


def gen_f2_f3_ffunc(ns_e2e, ns_indi, rng):
    n = 1024
    s = rand_state16(rng, n)
    u = rng.uniform(-1, 1, size=(n, 4)).astype(np.float32)
    d = (rng.uniform(-1, 1, size=(n, 6)) * 3 * np.array([0.03, 0.03, 0.01, 0.1, 0.1, 0.5])).astype(np.float32)
    ds = ns_e2e["f_func"](s.T, u.T, d.T).T
    assert ds.dtype == np.float32, ds.dtype
    save("f2_ffunc_e2e", state=s, control=u, disturbance=d, dstate=ds)
    s13 = s[:, :13].copy()
    s13[:, 12] = rng.uniform(-1, 1, size=n).astype(np.float32)
    ds13 = ns_indi["f_func"](s13.T, u.T).T
    assert ds13.dtype == np.float32, ds13.dtype
    save("f3_ffunc_indi", state=s13, control=u, dstate=ds13)


def gen_track_tables(ns_e2e):
    out = {}
    for name, fn in TRACKS.items():
        gp, gy, sp = fn()
        env = ns_e2e["Quadcopter3DGates"](num_envs=1, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
        out[name + "_gate_pos"] = gp
        out[name + "_gate_yaw"] = gy
        out[name + "_start_pos"] = sp
        out[name + "_gate_pos_f32"] = env.gate_pos
        out[name + "_gate_yaw_f32"] = env.gate_yaw
        out[name + "_gate_pos_rel"] = env.gate_pos_rel
        out[name + "_gate_yaw_rel"] = env.gate_yaw_rel
    save("tracks", **out)


def gen_f4_obs(ns_e2e, ns_indi, rng):
    """F4: gate-frame observation transform (R:365-450, I:218-265)."""
    n = 256
    out = {}
    for tname, fn in TRACKS.items():
        gp, gy, sp = fn()
        G = gp.shape[0]
        for ga in (0, 1, 2):
            for variant, ns, S in (("e2e", ns_e2e, 16), ("indi", ns_indi, 13)):
                env = ns["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=ga)
                ws = rand_state16(rng, n)[:, :S]
                # yaw beyond +-pi exercises the wrap logic (R:392-397)
                ws[:, 8] = rng.uniform(-3 * np.pi, 3 * np.pi, size=n).astype(np.float32)
                tg = rng.integers(0, G, size=n)
                env.world_states = ws.copy()
                env.target_gates = tg.copy()
                key = f"{tname}_{variant}_ga{ga}"
                out[key + "_world"] = ws
                out[key + "_target"] = tg.astype(np.int32)
                if variant == "e2e":
                    for rname, ranges in (("zero", None), ("train", TRAIN_DIST_RANGES)):
                        if ranges is not None:
                            env.disturbance_ranges = ranges  # float64, as assigned in R:780
                            dist = rng.uniform(ranges[:, 0], ranges[:, 1], size=(n, 6)).astype(np.float32)
                        else:
                            dist = (rng.uniform(-1, 1, size=(n, 6)) * 0.1).astype(np.float32)
                        env.disturbances = dist.copy()
                        env.update_states()
                        out[key + f"_dist_{rname}"] = dist
                        out[key + f"_obs_{rname}"] = env.states.copy()
                        assert env.states.dtype == np.float32
                else:
                    env.update_states()
                    out[key + "_obs"] = env.states.copy()
    save("f4_obs", **out)


class Recorder:
    def __init__(self, env, has_dist):
        self.env, self.has_dist = env, has_dist
        self.rows = {k: [] for k in ("world", "obs", "reward", "done", "target", "steps", "dist")}

    def snap_initial(self):
        e = self.env
        init = dict(world0=e.world_states.copy(), obs0=e.states.copy(),
                    target0=e.target_gates.astype(np.int32).copy(), steps0=e.step_counts.astype(np.int32).copy())
        if self.has_dist:
            init["dist0"] = e.disturbances.copy()
        return init

    def step(self, actions):
        e = self.env
        obs, rew, done, infos = e.step(actions.copy())
        self.rows["world"].append(e.world_states.copy())
        self.rows["obs"].append(obs.copy())
        self.rows["reward"].append(rew.copy())
        self.rows["done"].append(done.copy())
        self.rows["target"].append(e.target_gates.astype(np.int32).copy())
        self.rows["steps"].append(e.step_counts.astype(np.int32).copy())
        if self.has_dist:
            self.rows["dist"].append(e.disturbances.copy())
        return obs, rew, done, infos

    def arrays(self):
        out = {k: np.stack(v) for k, v in self.rows.items() if v}
        out["done"] = out["done"].astype(np.uint8)
        return out


def hover_actions(rng, H, n, amp=0.15, base=0.124):
    t = np.arange(H)[:, None, None] * 0.01
    phase = rng.uniform(0, 2 * np.pi, size=(1, n, 4))
    freq = rng.uniform(0.5, 3.0, size=(1, n, 4))
    a = base + amp * np.sin(2 * np.pi * freq * t + phase)
    return a.astype(np.float32)


def gen_f5_config1(ns, rng):
    """F5 (BASELINE config 1): 1 env, E2E, NO residual, zero disturbance, fixed action sequence."""
    gp, gy, sp = zigzag_track()
    saved = ns["thrust_moment_model_world_states"]
    ns["thrust_moment_model_world_states"] = lambda s: (np.zeros((s.shape[0], 1), np.float32),
                                                         np.zeros((s.shape[0], 3), np.float32))
    try:
        out = {}
        for ga in (0, 1):
            for tag, H, kind in (("ctrl", 1200, "ctrl"), ("hover", 200, "hover"), ("random", 400, "random")):
                np.random.seed(0)
                env = ns["Quadcopter3DGates"](num_envs=1, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=ga)
                env.reset()
                if kind != "random":
                    # injected benign initial state (parity is defined on injected states)
                    env.world_states[0] = np.array([sp[0] + 0.1, sp[1] - 0.2, sp[2] + 0.05, 0.1, 0.3, -0.05,
                                                    0.03, -0.02, 0.1, 0.01, -0.02, 0.005,
                                                    0.124, 0.124, 0.124, 0.124], np.float32)
                    env.update_states()
                rec = Recorder(env, True)
                init = rec.snap_initial()
                acts = np.zeros((H, 1, 4), np.float32)
                if kind == "hover":
                    acts = hover_actions(rng, H, 1, amp=0.01)
                elif kind == "random":
                    acts = rng.uniform(-1, 1, size=(H, 1, 4)).astype(np.float32)
                for k in range(H):
                    if kind == "ctrl":
                        acts[k] = ctrl_e2e(env.world_states, waypoint(env))
                    rec.step(acts[k])
                arr = rec.arrays()
                key = f"ga{ga}_{tag}_"
                for k, v in {**init, **arr, "actions": acts}.items():
                    out[key + k] = v
        save("f5_traj_e2e_noresidual", **out)
    finally:
        ns["thrust_moment_model_world_states"] = saved


def gen_f6_residual_traj(ns, rng):
    """F6: E2E + residual MLPs + training disturbance ranges, small batch, free-running with auto-reset."""
    out = {}
    for tname, fn in TRACKS.items():
        gp, gy, sp = fn()
        n, H = 8, 600
        np.random.seed(1)
        env = ns["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
        env.disturbance_ranges = TRAIN_DIST_RANGES
        env.reset()
        rec = Recorder(env, True)
        init = rec.snap_initial()
        acts = rng.uniform(-1, 1, size=(H, n, 4)).astype(np.float32)
        for k in range(H):
            acts[k, : n // 2] = ctrl_e2e(env.world_states, waypoint(env))[: n // 2]
            rec.step(acts[k])
        for k, v in {**init, **rec.arrays(), "actions": acts}.items():
            out[f"{tname}_{k}"] = v
    save("f6_traj_e2e_residual", **out)


def gen_f7_branches(ns_e2e, ns_indi):
    """F7: one-step branch known-answers (gate pass, gate collision, ground, out of bounds, max steps, idle)."""
    out = {}
    for variant, ns, S in (("e2e", ns_e2e, 16), ("indi", ns_indi, 13)):
        gp, gy, sp = zigzag_track()
        cases = []

        def ws(x, y, z, vx=0.0, vy=0.0, vz=0.0, p=0.0, q=0.0, r=0.0):
            s = np.zeros(S, dtype=np.float32)
            s[0:6] = [x, y, z, vx, vy, vz]
            s[9:12] = [p, q, r]
            return s

        # gate 0 at (-3,0,-1.5), yaw +pi/2 -> normal (0,1): crossing y from <0 to >0
        cases.append(("pass_clean", ws(-3, -0.005, -1.5, vy=1.0), 0, 0))
        cases.append(("pass_offcentre", ws(-3.3, -0.004, -1.3, vy=1.0), 0, 0))
        cases.append(("collide_x", ws(-3.7, -0.005, -1.5, vy=1.0), 0, 0))
        cases.append(("collide_z", ws(-3.0, -0.005, -0.8, vy=1.0), 0, 0))
        cases.append(("wrong_way", ws(-3, 0.005, -1.5, vy=-1.0), 0, 0))
        cases.append(("ground", ws(-3, -1, -0.001, vz=1.0), 0, 0))
        cases.append(("oob_x", ws(9.995, -1, -1.5, vx=1.0), 0, 0))
        cases.append(("oob_y", ws(0, -9.995, -1.5, vy=-1.0), 0, 0))
        cases.append(("oob_rate", ws(-3, -1, -1.5, p=1500.0), 0, 0))
        cases.append(("max_steps", ws(-3, -1, -1.5), 0, 1199))
        cases.append(("idle", ws(-3, -1, -1.5), 0, 5))
        cases.append(("pass_gate3_wrap", ws(3, 0.005, -1.5, vy=-1.0), 3, 17))
        cases.append(("pass_last_gate", ws(-3, -0.005, -1.5, vy=1.0), 6, 17))
        n = len(cases)
        for ga in (1,):
            np.random.seed(7)
            env = ns["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=ga)
            env.reset()
            env.world_states = np.stack([c[1] for c in cases]).astype(np.float32)
            env.target_gates = np.array([c[2] for c in cases])
            env.step_counts = np.array([c[3] for c in cases])
            if variant == "e2e":
                env.disturbances = np.zeros((n, 6), np.float32)
                acts = np.full((n, 4), 0.124, np.float32)
            else:
                acts = np.zeros((n, 4), np.float32)
                acts[:, 3] = 0.22625  # T = 9.81 -> hover
            env.update_states()
            pre = dict(world0=env.world_states.copy(), target0=env.target_gates.astype(np.int32),
                       steps0=env.step_counts.astype(np.int32), obs0=env.states.copy())
            obs, rew, done, infos = env.step(acts)
            key = f"{variant}_"
            out.update({key + k: v for k, v in pre.items()})
            out[key + "actions"] = acts
            out[key + "reward"] = rew
            out[key + "done"] = done.astype(np.uint8)
            out[key + "target"] = env.target_gates.astype(np.int32)
            out[key + "steps"] = env.step_counts.astype(np.int32)
            out[key + "world"] = env.world_states.copy()  # post-reset for done rows
            out[key + "obs"] = obs.copy()
            out[key + "names"] = np.array([c[0] for c in cases])
            out[key + "info_has_terminal_obs"] = np.array(["terminal_observation" in i for i in infos])
            out[key + "info_truncated"] = np.array([i.get("TimeLimit.truncated", False) for i in infos])
    save("f7_branches", **out)


def gen_f11_edges(ns_e2e, ns_indi):
    """F11: one step of the reference from states at the edges the other fixtures miss -- a NaN / inf component
    (a NaN env stays alive until max_steps), body rates within +-1.5 of the 1000 rad/s guard, theta within 1e-3 of
    +-pi/2 (tan / 1/cos of the Euler kinematics blow up), |psi| ~ 1e4 (argument reduction of sin / cos, the yaw wrap of the
    observation).  Arrays only: pre-step state, action, and what the reference's step() returned."""
    out = {}
    hp = np.float32(np.pi / 2)
    for variant, ns, S in (("e2e", ns_e2e, 16), ("indi", ns_indi, 13)):
        gp, gy, sp = zigzag_track()
        cases = []

        def ws(**kw):
            s = np.zeros(S, dtype=np.float32)
            s[0:3] = [-3.0, -1.0, -1.5]
            s[3:6] = [0.3, 0.5, -0.1]
            s[6:9] = [0.05, -0.04, 0.3]
            s[9:12] = [0.2, -0.1, 0.15]
            if S == 16:
                s[12:16] = [0.1, 0.12, 0.08, 0.11]
            else:
                s[12] = 0.2
            for k, v in kw.items():
                s[int(k[1:])] = v
            return s

        nan, inf = np.float32(np.nan), np.float32(np.inf)
        for idx, nm in ((0, "x"), (2, "z"), (5, "vz"), (6, "phi"), (8, "psi"), (9, "p"), (12, "w1_or_T")):
            cases.append(("nan_" + nm, ws(**{"s%d" % idx: nan}), 0, 3))
        cases.append(("inf_vx", ws(s3=inf), 0, 3))
        cases.append(("nan_x_at_max_steps", ws(s0=nan), 0, 1199))
        cases.append(("nan_all", np.full(S, nan, np.float32), 0, 3))
        # body rates whose NEW value lands next to the 1000 rad/s guard (R:549-550 tests the post-step rates): the pre-step value
        # where the reference's own `done` flips is bracketed on a grid, then on consecutive float32 values
        def edge_of(idx, sign):
            def dones(vals):
                np.random.seed(5)
                e = ns["Quadcopter3DGates"](num_envs=len(vals), gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
                e.reset()
                e.world_states = np.stack([ws(**{"s%d" % idx: np.float32(v)}) for v in vals]).astype(np.float32)
                e.target_gates = np.zeros(len(vals), int)
                e.step_counts = np.full(len(vals), 3)
                if variant == "e2e":
                    e.disturbances = np.zeros((len(vals), 6), np.float32)
                e.update_states()
                with np.errstate(all="ignore"):
                    return e.step(np.tile(act_row, (len(vals), 1)))[2]
            grid = (sign * np.linspace(900.0, 2500.0, 6401)).astype(np.float32)
            d = dones(grid)
            k = int(np.argmax(d))
            assert d[k] and not d[k - 1], (variant, idx, sign)
            lo, hi = grid[k - 1], grid[k]
            while np.nextafter(lo, hi) != hi:      # bisect on float32 values
                mid = np.float32((np.float64(lo) + np.float64(hi)) / 2)
                if mid == lo or mid == hi:
                    break
                if dones(np.array([mid, mid], np.float32))[0]:
                    hi = mid
                else:
                    lo = mid
            return lo, hi

        act_row = (np.array([[0.124, 0.2, 0.05, 0.15]], np.float32) if variant == "e2e"
                   else np.array([[0.1, -0.2, 0.05, 0.22625]], np.float32))
        for idx, nm in ((9, "p"), (10, "q"), (11, "r")):
            for sign in (1.0, -1.0):
                lo, hi = edge_of(idx, sign)
                for tag, v in (("last_alive", lo), ("first_oob", hi), ("alive-1", np.float32(lo - sign * 1.0)), ("oob+1", np.float32(hi + sign * 1.0))):
                    cases.append(("rate_%s%s_%s" % ("+" if sign > 0 else "-", nm, tag), ws(**{"s%d" % idx: v}), 0, 3))
        for v, nm in ((hp - np.float32(1e-3), "hp-1e-3"), (hp - np.float32(1e-4), "hp-1e-4"), (-hp + np.float32(1e-3), "-hp+1e-3"),
                      (hp + np.float32(1e-3), "hp+1e-3")):
            cases.append(("theta_" + nm, ws(s7=v), 0, 3))
        for v in (1.0e4, -1.0e4, 12345.678, -9876.543, 31415.926):
            cases.append(("psi_%g" % v, ws(s8=np.float32(v)), 0, 3))
        n = len(cases)
        np.random.seed(11)
        env = ns["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
        env.reset()
        env.world_states = np.stack([c[1] for c in cases]).astype(np.float32)
        env.target_gates = np.array([c[2] for c in cases])
        env.step_counts = np.array([c[3] for c in cases])
        if variant == "e2e":
            env.disturbances = np.zeros((n, 6), np.float32)
        acts = np.tile(act_row, (n, 1))
        env.update_states()
        key = variant + "_"
        out.update({key + "world0": env.world_states.copy(), key + "target0": env.target_gates.astype(np.int32),
                    key + "steps0": env.step_counts.astype(np.int32), key + "obs0": env.states.copy(), key + "actions": acts})
        with np.errstate(all="ignore"):
            obs, rew, done, infos = env.step(acts)
        out[key + "reward"] = rew
        out[key + "done"] = done.astype(np.uint8)
        out[key + "target"] = env.target_gates.astype(np.int32)
        out[key + "steps"] = env.step_counts.astype(np.int32)
        out[key + "world"] = env.world_states.copy()   # post-reset for done rows
        out[key + "obs"] = obs.copy()
        out[key + "names"] = np.array([c[0] for c in cases])
    save("f11_edges", **out)


def gen_f8_indi_traj(ns, rng):
    """F8: INDI variant trajectories (config 3 shape, small N)."""
    out = {}
    for tname, fn in TRACKS.items():
        gp, gy, sp = fn()
        n, H = 8, 600
        np.random.seed(2)
        env = ns["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
        env.reset()
        rec = Recorder(env, False)
        init = rec.snap_initial()
        t = np.arange(H)[:, None] * 0.01
        acts = np.zeros((H, n, 4), np.float32)
        acts[:, :, 3] = 0.22625 + 0.1 * np.sin(2 * np.pi * 1.0 * t + np.arange(n)[None, :])
        acts[:, :, 0:3] = 0.2 * np.sin(2 * np.pi * 0.7 * t[:, :, None] + rng.uniform(0, 6, size=(1, n, 3)))
        acts[:, n // 2:, :] = rng.uniform(-1, 1, size=(H, n - n // 2, 4))
        acts = acts.astype(np.float32)
        for k in range(H):
            acts[k, : n // 2] = ctrl_indi(env.world_states, waypoint(env))[: n // 2]
            rec.step(acts[k])
        for k, v in {**init, **rec.arrays(), "actions": acts}.items():
            out[f"{tname}_{k}"] = v
    # single-env long hover-ish run for the 1e-5 free-run criterion
    gp, gy, sp = square_track()
    np.random.seed(3)
    env = ns["Quadcopter3DGates"](num_envs=1, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
    env.reset()
    rec = Recorder(env, False)
    init = rec.snap_initial()
    H = 1200
    acts = np.zeros((H, 1, 4), np.float32)
    for k in range(H):
        acts[k] = ctrl_indi(env.world_states, waypoint(env))
        rec.step(acts[k])
    for k, v in {**init, **rec.arrays(), "actions": acts}.items():
        out[f"single_{k}"] = v
    save("f8_traj_indi", **out)


def gen_f9_modes(ns_e2e, ns_indi, rng):
    """F9: pause_if_collision / pause mode traces (R:570-578)."""
    out = {}
    for variant, ns in (("e2e", ns_e2e), ("indi", ns_indi)):
        gp, gy, sp = zigzag_track()
        n, H = 4, 40
        np.random.seed(4)
        env = ns["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1,
                                      pause_if_collision=True)
        env.reset()
        # env 0: about to hit the ground; env 1: will collide with the gate frame; env 2/3: fly on
        env.world_states[0, 0:6] = [-3, -1, -0.02, 0, 0, 1.0]
        env.world_states[1, 0:6] = [-3.7, -0.03, -1.5, 0, 1.0, 0]
        env.update_states()
        rec = Recorder(env, variant == "e2e")
        init = rec.snap_initial()
        if variant == "e2e":
            acts = hover_actions(rng, H, n)
        else:
            acts = np.zeros((H, n, 4), np.float32)
            acts[:, :, 3] = 0.22625
        for k in range(H):
            if k == 25:
                env.pause = True
            rec.step(acts[k])
        for k, v in {**init, **rec.arrays(), "actions": acts}.items():
            out[f"{variant}_{k}"] = v
        out[f"{variant}_pause_from_step"] = np.array(25)
    save("f9_modes", **out)


def gen_reset_stats(ns_e2e, ns_indi):
    """Reset distribution bounds observed on the reference (R:452-493) -- used for distributional checks."""
    gp, gy, sp = zigzag_track()
    n = 20000
    np.random.seed(5)
    env = ns_e2e["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
    env.disturbance_ranges = TRAIN_DIST_RANGES
    env.reset()
    out = dict(e2e_world_min=env.world_states.min(0), e2e_world_max=env.world_states.max(0),
               e2e_world_mean=env.world_states.mean(0), e2e_world_std=env.world_states.std(0),
               e2e_dist_min=env.disturbances.min(0), e2e_dist_max=env.disturbances.max(0),
               e2e_dist_mean=env.disturbances.mean(0), e2e_dist_std=env.disturbances.std(0))
    env = ns_indi["Quadcopter3DGates"](num_envs=n, gates_pos=gp, gate_yaw=gy, start_pos=sp, gates_ahead=1)
    env.reset()
    out.update(indi_world_min=env.world_states.min(0), indi_world_max=env.world_states.max(0),
               indi_world_mean=env.world_states.mean(0), indi_world_std=env.world_states.std(0))
    save("reset_stats", **out)


def gen_f10_policy(rng):
    """F10: the reference's deployed policy network (c_code/neural_network.c, compiled from where it lies by
    oracle/Makefile): its baked weights (data) + 512 observation rows -> nn_forward outputs."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from oracle import oracle as O

    L = O.ref_policy_lib()
    assert L is not None, "oracle/_ref/libref_policy.so not built"
    f32p = C.POINTER(C.c_float)

    def arr(name, n):
        a = (C.c_float * n).in_dll(L, name)
        return np.ctypeslib.as_array(a).copy()

    w = dict(w1=arr("weights_fc1", 120 * 24).reshape(120, 24), b1=arr("biases_fc1", 120),
             w2=arr("weights_fc2", 120 * 120).reshape(120, 120), b2=arr("biases_fc2", 120),
             w3=arr("weights_fc3", 120 * 120).reshape(120, 120), b3=arr("biases_fc3", 120),
             w4=arr("weights_fc4", 4 * 120).reshape(4, 120), b4=arr("biases_fc4", 4))
    n = 512
    scale = np.array([3, 3, 1, 5, 5, 3, 0.6, 0.6, 3.1, 3, 3, 2, 1, 1, 1, 1, 3, 3, 0.5, 2, 1, 1, 1, 1])
    x = (rng.uniform(-1, 1, size=(n, 24)) * scale).astype(np.float32)
    y = np.zeros((n, 4), np.float32)
    for i in range(n):
        L.nn_forward(x[i].ctypes.data_as(f32p), y[i].ctypes.data_as(f32p))
    save("f10_policy", obs=x, mean=y, **w)


def main():
    assert ref_import.reference_available(), "reference not mounted"
    if "--only-f11" in sys.argv:    # round 4 added F11; the other fixtures are unchanged (regenerating them is byte-identical)
        os.makedirs(OUT, exist_ok=True)
        gen_f11_edges(ref_import.load_e2e(), ref_import.load_indi())
        return
    os.makedirs(OUT, exist_ok=True)
    print("importing reference notebooks ...")
    ns_e2e = ref_import.load_e2e()
    ns_indi = ref_import.load_indi()
    rng = np.random.default_rng(20240928)
    export_weights(ns_e2e)
    gen_f1_residual(ns_e2e, rng)
    gen_f2_f3_ffunc(ns_e2e, ns_indi, rng)
    gen_track_tables(ns_e2e)
    gen_f4_obs(ns_e2e, ns_indi, rng)
    gen_f5_config1(ns_e2e, rng)
    gen_f6_residual_traj(ns_e2e, rng)
    gen_f7_branches(ns_e2e, ns_indi)
    gen_f8_indi_traj(ns_indi, rng)
    gen_f9_modes(ns_e2e, ns_indi, rng)
    gen_reset_stats(ns_e2e, ns_indi)
    gen_f10_policy(rng)
    gen_f11_edges(ns_e2e, ns_indi)
    print("done")


if __name__ == "__main__":
    main()
