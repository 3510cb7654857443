"""CPU suite: the predecessor-env oracle (oracle/quad3d_oracle.c) against fixtures generated from the real
"3D quad.ipynb" (tools/gen_golden_q3.py).  SURVEY.md section 8(f) #4."""
import numpy as np
import pytest

import parity_quad3d as pq
from parity import load, rel_err
from oracle import quad3d as q3


class OracleImpl:
    def __init__(self, kind, n, track):
        if kind == "hover":
            self.env = q3.Quad3DOracle(q3.HOVER, n)
        else:
            self.env = q3.Quad3DOracle(q3.GATES, n, *track)

    def set_state(self, states, target, steps):
        self.env.states[:] = states
        if target is not None:
            self.env.target[:] = target
        self.env.steps[:] = steps

    def get_state(self):
        return self.env.states.copy(), self.env.target.copy(), self.env.steps.copy()

    def step(self, actions):
        return self.env.step(actions)

    def reset(self):
        return self.env.reset()


def make(kind, n, track):
    return OracleImpl(kind, n, track)


def test_ffunc_matches_reference():
    d = load("q3_ffunc")
    got64 = q3.f_func(d["state64"], d["control"])
    assert rel_err(got64, d["dstate64"]).max() <= 1e-13
    got32 = q3.f_func(d["state32"], d["control"])
    # float32: the lambdified expression cancels +-100-sized terms; NumPy's SIMD sin/cos differ from libm by an ulp
    assert rel_err(got32, d["dstate32"]).max() <= 5e-6


def test_hover_step_branches():
    pq.check_hover_step(make)


def test_hover_free_run():
    pq.check_hover_free_run(make)


def test_gates_step_branches():
    pq.check_gates_step(make)


def test_gates_free_run():
    pq.check_gates_free_run(make)


def test_reset_distributions_match_reference():
    pq.check_reset_distribution(make)


def test_reset_is_counter_based():
    """Same (seed, env id, episode) -> same draw, independent of batch composition (sharding contract)."""
    gp, gy, sp = pq.gates_track()
    a = q3.Quad3DOracle(q3.GATES, 64, gp, gy, sp, seed=5)
    b = q3.Quad3DOracle(q3.GATES, 32, gp, gy, sp, env_id_base=32, seed=5)
    sa, sb = a.reset(), b.reset()
    assert np.array_equal(sa[32:], sb) and np.array_equal(a.target[32:], b.target)
    s2 = a.reset(np.arange(64) % 2 == 0)
    assert np.array_equal(s2[1::2], sa[1::2]) and not np.array_equal(s2[0::2], sa[0::2])
