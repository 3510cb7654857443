"""Wraps the TEST-ONLY CPU oracle in the interface tests/parity.py drives."""
import numpy as np

from oracle import oracle as O


class OracleAdapter:
    def __init__(self, variant, n, track, gates_ahead=1, residual=None, dist_ranges=None, pause_if_collision=False,
                 seed=0, env_id_base=0):
        gp, gy, sp = track
        self.env = O.OracleEnv(variant, n, gp, gy, sp, gates_ahead, pause_if_collision, env_id_base)
        self.variant = variant
        if variant == O.E2E:
            self.env.set_residual(residual)
            if dist_ranges is not None:
                self.env.set_disturbance(dist_ranges, 1.0)
        self.env.seed(seed)

    def set_state(self, world, dist, target, steps):
        e = self.env
        e.world_states[:] = world
        if dist is not None and self.variant == O.E2E:
            e.disturbances[:] = dist
        e.target_gates[:] = target
        e.step_counts[:] = steps

    def get_state(self):
        e = self.env
        return e.world_states.copy(), e.disturbances.copy(), e.target_gates.copy(), e.step_counts.copy()

    def step(self, actions):
        return self.env.step(actions)

    def observe(self):
        return self.env.observe()

    def reset(self, mask=None):
        return self.env.reset(mask)

    def set_pause(self, p):
        self.env.set_pause(p)
