"""Race tracks and domain-randomisation constants used by the reference (data, not code):
   zigzag_track      7-gate slalom of '3D quad race.ipynb'            (R:640-661)
   square_track      4-gate square, listed twice = 8 entries          (I:438-460, FP:191-203, c_code/nn_controller.c:14-38)
   TRAIN_DISTURBANCE_RANGES                                           (R:772-779)
"""
import numpy as np


def zigzag_track(l=1.0):
    gate_pos = np.array([[-3 * l, 0, -1.5], [-1 * l, 0, -1.5], [1 * l, 0, -1.5], [3 * l, 0, -1.5],
                         [1 * l, 0, -1.5], [-1 * l, 0, -1.5], [-3 * l, 0, -1.5]], dtype=np.float64)
    gate_yaw = np.array([np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2])
    start_pos = gate_pos[0] + np.array([0, -1.0, 0])
    return gate_pos, gate_yaw, start_pos


def square_track():
    gate_pos = np.array([[2, -1.5, -1.5], [2, 1.5, -1.5], [-2, 1.5, -1.5], [-2, -1.5, -1.5]] * 2, dtype=np.float64)
    gate_yaw = np.array([np.pi / 4, 3 * np.pi / 4, 5 * np.pi / 4, 7 * np.pi / 4] * 2)
    start_pos = gate_pos[3].copy()
    return gate_pos, gate_yaw, start_pos


# (min, max) of M_ext_x, M_ext_y, M_ext_z, F_ext_x, F_ext_y, F_ext_z
TRAIN_DISTURBANCE_RANGES = np.array([[-0.03, 0.03], [-0.03, 0.03], [-0.01, 0.01], [0, 0], [0, 0], [-0.5, 0.5]])
