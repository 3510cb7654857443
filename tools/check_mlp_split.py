#!/usr/bin/env python3
"""How accurate is layer 1 of the residual MLPs as SPLIT low-precision products with f32 accumulation?  (CPU, NumPy; no reference
access needed: the weights are the committed blob, the inputs the committed fixture F1.)  Emulates

    x = X0 + X1 (+ X2),  w = W0 + W1 (+ W2)  pieces rounded to f16 / bf16 (round to nearest even), products summed exactly, rounded once

against the float64 value of the layer, next to the float32 fmaf chain of rounds 1-3.  The choice made in round 4 -- f16, two pieces
each, three products -- is the row that matches the chain's own error.      python tools/check_mlp_split.py
"""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(ROOT, "tests", "golden", "f1_residual.npz"))
blob = np.fromfile(os.path.join(ROOT, "optimal_quad_control_rl_amd", "data", "residual_mlp_f32.bin"), np.float32)
o = 0
W1t = blob[o:o + 224].reshape(32, 7); o += 224; b1t = blob[o:o + 32]; o += 32; W2t = blob[o:o + 32].reshape(1, 32); o += 32; b2t = blob[o:o + 1]; o += 1
W1m = blob[o:o + 320].reshape(32, 10); o += 320; b1m = blob[o:o + 32]; o += 32; W2m = blob[o:o + 96].reshape(3, 32); o += 96; b2m = blob[o:o + 3]


def bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32)).view(np.float32)


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def split(x, n, rnd):
    out, r = [], np.asarray(x, np.float32)
    for _ in range(n):
        p = rnd(r); out.append(p); r = (r - p).astype(np.float32)
    return out


def emulated(terms, nx, nw, rnd):
    def layer(W, b, X):
        Xp, Wp, bp = split(X, nx, rnd), split(W, nw, rnd), split(b, nw, rnd)
        acc = sum(Xp[p].astype(np.float64) @ Wp[i].astype(np.float64).T for p, i in terms)
        return (acc + sum(x.astype(np.float64) for x in bp)[None, :]).astype(np.float32)
    return layer


def chain(W, b, X):   # k-ordered float32 multiply-add chain (what v_mfma_f32_32x32x2_f32 computes)
    acc = np.zeros((X.shape[0], W.shape[0]), np.float32)
    for k in range(X.shape[1]):
        acc = (acc.astype(np.float64) + X[:, k:k + 1].astype(np.float64) * W[None, :, k].astype(np.float64)).astype(np.float32)
    return (acc + b).astype(np.float32)


S, vb = d["states"], d["vb"]
Xt = np.concatenate([S[:, 12:16], vb], 1).astype(np.float32)
Xm = np.concatenate([Xt, S[:, 9:12]], 1).astype(np.float32)


def run(layer):
    ht, hm = np.maximum(layer(W1t, b1t, Xt), 0), np.maximum(layer(W1m, b1m, Xm), 0)
    return ((ht.astype(np.float64) @ W2t.T.astype(np.float64) + b2t).astype(np.float32),
            (hm.astype(np.float64) @ W2m.T.astype(np.float64) + b2m).astype(np.float32))


th0, mo0 = run(lambda W, b, X: (X.astype(np.float64) @ W.T.astype(np.float64) + b).astype(np.float32))
print("layer-1 arithmetic                thrust max rel err   moment max abs err   (257 fixture rows, vs float64)")
for name, layer in [("f32 fmaf chain (rounds 1-3)", chain),
                    ("f16  x2 w2, 3 products (round 4)", emulated([(0, 0), (0, 1), (1, 0)], 2, 2, f16)),
                    ("f16  x2 w3, 5 products", emulated([(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)], 2, 3, f16)),
                    ("bf16 x3 w3, 6 products", emulated([(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)], 3, 3, bf16)),
                    ("bf16 x2 w2, 3 products", emulated([(0, 0), (0, 1), (1, 0)], 2, 2, bf16))]:
    th, mo = run(layer)
    print(f"{name:34s} {(np.abs(th - th0) / np.maximum(1, np.abs(th0))).max():14.2e} {np.abs(mo - mo0).max():20.2e}")
