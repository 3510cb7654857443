#!/usr/bin/env python3
"""Accuracy check of the kernel's branch-free sincos (quadrace_device.hpp: qr_sincos) emulated in NumPy float32
(fma emulated through float64) against float64 sin/cos, next to NumPy's own float32 sin/cos (what the reference runs)."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def sincos32(x):
    x = x.astype(f32)
    k = np.rint(x * f32(0.6366197723675814)).astype(f32)
    r = fma(-k, f32(1.5707963705062866), x)
    r = fma(-k, f32(-4.371139000186241e-8), r)
    r = fma(-k, f32(-1.7151245100059e-15), r)
    z = r * r
    sp = fma(z, f32(2.7183114939898219e-6), f32(-0.00019839334836563469))
    sp = fma(z, sp, f32(0.0083333375930786133))
    sp = fma(z, sp, f32(-0.16666667163372040))
    s = fma(r * z, sp, r)
    cp = fma(z, f32(2.4390448796277409e-5), f32(-0.0013886763774609929))
    cp = fma(z, cp, f32(0.041666623323739063))
    cp = fma(z, cp, f32(-0.49999999725103100))
    c = fma(z, cp, f32(1.0))
    q = k.astype(np.int64)
    a = np.where(q & 1, c, s)
    b = np.where(q & 1, s, c)
    return np.where(q & 2, -a, a).astype(f32), np.where((q + 1) & 2, -b, b).astype(f32)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for R in (0.8, 3.2, 100, 2000, 20000):
        x = rng.uniform(-R, R, 2_000_000).astype(f32)
        s, c = sincos32(x)
        rs, rc = np.sin(x.astype(np.float64)), np.cos(x.astype(np.float64))
        print(f"|x| <= {R:7}: qr_sincos abs err sin {np.abs(s - rs).max():.2e} cos {np.abs(c - rc).max():.2e}   "
              f"numpy f32 sin {np.abs(np.sin(x) - rs).max():.2e} cos {np.abs(np.cos(x) - rc).max():.2e}")
        assert np.abs(s - rs).max() < 8e-8 and np.abs(c - rc).max() < 8e-8
