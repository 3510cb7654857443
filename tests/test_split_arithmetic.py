"""The operand-splitting arithmetic the reference-precision kernels rest on, pinned on the CPU with torch's round-to-nearest-even conversions
(the same rounding as v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950):

  * csrc/quadrace_ppo_f32.hip `split8`: x = X0 + X1 + X2 EXACTLY with three bf16 pieces (8 + 8 + 8 significant bits: the remainder after a
    round-to-nearest 8-bit piece is signed and at most half an ulp, so 24 bits fit), for every float32 whose last bit is representable in bf16;
  * why not two f16 pieces there (the form the forward kernels use, csrc/quadrace_policy.hpp `split_pack`): f16's quantum 2^-24 swallows the low
    piece of small values -- fine for observations and activations of O(1) whose products are summed (7e-7 of nn_forward), wrong for deltas and
    small activations in a 40 000-row weight gradient;
  * the dropped products of the six-instruction form (X1 Y2, X2 Y1, X2 Y2) are <= 2^-23 of |x y|.
"""
import numpy as np
import torch


def _bf16_pieces(x):
    p0 = x.to(torch.bfloat16).to(torch.float32)
    r1 = x - p0
    p1 = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - p1
    p2 = r2.to(torch.bfloat16).to(torch.float32)
    return p0, p1, p2


def _values(seed=0):
    g = torch.Generator().manual_seed(seed)
    mant = torch.rand(200_000, generator=g, dtype=torch.float64) + 1.0              # [1, 2)
    expo = torch.randint(-60, 60, (200_000,), generator=g).to(torch.float64)
    sign = torch.where(torch.rand(200_000, generator=g) < 0.5, -1.0, 1.0).to(torch.float64)
    x = (sign * mant * torch.pow(torch.tensor(2.0, dtype=torch.float64), expo)).to(torch.float32)
    edge = torch.tensor([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 2.0 - 2.0 ** -23, 0.1, 1e-3, 3.0e-7, 65504.0, 1e30, -1e-30,
                         float(np.nextafter(np.float32(0.5), np.float32(1.0))), 255.0 / 256.0 + 2.0 ** -24], dtype=torch.float32)
    return torch.cat([x, edge])


def test_three_bf16_pieces_reproduce_a_float32_exactly():
    x = _values()
    p0, p1, p2 = _bf16_pieces(x)
    assert torch.equal((p0.double() + p1.double()) + p2.double(), x.double())        # exact, not merely close
    assert torch.equal((p0 + p1) + p2, x)                                            # and the f32 sum of the pieces is x again
    # each piece really is a bf16 (8 significant bits): converting it again changes nothing
    for p in (p0, p1, p2):
        assert torch.equal(p.to(torch.bfloat16).to(torch.float32), p)
    # the remainders shrink by 2^-8 per piece (what makes the dropped products negligible)
    nz = x != 0
    assert float((p1[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -8
    assert float((p2[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -16


def test_dropped_products_of_the_six_instruction_form_are_below_f32_resolution():
    x, y = _values(1)[:100_000], _values(2)[:100_000]
    a, b = _bf16_pieces(x), _bf16_pieces(y)
    kept = sum((a[i].double() * b[j].double()) for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)))
    exact = x.double() * y.double()
    nz = exact != 0
    rel = ((kept - exact).abs() / exact.abs())[nz]
    assert float(rel.max()) <= 2.0 ** -22.9, float(rel.max())                         # a1 b2 + a2 b1 + a2 b2: 3 x 2^-24 at worst... below one f32 ulp


def test_two_f16_pieces_lose_small_values_which_is_why_the_gradient_path_uses_bf16():
    """f16 pieces: the low piece cannot hold anything below f16's smallest subnormal 2^-24 -- a delta of 3e-4 keeps 2^-24 / 3e-4 = 2e-4 relative
    error, the 1e-3 class error measured on a 40 000-row weight gradient with two f16 pieces; at O(1) magnitudes the same split is good to 2^-22."""
    def f16_pieces(x):
        p0 = x.to(torch.float16).to(torch.float32)
        p1 = (x - p0).to(torch.float16).to(torch.float32)
        return p0, p1
    g = torch.Generator().manual_seed(3)
    big = (torch.rand(100_000, generator=g) + 0.5)                                   # [0.5, 1.5): observations, activations
    p0, p1 = f16_pieces(big)
    assert float(((p0 + p1 - big).abs() / big).max()) <= 2.0 ** -21
    small = (torch.rand(100_000, generator=g) + 0.5) * 3e-4                           # deltas of a large minibatch
    p0, p1 = f16_pieces(small)
    err16 = float(((p0 + p1 - small).abs() / small).max())
    assert err16 > 2e-5                                                               # four orders worse than float32 ...
    q = _bf16_pieces(small)
    assert torch.equal((q[0] + q[1]) + q[2], small)                                   # ... where three bf16 pieces are still exact
