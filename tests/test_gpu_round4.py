"""GPU: round-4 tests -- bench.py's own rank launcher on a box with too few GPUs, the edge-case fixture F11 (NaN states, rates at the
1000 rad/s guard, theta at +-pi/2, huge psi) and whatever else round 4 adds."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_on_a_smaller_box_fails_loudly():
    """`python bench.py --gpus N` with fewer than N visible GPUs exits non-zero with a clear message -- it never
    prints a line for a smaller job (it used to run ONE rank and print n_gpus 1)."""
    want = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "QR_BENCH_TEST_FACTORY")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-parity", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert f"--gpus {want} needs {want} visible GPUs" in r.stderr, r.stderr[-1000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]       # no JSON line at all


@pytest.mark.parametrize("variant,vname", [(0, "e2e"), (1, "indi")])
def test_edge_states_f11(variant, vname, residual_blob):
    """the reference's own step() from states at the edges -- NaN / inf components (alive until max_steps, NaN
    pattern identical), rates bracketing the 1000 rad/s guard on consecutive float32 values, theta within 1e-3 of +-pi/2, |psi| ~ 1e4
    -- replayed on the HIP path (tests/parity.py check_edges; the CPU suite holds the oracle to the same fixture)."""
    import parity as P
    from product_adapter import ProductAdapter

    print(P.check_edges(ProductAdapter, variant, vname, residual_blob))
