"""Stress of the memory-model-by-convention synchronisation (fence-free ticketed reductions, the
in-kernel publish / wait between the two Harris passes): several processes pivot concurrently on ONE GPU, so their
workgroups compete for the same CUs, and every process must still take the pivot sequence of a solo run.  The
documented fallback (MLP_RATIO_TWO_KERNELS=1: two launches instead of the in-kernel wait) is held to the same gate."""
import os
import subprocess
import sys

import pytest

from tests.common import ROOT

pytestmark = pytest.mark.gpu


def _run(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_concurrent.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "identical to solo: True" in r.stdout
    return r.stdout


def test_four_concurrent_primal_solves_take_the_solo_pivot_sequence():
    # 40 000 rows: 40-block grids in the reductions, the banded sweep (one workgroup per CU) and the fused ratio test
    _run([4, 40000, 40000, 30, 1200])


def test_three_concurrent_dual_solves_take_the_solo_pivot_sequence():
    _run([3, 20000, 24000, 16, 800, "cover"])


def test_two_kernel_ratio_fallback_under_the_same_load():
    _run([3, 40000, 40000, 30, 600], env=dict(os.environ, MLP_RATIO_TWO_KERNELS="1"))
