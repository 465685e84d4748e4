"""GPU: the one-launch form of BTRAN + pass over the nucleus inverse + v tail + touched-column list (k_small_basis) that the
lazy primal iteration uses while the inverse still has its first capacity (256 slots) — solver.rs:1106-1132 (rho, v) and
1274-1284 (the eta transformation) for a small nucleus.

The kernel forms every sum in the order of the three launches it replaces (k_btran, k_fused_w<8>, k_post_fused), so
  (a) a solve with it and a solve without it (MLP_SMALL_BASIS=0) must agree BIT FOR BIT — trace, objective, values —
      and both take the oracle's pivots;
  (b) a solve whose nucleus outgrows the capacity moves to the three launches mid-solve;
  (c) its only in-kernel wait (block 0 for the t_K tickets) giving up — MLP_RATIO_SPIN_LIMIT=0 on a model small enough
      that the ratio tests run in one block and cannot stall themselves — ends the batch with ITER_STALL before anything
      was applied to W; the engine re-runs the iteration on the three launches and nothing shows in the pivots.
state("small_basis_launches") counts the iterations the kernel ran, so "the path was taken" is checked, not assumed."""
import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, X_ATOL, obj_close

pytestmark = pytest.mark.gpu

CASES = [("sparse", (700, 600, 12, 6)), ("dense", (150, 100, 3)), ("sparse", (1000, 1000, 100, 2)), ("twophase", (600, 600, 12, 44))]


@pytest.mark.parametrize("fam,args", CASES, ids=str)
@pytest.mark.parametrize("grid_ratio", [0, 1], ids=["t_K inside", "t_K riding in the grid-form ratio launch"])
def test_one_launch_form_is_bit_identical_to_the_three_launches(monkeypatch, fam, args, grid_ratio):
    monkeypatch.setenv("MLP_HYPER", "0")
    if grid_ratio:
        # models above 16 384 rows (config 4) run the grid form of the ratio test, and t_K then rides behind its blocks with y_S formed
        # on the fly (k_small_basis carries no t_K blocks and waits for none); forced here at small sizes
        monkeypatch.setenv("MLP_RATIO_ONE", "0")
    monkeypatch.setenv("MLP_SMALL_BASIS_K", "256")  # (every size the sparse tableau row allows, not just the sizes at which the form pays)
    lp = GEN[fam](*args)
    runs = []
    for on in ("1", "0"):
        monkeypatch.setenv("MLP_SMALL_BASIS", on)
        s = lpgen.build_problem(M.Problem, lp).solve(trace=True)
        runs.append((s.trace(), s.objective(), s.values().tobytes(), int(s.state("small_basis_launches")[0]), s.stats()))
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    assert runs[0][3] > 0 and runs[1][3] == 0, (runs[0][3], runs[1][3])
    assert runs[0][0] == runs[1][0]
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    assert [t[:5] for t in runs[0][0]] == [t[:5] for t in so.trace()]
    assert obj_close(runs[0][1], so.objective())
    assert np.abs(np.frombuffer(runs[0][2]) - so.values()).max() <= X_ATOL
    print(f"{fam}{args}: {len(runs[0][0])} pivots, {runs[0][3]} through k_small_basis, nucleus {runs[0][4]['nucleus_size']}")


def test_nucleus_outgrowing_the_first_capacity_moves_to_the_three_launches():
    lp = lpgen.gen_sparse_lp(2500, 2000, 10, 4)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    st = sg.stats()
    n_sb = int(sg.state("small_basis_launches")[0])
    assert st["nucleus_capacity"] > 256 and 0 < n_sb < st["iterations"], (st["nucleus_capacity"], n_sb, st["iterations"])
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert sg.reinvert() < 1e-8


def test_a_stalled_wait_falls_back_before_anything_was_applied(monkeypatch):
    monkeypatch.setenv("MLP_RATIO_SPIN_LIMIT", "0")
    monkeypatch.setenv("MLP_HYPER", "0")
    lp = lpgen.gen_sparse_lp(700, 600, 12, 6)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    st = sg.stats()
    assert st["ratio_stalls"] == 1, st["ratio_stalls"]
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL
    assert sg.reinvert() < 1e-8
