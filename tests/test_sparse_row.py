"""GPU: the sparse tableau row of the multi-kernel iteration (k_row_touch / k_row_pull): while the nucleus is small,
alpha_r = rho^T N and the steepest-edge helper N^T v are formed on the columns that meet a row of supp(rho) only
(solver.rs:685-692 iterates over exactly those rows) instead of a pass over all of A.  Forced here up to a nucleus of
4 000 (MLP_STR_K) so that whole solves run on it — primal with steepest edge, dual, two-phase (dual with the helper pass),
bound flips, warm starts — and must take the oracle's pivots; the default threshold is exercised by every other test."""
import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, X_ATOL, check_feasible, obj_close

pytestmark = pytest.mark.gpu

CASES = [("sparse", (700, 600, 12, 6)), ("sparse", (2500, 2000, 10, 4)), ("cover", (700, 900, 12, 5)), ("twophase", (600, 600, 12, 44)),
         ("dense", (150, 100, 3))]


@pytest.mark.parametrize("fam,args", CASES, ids=str)
@pytest.mark.parametrize("banded", [0, 1, 2], ids=["csc sweep", "banded sweep", "pushed F products + sparse ratio test"])
def test_sparse_tableau_row_takes_the_oracles_pivots(monkeypatch, fam, args, banded):
    monkeypatch.setenv("MLP_STR_K", "4000")
    monkeypatch.setenv("MLP_HYPER", "0")       # (the multi-kernel iteration is the one under test)
    if banded == 1:
        monkeypatch.setenv("MLP_BANDED", "1")  # the form config 4 uses beyond the threshold
    if banded == 2:
        # the F products pushed with atomics as on config 4 (small models pull them by default): the FTRAN then lists
        # supp(alpha_q) and the primal Harris test runs over that list in one block (k_ratio_primal_fused, sparse form)
        monkeypatch.setenv("MLP_DETERMINISTIC", "0")
    lp = GEN[fam](*args)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL
    assert sg.reinvert() < 1e-8


def test_switch_from_sparse_to_dense_row_mid_solve(monkeypatch):
    """Threshold 40: the solve starts on the sparse form and moves to the sweep when the nucleus outgrows it."""
    monkeypatch.setenv("MLP_STR_K", "140")
    monkeypatch.setenv("MLP_HYPER", "0")
    lp = lpgen.gen_sparse_lp(700, 600, 12, 6)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert sg.stats()["nucleus_size"] > 140
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]


def test_warm_start_and_mutators_on_the_sparse_row(monkeypatch):
    monkeypatch.setenv("MLP_STR_K", "4000")
    monkeypatch.setenv("MLP_HYPER", "0")
    lp = lpgen.gen_sparse_lp(300, 260, 8, 21)
    so = lpgen.build_problem(O.Problem, lp).solve()
    sg = lpgen.build_problem(M.Problem, lp).solve()
    rng = np.random.default_rng(11)
    x = so.values()
    for step in range(8):
        vars_ = rng.choice(lp["n"], size=4, replace=False)
        coef = rng.integers(1, 4, size=4).astype(float)
        rhs = 0.9 * float(np.dot(coef, x[vars_])) + 0.01
        expr = list(zip(vars_.tolist(), coef.tolist()))
        so, sg = so.add_constraint(expr, O.LE, rhs), sg.add_constraint(expr, M.LE, rhs)
        assert obj_close(sg.objective(), so.objective()), step
        x = so.values()
        assert np.abs(sg.values() - x).max() <= X_ATOL
    v = int(np.argmax(x))
    so, sg = so.fix_var(v, 0.5 * x[v]), sg.fix_var(v, 0.5 * x[v])
    assert obj_close(sg.objective(), so.objective())
    (so, wo), (sg, wg) = so.unfix_var(v), sg.unfix_var(v)
    assert wo and wg and obj_close(sg.objective(), so.objective())
    frac = [i for i in range(lp["n"]) if abs(x[i] - round(x[i])) > 1e-3][:1]
    for i in frac:   # a Gomory cut reads the dense tableau row in between (calc_row_coeffs)
        so, sg = so.add_gomory_cut(i), sg.add_gomory_cut(i)
        assert obj_close(sg.objective(), so.objective())
