"""Row a13 (BasisSolver::reset, solver.rs:1286-1303): the device re-inversion of the basis checked against the ORACLE's
LU (lu.rs:118-304 restated, pinned by the reference's lu_simple / lu_rand KATs) — not against itself.

After `mlp_solution_reinvert` (and after `mlp_problem_solve_from_basis`, which rebuilds the inverse from A as well)
one iteration is stepped through the C ABI and its two solves are compared with the oracle's LU solves on the same
basis matrix:  alpha_q = B^-1 a_q  (LUFactors::solve, lu.rs:79-106)  and  rho_r = B^-T e_r  (the transposed factors).
Both hand-written back ends are covered: the unblocked Gauss-Jordan (nucleus < 384) and the blocked in-place Gauss-Jordan
whose rank-32 update is the fold kernel (csrc/inverse.inc; rounds 1-3 called rocSOLVER there), the latter also against the former
on the same nucleus (MLP_REINVERT_GJ=1)."""
import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN

pytestmark = pytest.mark.gpu
A = M.api


def _basis_csc(lp, basic_vars):
    """CSC of B = columns `basic_vars` of [A | I] (slack coefficient +1, solver.rs:250)."""
    m, n = lp["m"], lp["n"]
    ip, ix, dv = lp["indptr"], lp["indices"], lp["data"]
    cols = [[] for _ in range(n)]
    for i in range(m):
        for q in range(int(ip[i]), int(ip[i + 1])):
            cols[int(ix[q])].append((i, float(dv[q])))
    indptr, rows, vals = [0], [], []
    for v in basic_vars:
        v = int(v)
        ent = cols[v] if v < n else [(v - n, 1.0)]
        for r, a in ent:
            rows.append(r)
            vals.append(a)
        indptr.append(len(rows))
    return indptr, rows, vals, cols


def _check_one_iteration_against_lu(lp, s, tol=1e-9):
    bv = np.asarray(s.state("basic_vars")).astype(np.int64)
    m, n = lp["m"], lp["n"]
    indptr, rows, vals, cols = _basis_csc(lp, bv)
    lu = O.LU(m, indptr, rows, vals, 0.1)                     # the reference's stability coefficient (solver.rs:313)
    st, info = s.engine_open()
    assert st == A.ITER_PIVOT and info["phase"] == 0
    st, info = s.engine_stage(A.STAGE_FTRAN)
    alpha = np.asarray(s.state("col_coeffs"))
    q_var = int(info["entering_var"])
    rhs = np.zeros(m)
    for r, a in (cols[q_var] if q_var < n else [(q_var - n, 1.0)]):
        rhs[r] = a
    want = lu.solve_dense(rhs)                                  # B x = a_q
    scale = max(1.0, np.abs(want).max())
    assert np.abs(alpha - want).max() <= tol * scale, np.abs(alpha - want).max()
    st, info = s.engine_stage(A.STAGE_RATIO)
    if st != A.ITER_PIVOT:
        return False
    r = int(info["row"])
    st, info = s.engine_stage(A.STAGE_BTRAN)
    rho = np.asarray(s.state("inv_basis_row_coeffs"))
    e = np.zeros(m)
    e[r] = 1.0
    want = lu.solve_dense(e, transp=True)                       # B^T y = e_r
    scale = max(1.0, np.abs(want).max())
    assert np.abs(rho - want).max() <= tol * scale, np.abs(rho - want).max()
    for stage in (A.STAGE_BASIS, A.STAGE_ROW, A.STAGE_APPLY):
        st, info = s.engine_stage(stage)
    return True


@pytest.mark.parametrize("fam,args,pivots,min_k,max_k", [("sparse", (300, 260, 10, 51), 150, 20, 383),
                                                         ("sparse", (1500, 1500, 30, 52), 1500, 384, 10 ** 9)], ids=["gauss_jordan", "blocked"])
def test_reinverted_basis_solves_like_the_oracle_lu(fam, args, pivots, min_k, max_k):
    lp = GEN[fam](*args)
    s = lpgen.build_problem(M.Problem, lp).solve(budget=pivots)
    assert s.budget_exhausted
    k = s.stats()["nucleus_size"]
    assert min_k <= k <= max_k, k
    drift = s.reinvert()                                        # fresh inverse from A replaces the incremental one
    assert drift <= 1e-8
    assert s.stats()["reinversions"] == 1
    done = sum(_check_one_iteration_against_lu(lp, s) for _ in range(3))
    assert done >= 1
    # and the solve goes on to the oracle's optimum
    s.continue_solve(-1)
    so = lpgen.build_problem(O.Problem, lp).solve()
    assert abs(s.objective() - so.objective()) <= 1e-9 * max(1.0, abs(so.objective()))


def test_basis_loaded_from_a_checkpoint_solves_like_the_oracle_lu():
    lp = GEN["sparse"](1500, 1500, 30, 52)
    prob = lpgen.build_problem(M.Problem, lp)
    blob = prob.solve(budget=1200).save_basis(2)
    s = prob.solve_from_basis(blob, budget=0)                   # load = classify columns + invert the nucleus from A
    assert s.stats()["reinversions"] == 1 and s.stats()["nucleus_size"] >= 384
    assert _check_one_iteration_against_lu(lp, s) in (True, False)


def test_blocked_and_unblocked_inversion_agree_on_the_same_nucleus(monkeypatch):
    """k = 1 0xx is not a multiple of the panel width (a ragged last block) and the nucleus of this family needs row swaps in
    most panels; both kernels pivot by the same rule, so the two inverses agree to rounding."""
    lp = GEN["sparse"](1500, 1500, 30, 52)
    prob = lpgen.build_problem(M.Problem, lp)
    blob = prob.solve(budget=1200).save_basis(2)
    runs = []
    for gj in ("", "1"):
        if gj:
            monkeypatch.setenv("MLP_REINVERT_GJ", gj)
        s = prob.solve_from_basis(blob, budget=0)
        k = s.stats()["nucleus_size"]
        assert k >= 384 and k % 32 != 0, k
        s.continue_solve(40)
        runs.append((s.trace(), s.objective(), s.values()))
    assert [t[:5] for t in runs[0][0]] == [t[:5] for t in runs[1][0]]
    assert abs(runs[0][1] - runs[1][1]) <= 1e-10 * max(1.0, abs(runs[1][1]))
    assert np.abs(runs[0][2] - runs[1][2]).max() <= 1e-8
