"""Pins the CPU oracle (oracle/minilp_oracle.cpp) against EVERY known-answer test the
reference holds for the hot path.  Exact `==` wherever the reference
uses assert_eq!."""
import math

import numpy as np
import pytest

from oracle import minilp_oracle as mo
from oracle.minilp_oracle import EQ, GE, LE, MAXIMIZE, MINIMIZE, Infeasible, Problem, Unbounded

INF = math.inf


def test_readme_doctest():  # lib.rs:27-44 / README.md:28-45  (BASELINE config 1)
    p = Problem(MAXIMIZE)
    x = p.add_var(1.0, (0.0, INF))
    y = p.add_var(2.0, (0.0, 3.0))
    p.add_constraint([(x, 1.0), (y, 1.0)], LE, 4.0)
    p.add_constraint([(x, 2.0), (y, 1.0)], GE, 2.0)
    s = p.solve(trace=True)
    assert s.objective() == 7.0
    assert s[x] == 1.0
    assert s[y] == 3.0
    st = s.stats()  # hand trace SURVEY B.1: exactly one basis change, zero bound flips
    assert st["pivots"] == 1 and st["bound_flips"] == 0


def test_optimize():  # lib.rs:470-482
    p = Problem(MAXIMIZE)
    v1 = p.add_var(3.0, (12.0, INF))
    v2 = p.add_var(4.0, (5.0, INF))
    p.add_constraint([(v1, 1.0), (v2, 1.0)], LE, 20.0)
    p.add_constraint([(v2, -4.0), (v1, 1.0)], GE, -20.0)  # unsorted on purpose
    s = p.solve()
    assert s[v1] == 12.0
    assert s[v2] == 8.0
    assert s.objective() == 68.0


def test_empty_expr_constraints():  # lib.rs:484-526
    trivial = [([], EQ, 0.0), ([], GE, -1.0), ([], LE, 1.0)]
    p = Problem(MINIMIZE)
    p.add_var(1.0, (0.0, INF))
    for e, op, b in trivial:
        p.add_constraint(e, op, b)
    assert p.solve().objective() == 0.0
    s = p.solve()
    for e, op, b in trivial:
        s = s.add_constraint(e, op, b)
    assert s.objective() == 0.0
    infeasible = [([], EQ, 12.0), ([], GE, 34.0), ([], LE, -56.0)]
    for e, op, b in infeasible:
        c = p.clone()
        c.add_constraint(e, op, b)
        with pytest.raises(Infeasible):
            c.solve()
    for e, op, b in infeasible:
        with pytest.raises(Infeasible):
            p.solve().add_constraint(e, op, b)
    p.add_var(-1.0, (0.0, INF))
    with pytest.raises(Unbounded):
        p.solve()


def test_free_variables():  # lib.rs:528-541
    p = Problem(MAXIMIZE)
    v1 = p.add_var(1.0, (0.0, INF))
    v2 = p.add_var(2.0, (-INF, INF))
    p.add_constraint([(v1, 1.0), (v2, 1.0)], LE, 4.0)
    p.add_constraint([(v1, 1.0), (v2, 1.0)], GE, 2.0)
    p.add_constraint([(v1, 1.0), (v2, -1.0)], GE, 0.0)
    s = p.solve()
    assert s[v1] == 2.0
    assert s[v2] == 2.0
    assert s.objective() == 6.0


def test_fix_unfix_var():  # lib.rs:543-576
    p = Problem(MAXIMIZE)
    v1 = p.add_var(1.0, (0.0, 3.0))
    v2 = p.add_var(2.0, (0.0, 3.0))
    p.add_constraint([(v1, 1.0), (v2, 1.0)], LE, 4.0)
    p.add_constraint([(v1, 1.0), (v2, 1.0)], GE, 1.0)
    orig = p.solve()
    s = orig.clone().fix_var(v1, 0.5)
    assert (s[v1], s[v2], s.objective()) == (0.5, 3.0, 6.5)
    s, was = s.unfix_var(v1)
    assert was
    assert (s[v1], s[v2], s.objective()) == (1.0, 3.0, 7.0)
    s = orig.clone().fix_var(v2, 2.5)
    assert (s[v1], s[v2], s.objective()) == (1.5, 2.5, 6.5)
    s, was = s.unfix_var(v2)
    assert (s[v1], s[v2], s.objective()) == (1.0, 3.0, 7.0)


def test_add_constraint():  # lib.rs:578-621
    p = Problem(MINIMIZE)
    v1 = p.add_var(2.0, (0.0, INF))
    v2 = p.add_var(1.0, (0.0, INF))
    p.add_constraint([(v1, 1.0), (v2, 1.0)], LE, 4.0)
    p.add_constraint([(v1, 1.0), (v2, 1.0)], GE, 2.0)
    orig = p.solve()
    s = orig.clone().add_constraint([(v1, -1.0), (v2, 1.0)], LE, 0.0)
    assert (s[v1], s[v2], s.objective()) == (1.0, 1.0, 3.0)
    s = orig.clone().fix_var(v2, 1.5).add_constraint([(v1, -1.0), (v2, 1.0)], LE, 0.0)
    assert (s[v1], s[v2], s.objective()) == (1.5, 1.5, 4.5)
    s = orig.clone().add_constraint([(v1, -1.0), (v2, 1.0)], GE, 3.0)
    assert (s[v1], s[v2], s.objective()) == (0.0, 3.0, 3.0)


def test_gomory_cut():  # lib.rs:623-645
    p = Problem(MINIMIZE)
    v1 = p.add_var(0.0, (0.0, INF))
    v2 = p.add_var(-1.0, (0.0, INF))
    p.add_constraint([(v1, 3.0), (v2, 2.0)], LE, 6.0)
    p.add_constraint([(v1, -3.0), (v2, 2.0)], LE, 0.0)
    s = p.solve()
    assert (s[v1], s[v2], s.objective()) == (1.0, 1.5, -1.5)
    s = s.add_gomory_cut(v2)
    assert abs(s[v1] - 2.0 / 3.0) < 1e-8
    assert s[v2] == 1.0
    assert s.objective() == -1.0
    s = s.add_gomory_cut(v1)
    assert abs(s[v1] - 1.0) < 1e-8
    assert s[v2] == 1.0
    assert s.objective() == -1.0


def _dense_rows(s, n_rows, n_cols):
    ip = s.state("csr_indptr").astype(int)
    ix = s.state("csr_indices").astype(int)
    d = s.state("csr_data")
    out = np.zeros((n_rows, n_cols))
    for r in range(n_rows):
        for q in range(ip[r], ip[r + 1]):
            out[r, ix[q]] = d[q]
    return out


def test_solver_initialize():  # solver.rs:1391-1441 (white-box)
    p = Problem(MINIMIZE)
    p.add_var(2.0, (-INF, 0.0))
    p.add_var(1.0, (5.0, INF))
    p.add_constraint([(0, 1.0), (1, 1.0)], LE, 6.0)
    p.add_constraint([(0, 1.0), (1, 2.0)], LE, 8.0)
    p.add_constraint([(0, 1.0), (1, 1.0)], GE, 2.0)
    p.add_constraint([(1, 1.0)], EQ, 3.0)
    s = p.try_new()
    flags = s.state("flags")
    assert flags[0] == 0 and flags[1] == 0
    assert list(s.state("orig_obj_coeffs")) == [2.0, 1.0, 0.0, 0.0, 0.0, 0.0]
    assert list(s.state("orig_var_mins")) == [-INF, 5.0, 0.0, 0.0, -INF, 0.0]
    assert list(s.state("orig_var_maxs")) == [0.0, INF, INF, INF, 0.0, 0.0]
    ref = [[1, 1, 1, 0, 0, 0], [1, 2, 0, 1, 0, 0], [1, 1, 0, 0, 1, 0], [0, 1, 0, 0, 0, 1]]
    assert (_dense_rows(s, 4, 6) == np.array(ref, dtype=float)).all()
    assert list(s.state("orig_rhs")) == [6.0, 8.0, 2.0, 3.0]
    assert list(s.state("basic_vars")) == [2, 3, 4, 5]
    assert list(s.state("basic_var_vals")) == [1.0, -2.0, -3.0, -2.0]
    assert list(s.state("dual_edge_sq_norms")) == [1.0, 1.0, 1.0, 1.0]
    assert list(s.state("nb_vars")) == [0, 1]
    assert list(s.state("nb_var_obj_coeffs")) == [-1.0, 1.0]
    assert list(s.state("nb_var_vals")) == [0.0, 5.0]
    assert list(s.state("primal_edge_sq_norms")) == [4.0, 8.0]
    assert s.state("cur_obj_val")[0] == 0.0


def test_solver_initial_solve():  # solver.rs:1443-1479 (white-box)
    p = Problem(MINIMIZE)
    p.add_var(-3.0, (-INF, 20.0))
    p.add_var(-4.0, (5.0, INF))
    p.add_constraint([(0, 1.0), (1, 1.0)], LE, 20.0)
    p.add_constraint([(0, -1.0), (1, 4.0)], LE, 20.0)
    s = p.solve(trace=True)
    flags = s.state("flags")
    assert flags[0] == 1 and flags[1] == 1
    assert list(s.state("basic_vars")) == [0, 1]
    assert list(s.state("basic_var_vals")) == [12.0, 8.0]
    assert list(s.state("nb_vars")) == [2, 3]
    assert list(s.state("nb_var_vals")) == [0.0, 0.0]
    assert list(s.state("nb_var_obj_coeffs")) == [3.2, 0.2]
    assert s.state("cur_obj_val")[0] == -68.0
    # SURVEY B.1 hand trace: dual pivot (row 0, col 0) then primal pivot (row 1, col 1)
    tr = s.trace()
    assert [(t[0], t[1], t[2]) for t in tr] == [(1, 0, 0), (0, 1, 1)]

    q = Problem(MINIMIZE)
    q.add_var(1.0, (0.0, INF))
    q.add_var(1.0, (0.0, INF))
    q.add_constraint([(0, 1.0), (1, 1.0)], GE, 10.0)
    q.add_constraint([(0, 1.0), (1, 1.0)], LE, 5.0)
    with pytest.raises(Infeasible):
        q.solve()


def _csc_from_triplets(n_rows, n_cols, trip):
    """sprs TriMat::to_csc: ascending row inside each column."""
    cols = [[] for _ in range(n_cols)]
    for r, c, v in trip:
        cols[c].append((r, v))
    indptr, rows, vals = [0], [], []
    for c in range(n_cols):
        for r, v in sorted(cols[c]):
            rows.append(r)
            vals.append(v)
        indptr.append(len(rows))
    return indptr, rows, vals


def _select_cols(indptr, rows, vals, sel):
    ip, rr, vv = [0], [], []
    for c in sel:
        rr += rows[indptr[c]:indptr[c + 1]]
        vv += vals[indptr[c]:indptr[c + 1]]
        ip.append(len(rr))
    return ip, rr, vv


def test_lu_simple():  # lu.rs:479-552
    trip = [(0, 1, 2.0), (0, 0, 2.0), (0, 2, 123.0), (1, 2, 456.0), (1, 3, 1.0), (2, 1, 4.0), (2, 0, 3.0),
            (2, 2, 789.0), (2, 3, 1.0)]
    ip, rr, vv = _select_cols(*_csc_from_triplets(3, 4, trip), [1, 0, 3])
    lu = mo.LU(3, ip, rr, vv, 0.9)
    L, Ld = lu.factor(0)
    assert (L == np.array([[0, 0, 0], [0.5, 0, 0], [0, 0, 0]])).all()
    assert Ld is None
    U, Ud = lu.factor(1)
    assert (U == np.array([[0, 3.0, 1.0], [0, 0, -0.5], [0, 0, 0]])).all()
    assert list(Ud) == [4.0, 0.5, 1.0]
    pm = lu.perms()
    assert list(pm["row_new2orig"]) == [2, 0, 1]
    assert list(pm["col_new2orig"]) == [0, 1, 2]
    assert list(lu.solve_dense([6.0, 3.0, 13.0])) == [1.0, 2.0, 3.0]
    assert list(lu.solve_dense([14.0, 11.0, 5.0], transp=True)) == [1.0, 2.0, 3.0]

    def dense(idx, val):
        d = np.zeros(3)
        d[idx] = val
        return list(d)

    assert dense(*lu.solve_sparse([1], [-1.0])) == [1.0, -1.0, -1.0]
    assert dense(*lu.solve_sparse([1, 2], [-1.0, 1.0], transp=True)) == [-2.0, 0.0, 1.0]


def test_lu_singular():  # lu.rs:554-609
    sym = [(0, 0, 1.0), (1, 0, 1.0), (1, 1, 2.0), (1, 2, 3.0)]
    with pytest.raises(ArithmeticError):
        mo.LU(3, *_csc_from_triplets(3, 3, sym), 0.9)
    num = sym + [(2, 0, 2.0), (2, 1, 2.0), (2, 2, 3.0)]
    with pytest.raises(ArithmeticError):
        mo.LU(3, *_csc_from_triplets(3, 3, num), 0.9)


@pytest.mark.parametrize("seed", [12345, 1, 2, 3, 4])
def test_lu_rand(seed):  # lu.rs:611-704 (property; the reference's Pcg64 stream is not reproducible here)
    rng = np.random.default_rng(seed)
    n = 10
    for _ in range(20):
        A = np.where(rng.integers(0, 2, (n, n)) == 0, rng.random((n, n)), 0.0)
        if abs(np.linalg.det(A)) < 1e-6:
            continue
        trip = [(r, c, A[r, c]) for r in range(n) for c in range(n) if A[r, c] != 0.0]
        lu = mo.LU(n, *_csc_from_triplets(n, n, trip), 0.1)
        L, _ = lu.factor(0)
        U, Ud = lu.factor(1)
        pm = lu.perms()
        LU = (L + np.eye(n)) @ (U + np.diag(Ud))
        for c in range(n):
            permuted = np.zeros(n)
            for r in range(n):
                permuted[int(pm["row_orig2new"][r])] = A[r, c]
            assert np.abs(LU[:, int(pm["col_orig2new"][c])] - permuted).sum() < 1e-5
        b = rng.random(n)
        assert np.linalg.norm(b - A @ lu.solve_dense(b)) < 1e-5
        assert np.linalg.norm(b - A.T @ lu.solve_dense(b, transp=True)) < 1e-5
        sel = np.nonzero(rng.integers(0, 3, n) == 0)[0]
        sb = np.zeros(n)
        sb[sel] = rng.random(len(sel))
        for transp in (False, True):
            idx, val = lu.solve_sparse(sel, sb[sel], transp=transp)
            x = np.zeros(n)
            x[idx] = val
            M = A.T if transp else A
            assert np.abs(sb - M @ x).sum() < 1e-5


def test_mat_transpose():  # sparse.rs:344-359
    oi, ox, od = mo.sparse_transpose(2, [0, 2, 3, 4], [0, 1, 1, 0], [1.1, 2.2, 3.3, 4.4])
    assert list(oi) == [0, 2, 4]
    assert list(ox) == [2, 0, 1, 0]
    assert list(od) == [4.4, 1.1, 3.3, 2.2]


MPS_TESTPROB = """\
* test file
NAME          TESTPROB
ROWS
 N  COST
 L  LIM1
 G  LIM2
 E  MYEQN
COLUMNS
    XONE      COST                 1   LIM1                 1
    XONE      LIM2                 1

    YTWO      COST                 4   LIM1                 1
    YTWO      MYEQN               -1

    ZTHREE    COST                 9   LIM2                 1
    ZTHREE    MYEQN                1
RHS
    RHS1      LIM1                 5   LIM2                10
    RHS1      MYEQN                7
BOUNDS
 UP BND1      XONE                 4
 LO BND1      YTWO                -1
 UP BND1      YTWO                 1
ENDATA
"""


def test_parse_mps_file():  # mps.rs:437-476 (the MPS text is the reference test's DATA fixture)
    f = mo.MpsFile(MPS_TESTPROB, MINIMIZE)
    assert f.problem_name == "TESTPROB"
    assert len(f.variables) == 3
    s = f.problem.solve()
    assert s[f.variables["XONE"]] == 4.0
    assert s[f.variables["YTWO"]] == -1.0
    assert s[f.variables["ZTHREE"]] == 6.0
    assert s.objective() == 54.0
