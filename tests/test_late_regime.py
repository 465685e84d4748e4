"""GPU: the large-nucleus regime of config 4 at FULL SIZE with the DEFAULT machinery (nothing forced): the two committed
mid-solve bases that bench.py times (tests/golden/cfg4_basis_p45000 / p240000: nucleus 9 999 / 20 493, capacity 14 336 /
24 576: growth by a quarter from 8 192 slots on) are loaded through mlp_problem_solve_from_basis and continued.  From capacity 8 192 on the engine runs the
delayed-update mode (k_stream_w every pivot, k_fold_w every 32nd), the blocked F push, the banded sweep in locality order
with the packed non-basic copy — the kernels the solve spends > 90 % of its wall time in.

(a) size-independent properties after 256 pivots: primal feasibility, monotone objective, c.x == objective, a fresh
    inverse agrees with the incrementally maintained one, the two-way pivot check stayed at rounding level;
(b) 8 pivots stepped stage by stage (mlp_engine_stage): every solve the iteration performs is checked against the
    constraint matrix ITSELF with scipy on the box — B alpha_q = a_q, B^T rho = e_r, B tau = rho, B^T v = alpha_q,
    alpha_r = N^T rho — as a componentwise backward error.  Nothing of the engine (no W, no other mode) is on the
    reference side.  solver.rs:671-693, 1106-1174;
(c) sharding at config-4 size lives in tests/test_dist_gpu.py (2 ranks and 8 ranks on the one GPU, identical traces).
"""
import gzip
import os

import numpy as np
import pytest
import scipy.sparse as sp

import minilp_amd as M
from minilp_amd import lpgen
from tests.common import ROOT, check_feasible, objective_of

pytestmark = pytest.mark.gpu
A = M.api
MID = os.path.join(ROOT, "tests", "golden", "cfg4_basis_p45000.bin.gz")
LATE = os.path.join(ROOT, "tests", "golden", "cfg4_basis_p240000.bin.gz")


@pytest.fixture(scope="module")
def cfg4():
    lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
    return lp, lpgen.build_problem(M.Problem, lp)


def _load(prob, path, **kw):
    with gzip.open(path, "rb") as f:
        return prob.solve_from_basis(f.read(), budget=0, **kw)


@pytest.mark.parametrize("path,k0,cap", [(MID, 9999, 14336), (LATE, 20493, 24576)], ids=["mid k=9999", "late k=20493"])
def test_default_machinery_from_saved_basis_keeps_the_invariants(cfg4, path, k0, cap):
    lp, prob = cfg4
    s = _load(prob, path, trace=True)
    st = s.stats()
    assert st["nucleus_size"] == k0 and st["nucleus_capacity"] == cap and st["banded_sweep"] == 1
    obj0 = s.objective()
    s.continue_solve(256)
    st = s.stats()
    assert st["iterations"] == 256 and s.budget_exhausted
    tr = s.trace()
    assert len(tr) == 256 and all(t[0] == 0 for t in tr)                      # primal loop
    objs = np.array([obj0] + [-t[6] for t in tr])                             # trace holds the minimised form (lib.rs:235-238)
    assert (np.diff(objs) >= -1e-9 * np.abs(objs[:-1])).all()                 # Maximize: never decreases
    assert objs[-1] > objs[0]
    x = s.values()
    check_feasible(lp, x, tol=1e-7)
    assert abs(objective_of(lp, x) - s.objective()) <= 1e-9 * abs(s.objective())
    assert abs(s.objective() - objs[-1]) <= 1e-9 * abs(objs[-1])
    assert st["max_pivot_err"] < 1e-9, st["max_pivot_err"]                    # alpha_q[r] (FTRAN) against alpha_r[q] (tableau row)
    assert sum(st["kase"]) == st["basis_changes"]
    assert s.stats()["nucleus_size"] >= k0 - 256
    drift = s.reinvert()                                                      # folds the pending terms, then max |W - W_fresh|
    scale = float(s.state("reinvert_scale")[0])                               # max |W_fresh|: the inverse's entries are far from O(1) here
    # two inversions of the same K agree to eps * cond(K) * max|W| at best (the nucleus of config 4 is ill-conditioned:
    # measured 2.6e-8 of max|W| at k = 20 493); the check guards against a WRONG update, which shows up at O(1)
    assert drift <= 5e-7 * max(1.0, scale), (drift, scale)
    print(f"{os.path.basename(path)}: 256 pivots, objective {objs[0]:.6f} -> {objs[-1]:.6f}, max_pivot_err {st['max_pivot_err']:.2e}, "
          f"max |W - W_fresh| {drift:.2e} (max |W| {scale:.2e}), cases {st['kase']}")


def test_t_K_riding_in_the_ratio_launch_changes_no_bit(cfg4, monkeypatch):
    """Large nucleus, lazy primal iteration: t_K = alpha_K - F^T y_S is formed by blocks that ride behind the ratio blocks of
    k_ratio_primal_fused (y_S left by row by the F push's combine) instead of in the BTRAN launch (MLP_TK_RIDE=0).  The same
    quotients, the same sums in the same order: 96 pivots from the mid basis must agree bit for bit."""
    lp, prob = cfg4
    monkeypatch.setenv("MLP_RK_RIDE", "0")  # (rho_K riding behind the v tail needs the t_K ride and is not bit-neutral on folding pivots: next test)
    monkeypatch.setenv("MLP_FPULL", "0")    # (round 6: with the t_K ride the F product is PULLED — another summation order; its own tests are below)
    runs = []
    for on in ("1", "0"):
        monkeypatch.setenv("MLP_TK_RIDE", on)
        s = _load(prob, MID, trace=True)
        s.continue_solve(96)
        runs.append((s.trace(), s.objective(), s.values().tobytes()))
    assert runs[0][0] == runs[1][0]
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]


def test_rho_K_riding_behind_the_v_tail_takes_the_same_pivots(cfg4, monkeypatch):
    """Large nucleus, lazy primal iteration: nothing between the ratio test and the tableau row reads rho, so rho_K is formed by
    blocks riding behind the v tail of the pass (k_post_fused) instead of in a BTRAN launch of its own (MLP_RK_RIDE=0).  On a
    folding pivot those blocks read the FOLDED inverse (W0 with the 32 terms added in) where the BTRAN launch read W0 and the
    terms separately: rounding-level differences from the first fold on, the same pivots and the same end point (96 pivots)."""
    lp, prob = cfg4
    runs = []
    for on in ("1", "0"):
        monkeypatch.setenv("MLP_RK_RIDE", on)
        s = _load(prob, MID, trace=True)
        s.continue_solve(96)
        runs.append((s.trace(), s.objective(), s.values(), s.stats()["max_pivot_err"]))
    assert [t[:5] for t in runs[0][0]] == [t[:5] for t in runs[1][0]]
    assert runs[0][0][:30] == runs[1][0][:30]          # bit for bit until the first fold
    assert abs(runs[0][1] - runs[1][1]) <= 1e-11 * abs(runs[0][1])
    assert np.abs(runs[0][2] - runs[1][2]).max() <= 1e-9
    assert runs[0][3] < 1e-9 and runs[1][3] < 1e-9


@pytest.mark.parametrize("path,pivots", [(MID, 1100), (LATE, 160)], ids=["mid, across a rebuild of the packed copy", "late"])
def test_pulled_F_product_takes_the_pushed_form_s_pivots(cfg4, monkeypatch, path, pivots):
    """Round 6 (csrc/fpull.inc): in the large-nucleus lazy primal iteration the singleton part of the FTRAN, alpha_S = D^-1 (a_q - F alpha_K)
    (solver.rs:671-677 -> 1305-1319), is PULLED per singleton row from a row-major packed copy of the nucleus columns inside the launch
    that runs Harris pass 1, instead of pushed through LDS atomics (k_push_stage1 + k_push_combine) in front of k_ratio_primal_fused
    (MLP_FPULL=0).  Other summation orders, the same numbers to rounding: both forms must take the same pivots from the committed
    basis — over a stretch that appends > 1 000 entering columns to the copy and crosses its periodic rebuild (every 1 024 pivots) —
    end at the same objective, and keep x_B consistent with x_B = B^-1 (b - N x_N) recomputed from scratch (a wrong alpha_q in ANY
    pivot shows there: the cooperative-push bug found while building this did, at 1e-9 against 1e-11)."""
    lp, prob = cfg4
    runs = []
    for on in ("1", "0"):
        monkeypatch.setenv("MLP_FPULL", on)
        s = _load(prob, path, trace=True)
        s.continue_solve(pivots)
        fp = s.state("fpull")
        assert int(fp[0]) == int(on) and int(fp[3]) == int(on)          # the path under test is the one that ran
        if on == "1":
            assert int(fp[1]) >= (2 if pivots > 1024 else 1)             # builds of the packed copy: load, and the periodic one
        x1 = s.values().copy()
        s.recompute_basic_values()
        gap = float(np.abs(x1 - s.values()).max())
        runs.append(([t[:5] for t in s.trace()], s.objective(), x1, s.stats()["max_pivot_err"], gap))
    assert runs[0][0] == runs[1][0]
    assert abs(runs[0][1] - runs[1][1]) <= 1e-11 * abs(runs[0][1])
    assert np.abs(runs[0][2] - runs[1][2]).max() <= 1e-8
    assert runs[0][3] < 1e-9 and runs[1][3] < 1e-9
    assert runs[0][4] <= max(1e-7, 20.0 * runs[1][4]), (runs[0][4], runs[1][4])   # maintained x_B vs recomputed: the pulled form no worse


def test_pulled_F_product_makes_the_late_pivot_reproducible_bit_for_bit(cfg4):
    """The pushed form's LDS float atomics made alpha_S — hence every later number — reproducible only to rounding (the pivot count of a
    config-4 solve varied by ~1 % from run to run).  The pull has a fixed summation order: two runs of 200 pivots from the late basis
    with the default machinery must agree in every bit of the trace (pivot elements, objectives), of x and of the objective."""
    lp, prob = cfg4
    runs = []
    for _ in range(2):
        s = _load(prob, LATE, trace=True)
        s.continue_solve(200)
        assert int(s.state("fpull")[0]) == 1
        runs.append((s.trace(), s.objective(), s.values().tobytes()))
    assert runs[0][0] == runs[1][0]
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]


def _backward_error(resid, *abs_terms):
    den = sum(abs_terms)
    den = np.where(den > 0, den, 1.0)
    return float((np.abs(resid) / den).max())


@pytest.mark.parametrize("path", [MID, LATE], ids=["mid", "late"])
def test_stepped_stages_satisfy_the_defining_equations_against_the_matrix(cfg4, path):
    lp, prob = cfg4
    m, n = lp["m"], lp["n"]
    Acsc = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n)).tocsc()
    Aext = sp.hstack([Acsc, sp.identity(m, format="csc")], format="csc")      # slack coefficient +1 (solver.rs:250)
    Aabs = abs(Aext)
    s = _load(prob, path)
    st, info = s.engine_open()
    assert st == A.ITER_PIVOT and info["phase"] == 0
    worst = {}
    done = 0
    while done < 8:
        bv = s.state("basic_vars").astype(np.int64)
        nb = s.state("nb_vars").astype(np.int64)
        B, Babs = Aext[:, bv], Aabs[:, bv]
        q = int(info["col"])
        aq = np.asarray(Aext[:, int(nb[q])].todense()).ravel()
        got = {}
        while True:
            stage = info["next_stage"]
            st, info = s.engine_stage(stage)
            if stage == A.STAGE_FTRAN:
                got["alpha"] = s.state("col_coeffs")
            elif stage == A.STAGE_RATIO:
                r = int(info["row"])
            elif stage == A.STAGE_BTRAN:
                got["rho"] = s.state("inv_basis_row_coeffs")
            elif stage == A.STAGE_BASIS:
                got["tau"], got["v"] = s.state("tau"), s.state("v")
            elif stage == A.STAGE_ROW:
                got["alpha_r"] = s.state("row_coeffs")
            if stage == A.STAGE_APPLY or st not in (A.ITER_PIVOT, A.ITER_FLIP):
                break
        assert stage == A.STAGE_APPLY and st in (A.ITER_PIVOT, A.ITER_FLIP), (stage, st)
        if r < 0:
            continue  # bound flip: no BTRAN / basis update in this iteration
        alpha, rho, tau, v, alpha_r = got["alpha"], got["rho"], got["tau"], got["v"], got["alpha_r"]
        for nm_, vec_ in got.items():
            bad_ = np.nonzero(~np.isfinite(vec_))[0]
            assert len(bad_) == 0, (nm_, "non-finite entries", len(bad_), bad_[:8].tolist(), "pivot", done, "q", q, "r", r)
        e_r = np.zeros(m)
        e_r[r] = 1.0
        errs = dict(
            ftran=_backward_error(B @ alpha - aq, Babs @ np.abs(alpha), np.abs(aq)),               # B alpha_q = a_q
            btran=_backward_error(B.T @ rho - e_r, Babs.T @ np.abs(rho), e_r),                     # B^T rho_r = e_r
            tau=_backward_error(B @ tau - rho, Babs @ np.abs(tau), np.abs(rho)),                   # B tau = rho_r   (solver.rs:1157)
            v=_backward_error(B.T @ v - alpha, Babs.T @ np.abs(v), np.abs(alpha)),                 # B^T v = alpha_q (solver.rs:1114)
        )
        N, Nabs = Aext[:, nb], Aabs[:, nb]
        want = N.T @ rho
        errs["row"] = float((np.abs(alpha_r - want) / np.maximum(Nabs.T @ np.abs(rho), 1e-300)).max())  # alpha_r = N^T rho
        assert abs(alpha[r] - alpha_r[q]) <= 1e-9 * max(1.0, abs(alpha[r]))                        # the pivot element, both ways
        for kname, e in errs.items():
            worst[kname] = max(worst.get(kname, 0.0), e)
            assert e < 1e-8, (kname, e, done)
        done += 1
    print(os.path.basename(path), "8 stepped pivots; worst componentwise backward errors:", {k: f"{e:.1e}" for k, e in worst.items()})
