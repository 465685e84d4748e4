"""GPU: the COMPACT FACTOR of the basis (minilp_amd/csrc/factor.inc, HISTORY.md §2.6) — the second
representation of B^-1 next to the explicit nucleus inverse: an iterated column-singleton peel of the basis (a triangular
factor without fill: lu.rs:118-304 / ordering.rs:4-21 carried to the fixed point), level-scheduled pulls for FTRAN / BTRAN
(lu.rs:79-106, 432-463) and the eta transformations since the last refactorisation as additive rank-1 terms
(solver.rs:1274-1284).  Gates: the oracle's pivot sequence on the non-degenerate families (the operator is the same
whichever representation applies it), objective / values at the optimum, the defining equations of every solve against the
matrix itself, and the switches between the two representations."""
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import minilp_amd as M
from minilp_amd import api as A
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import X_ATOL, check_feasible, obj_close

pytestmark = pytest.mark.gpu


def _pair(lp, **kw):
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True, **kw)
    return so, sg


def _sb(s):
    return dict(zip(("in_use", "factorisations", "fallbacks", "rounds", "tail", "skipped", "failed_bump"), s.state("factor_sb").astype(int).tolist()))


@pytest.mark.parametrize("args,tight", [((300, 300, 3, 7), 1.0), ((800, 1000, 4, 11), 0.5), ((2500, 2500, 4, 3), 0.4), ((4000, 5000, 5, 9), 0.4)], ids=str)
def test_transport_family_on_the_compact_factor_takes_the_oracles_pivots(monkeypatch, args, tight):
    """Network-with-gains instances (every column has two entries: every basis is a forest, its peel leaves no bump): the whole
    solve — the dual simplex — runs on the compact factor from the slack basis on (MLP_FACTOR=1), refactoring every 48 pivots (64 when the peel is deep)."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    lp = lpgen.gen_transport_lp(*args, tight=tight)
    so, sg = _pair(lp)
    st = sg.stats()
    print(args, "pivots", st["iterations"], "refactorisations", st["factor_refactors"], "levels", st["factor_levels"], "switches", st["factor_switches"])
    assert st["factor_active"] == 1 and st["factor_switches"] == 1 and st["factor_refactors"] >= st["iterations"] // 64
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL
    check_feasible(lp, sg.values())
    assert sg.stats()["max_pivot_err"] < 1e-9      # the pivot element from FTRAN and from the tableau row agree
    bo, bg = so.state("dual_edge_sq_norms"), sg.state("dual_edge_sq_norms")
    assert np.abs(bo - bg).max() <= 1e-8 * max(1.0, np.abs(bo).max())
    # round 6: the dual Harris test walks the LISTED non-zeros of the tableau row (solver.rs:962-1002) — on a network every row of the
    # tableau has a handful of them — instead of two grid-wide passes over all columns; most pivots of this solve must have taken that form
    assert int(sg.state("dual_list_tests")[0]) >= st["iterations"] // 2, (sg.state("dual_list_tests"), st["iterations"])


@pytest.mark.parametrize("J", [1, 5, 64])
def test_refactor_period_does_not_change_the_pivot_sequence(monkeypatch, J):
    """MLP_FACTOR_J: one term (refactor after every pivot), five, sixty-four pending rank-1 terms."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_J", str(J))
    lp = lpgen.gen_transport_lp(600, 700, 4, 5, tight=0.45)
    so, sg = _pair(lp)
    st = sg.stats()
    print("J", J, "pivots", st["iterations"], "refactorisations", st["factor_refactors"], "largest bump", st["factor_bump_max"])
    assert st["factor_active"] == 1
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    if J == 1:
        # refactoring after every pivot samples EVERY basis of the solve: some of them carry one-cycle components (a network with
        # gains has them), which the peel leaves as a bump — solved through its explicit inverse
        assert st["factor_refactors"] >= st["basis_changes"] and st["factor_bump_max"] > 0


def test_a_basis_that_stops_peeling_goes_back_to_the_explicit_inverse(monkeypatch):
    """Config-4 family (random sparse rows): the slack basis peels (one level) and so do the first bases, then the nucleus
    closes cycles — a bump — and the solve continues on the explicit inverse; the oracle's pivots throughout (primal loop
    with both steepest-edge recurrences: v = B^-T alpha_q and tau = B^-1 rho through the factor while it lasts)."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_BUMP", "8")   # (default 256 columns: beyond it the bump's O(b^2) solve in one workgroup stops paying)
    lp = lpgen.gen_sparse_lp(400, 300, 12, 7)
    so, sg = _pair(lp)
    st = sg.stats()
    print("pivots", st["iterations"], "refactorisations", st["factor_refactors"], "switches", st["factor_switches"], "largest bump carried", st["factor_bump_max"])
    assert st["factor_switches"] >= 2 and st["factor_active"] == 0 and st["factor_refactors"] >= 1
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL


def test_a_growing_bump_is_carried_through_its_explicit_inverse(monkeypatch):
    """The same instance with the default bump limit: the nucleus of a random sparse model is all cycles, so the bump grows to the
    size of the nucleus and every solve goes through B0 = [[U, F], [0, K]] with the dense K^-1 — FTRAN solves the bump first,
    BTRAN last.  The oracle's pivots, primal loop with both steepest-edge solves."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_J", "7")
    lp = lpgen.gen_sparse_lp(400, 300, 12, 7)
    so, sg = _pair(lp)
    st = sg.stats()
    print("pivots", st["iterations"], "refactorisations", st["factor_refactors"], "largest bump", st["factor_bump_max"], "levels", st["factor_levels"])
    assert st["factor_active"] == 1 and st["factor_bump_max"] >= 20
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL


def test_two_phase_instance_on_the_compact_factor(monkeypatch):
    """Neither primal nor dual feasible at the start (lpgen.gen_twophase_lp): the dual loop on the artificial objective,
    recalc_obj_coeffs through the factor (dense-rhs BTRAN y = B^-T c_B), then the primal loop with steepest edge — v = B^-T alpha_q
    and tau = B^-1 rho are two more level-scheduled solves per pivot; the rows are random, so the basis carries a bump."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    lp = lpgen.gen_twophase_lp(300, 260, 8, 6)
    so, sg = _pair(lp)
    st = sg.stats()
    print("pivots", st["iterations"], "primal", st["primal_iters"], "dual", st["dual_iters"], "refactorisations", st["factor_refactors"], "active", st["factor_active"],
          "largest bump", st["factor_bump_max"])
    assert st["primal_iters"] > 0 and st["dual_iters"] > 0 and st["factor_active"] == 1
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL


def test_auto_selection_switches_when_the_nucleus_outgrows_the_threshold(monkeypatch):
    """Default policy with the threshold lowered (MLP_FACTOR_FROM: 8 192 slots by default): the solve starts on the explicit
    inverse, and when the nucleus needs more capacity the current basis is peeled — it leaves no bump, so the solve continues
    on the compact factor, same pivots as the oracle."""
    monkeypatch.setenv("MLP_FACTOR_FROM", "256")
    monkeypatch.setenv("MLP_HYPER", "0")
    lp = lpgen.gen_transport_lp(1500, 1500, 4, 21, tight=0.45)
    so, sg = _pair(lp)
    st = sg.stats()
    print("pivots", st["iterations"], "refactorisations", st["factor_refactors"], "levels", st["factor_levels"], "nucleus capacity", st["nucleus_capacity"])
    assert st["factor_active"] == 1 and st["factor_switches"] == 1 and st["iterations"] > 1000
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(sg.values() - so.values()).max() <= X_ATOL


def test_auto_selection_keeps_the_explicit_inverse_for_a_nucleus_that_does_not_peel(monkeypatch):
    """Config-4 family (random sparse rows): when the nucleus outgrows the threshold the peel of the current basis leaves a bump of
    hundreds of columns (the nucleus is all cycles) — far beyond the 32 the automatic selection accepts — so the solve stays on
    the explicit inverse; the attempt costs one peel and is repeated only after the nucleus has doubled."""
    monkeypatch.setenv("MLP_FACTOR_FROM", "1024")
    lp = lpgen.gen_sparse_lp(6000, 3000, 12, 4)
    so, sg = _pair(lp)
    st = sg.stats()
    print("pivots", st["iterations"], "nucleus", st["nucleus_size"], "switches", st["factor_switches"], "peels tried", st["factor_refactors"])
    assert st["factor_active"] == 0 and st["factor_switches"] == 0 and st["nucleus_size"] > 1024
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]


_NORMWISE = False


def _backward_error(resid, *abs_terms):
    den = sum(abs_terms)
    if _NORMWISE:  # (a basis with cycles: entries that are zero in exact arithmetic come out as rounding noise, and the componentwise
        # measure divides noise by noise there; the normwise one does not)
        return float(np.abs(resid).max() / max(float(den.max()), 1e-300))
    den = np.where(den > 0, den, 1.0)
    return float((np.abs(resid) / den).max())


def test_stepped_stages_on_the_compact_factor_satisfy_the_defining_equations(monkeypatch):
    """Engine-level stepping with the compact factor and several pending rank-1 terms: after every stage the vector it produced
    is checked against the constraint matrix itself by scipy — B alpha_q = a_q, B^T rho = e_r, B tau = rho, alpha_r = N^T rho —
    nothing of the engine on the reference side (the same check tests/test_late_regime.py makes on the explicit inverse)."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_J", "8")
    _stepped_check(lpgen.gen_transport_lp(900, 900, 4, 17, tight=0.45), 300)    # 300 pivots in: pending terms, several levels


@pytest.mark.parametrize("carrier", ["sparse_lu", "dense_inverse"])
def test_stepped_stages_through_the_bump_satisfy_the_defining_equations(monkeypatch, carrier):
    """The same check late in config 3's solve on the compact factor, where the peel leaves a bump of ~30 columns — carried as a sparse
    LU with fill (factor_sb.inc) or through its dense inverse: every FTRAN / BTRAN / tau solve of 20 stepped pivots against the
    matrix itself."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_J", "8")
    monkeypatch.setenv("MLP_FACTOR_SB_FROM", "2")
    if carrier == "dense_inverse":
        monkeypatch.setenv("MLP_FACTOR_SB", "0")
    monkeypatch.setattr(sys.modules[__name__], "_NORMWISE", True)
    s = _stepped_check(lpgen.gen_mixed_lp(6000, 10000, 4, 3), 3700)
    sb = _sb(s)
    print("sparse bump", sb, "bump", s.stats()["factor_bump"])
    assert s.stats()["factor_bump"] >= 2
    if carrier == "sparse_lu":
        assert sb["factorisations"] >= 1 and sb["fallbacks"] == 0 and sb["in_use"] == 1
    else:
        assert sb["factorisations"] == 0


def _stepped_check(lp, budget):
    m, n = lp["m"], lp["n"]
    Acsc = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n)).tocsc()
    Aext = sp.hstack([Acsc, sp.identity(m, format="csc")], format="csc")
    Aabs = abs(Aext)
    s = lpgen.build_problem(M.Problem, lp).solve(budget=budget)
    assert s.stats()["factor_active"] == 1
    st, info = s.engine_open()
    assert st == A.ITER_PIVOT and info["phase"] == 1
    worst = {}
    for done in range(20):
        bv = s.state("basic_vars").astype(np.int64)
        nb = s.state("nb_vars").astype(np.int64)
        B, Babs = Aext[:, bv], Aabs[:, bv]
        r = int(info["row"])
        got = {}
        while True:
            stage = info["next_stage"]
            st, info = s.engine_stage(stage)
            if stage == A.STAGE_BTRAN:
                got["rho"] = s.state("inv_basis_row_coeffs")
            elif stage == A.STAGE_ROW:
                got["alpha_r"] = s.state("row_coeffs")
            elif stage == A.STAGE_RATIO:
                q = int(info["col"])
            elif stage == A.STAGE_FTRAN:
                got["alpha"] = s.state("col_coeffs")
            elif stage == A.STAGE_BASIS:
                got["tau"] = s.state("tau")
            if stage == A.STAGE_APPLY or st not in (A.ITER_PIVOT, A.ITER_FLIP):
                break
        assert stage == A.STAGE_APPLY and st in (A.ITER_PIVOT, A.ITER_FLIP, A.ITER_FEASIBLE), (stage, st)
        aq = np.asarray(Aext[:, int(nb[q])].todense()).ravel()
        e_r = np.zeros(m)
        e_r[r] = 1.0
        alpha, rho, tau, alpha_r = got["alpha"], got["rho"], got["tau"], got["alpha_r"]
        errs = dict(ftran=_backward_error(B @ alpha - aq, Babs @ np.abs(alpha), np.abs(aq)),
                    btran=_backward_error(B.T @ rho - e_r, Babs.T @ np.abs(rho), e_r),
                    tau=_backward_error(B @ tau - rho, Babs @ np.abs(tau), np.abs(rho)))
        N, Nabs = Aext[:, nb], Aabs[:, nb]
        errs["row"] = _backward_error(alpha_r - N.T @ rho, np.maximum(Nabs.T @ np.abs(rho), 1e-300))
        for kname, e in errs.items():
            worst[kname] = max(worst.get(kname, 0.0), e)
            assert e < 1e-9, (kname, e, done)
        if st == A.ITER_FEASIBLE:
            break
    print("stepped pivots on the compact factor; worst componentwise backward errors:", {k: f"{e:.1e}" for k, e in worst.items()})
    return s


def test_clone_fix_and_unfix_on_the_compact_factor(monkeypatch):
    """Solution::clone re-peels the same basis; fix_var / unfix_var (solver.rs:378-438) run their forced dual pivot and the
    re-solves on the factor; add_constraint (a new row) re-peels the extended basis and stays on the factor (round 5)."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    lp = lpgen.gen_transport_lp(400, 500, 4, 23, tight=0.5)
    so = lpgen.build_problem(O.Problem, lp).solve()
    sg = lpgen.build_problem(M.Problem, lp).solve()
    assert sg.stats()["factor_active"] == 1 and obj_close(sg.objective(), so.objective())
    x = sg.values()
    j = int(np.argmax(x))
    cg, co = sg.clone(), so.clone()
    cg = cg.fix_var(j, 0.5 * x[j])
    co = co.fix_var(j, 0.5 * x[j])
    assert cg.stats()["factor_active"] == 1
    assert obj_close(cg.objective(), co.objective())
    cg, _ = cg.unfix_var(j)
    co, _ = co.unfix_var(j)
    assert obj_close(cg.objective(), co.objective()) and obj_close(cg.objective(), so.objective())
    assert obj_close(sg.objective(), so.objective())                      # the original is untouched by its clone's pivots
    k = int(np.argsort(x)[-2])
    sg2 = sg.add_constraint([(j, 1.0), (k, 1.0)], lpgen.LE, 0.7 * (x[j] + x[k]))
    so2 = so.add_constraint([(j, 1.0), (k, 1.0)], lpgen.LE, 0.7 * (x[j] + x[k]))
    assert sg2.stats()["factor_active"] == 1   # (round 5: a new row is one more refactorisation, not a return to the explicit inverse)
    assert obj_close(sg2.objective(), so2.objective())


def test_cutting_plane_loop_stays_on_the_compact_factor(monkeypatch):
    """Solution::add_constraint (lib.rs:368 -> solver.rs:549-634) twelve times on a transport instance held as the compact factor:
    every cut extends the factor by a re-peel of the extended basis (the new slack is a column singleton on the new row: level 1),
    the dual re-solve runs on it, and objective, feasibility and the re-solve's pivots match the oracle after every cut; the
    explicit inverse is never built (factor_active stays 1, no mode switch beyond the first)."""
    monkeypatch.setenv("MLP_FACTOR", "1")
    lp = lpgen.gen_transport_lp(1500, 2000, 4, 31, tight=0.5)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert sg.stats()["factor_active"] == 1 and obj_close(sg.objective(), so.objective())
    switches0 = sg.stats()["factor_switches"]
    rng = np.random.default_rng(5)
    for cut in range(12):
        x = np.asarray(so.values())
        nzs = np.flatnonzero(x > 1e-9)
        pick = rng.choice(nzs, size=min(4, len(nzs)), replace=False)
        coef = rng.uniform(0.5, 1.5, size=len(pick))
        rhs = 0.8 * float(coef @ x[pick])
        terms = [(int(j), float(a)) for j, a in zip(pick, coef)]
        no, ng = len(so.trace()), len(sg.trace())
        so = so.add_constraint(terms, lpgen.LE, rhs)
        sg = sg.add_constraint(terms, lpgen.LE, rhs)
        assert sg.stats()["factor_active"] == 1, cut
        assert obj_close(sg.objective(), so.objective()), (cut, sg.objective(), so.objective())
        assert [t[:5] for t in sg.trace()[ng:]] == [t[:5] for t in so.trace()[no:]], cut
        xg = np.asarray(sg.values())
        assert float(np.dot(coef, xg[pick])) <= rhs + 1e-7
    assert sg.stats()["factor_switches"] == switches0


def test_config3_on_the_compact_factor_with_the_bump_as_a_sparse_lu(monkeypatch):
    """Config 3 (6 000 x 10 000, four entries per row) solved on the compact factor from the slack basis on, the bump the two-sided
    peel leaves (2 ... 32 columns on cycles of the basis graph) carried as a SPARSE LU WITH FILL (factor_sb.inc: rounds of
    independent pivots, threshold 0.1, lowest Markowitz count) instead of its dense inverse (MLP_FACTOR_SB_FROM lowered from 48):
    the oracle's pivots and optimum; then the same solve with the dense inverse: the same pivots again."""
    lp = lpgen.gen_mixed_lp(6000, 10000, 4, 3)
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_SB_FROM", "2")
    so, sg = _pair(lp)
    st, sb = sg.stats(), _sb(sg)
    print("pivots", st["iterations"], "refactorisations", st["factor_refactors"], "largest bump", st["factor_bump_max"], "sparse bump", sb)
    assert st["factor_active"] == 1 and st["factor_bump_max"] >= 10
    assert sb["factorisations"] >= 10 and sb["fallbacks"] == 0
    assert obj_close(sg.objective(), so.objective())
    check_feasible(lp, sg.values())  # (another optimal vertex than the oracle's: see below)
    assert sg.stats()["max_pivot_err"] < 1e-9
    # the pivot sequence: config 3 is degenerate — from pivot 663 on the dual ratio test meets floating-point ties, and which vertex
    # path a solve continues along depends on the last bits of the solves (measured: the oracle 3 954 pivots; the engine 3 955 on every
    # representation, first difference from the oracle at pivot 663 with a refactor period of 32, at 1 289 with a period of 1 or 64) —
    # so the sequence is compared with the oracle's up to the first tie, and the carriers of the bump with each other by their optimum
    to, tg = [t[:5] for t in so.trace()], [t[:5] for t in sg.trace()]
    div = next((i for i, (a, b) in enumerate(zip(tg, to)) if a != b), min(len(tg), len(to)))
    assert div >= 600, div
    assert abs(len(tg) - len(to)) <= len(to) // 50
    monkeypatch.setenv("MLP_FACTOR_SB", "0")
    sd = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert _sb(sd)["factorisations"] == 0 and sd.stats()["factor_bump_max"] >= 10
    td = [t[:5] for t in sd.trace()]
    assert next((i for i, (a, b) in enumerate(zip(td, tg)) if a != b), len(tg)) >= 600
    assert obj_close(sd.objective(), sg.objective())


def test_a_dense_bump_is_carried_by_the_dense_tail_of_the_elimination(monkeypatch):
    """Config-4 family (twelve entries per row, a nucleus that is all cycles): nothing of its bump (up to 133 columns) is sparse, and the
    elimination hands it — whole: fewer than 128 columns are left before the first round — to its DENSE TAIL (Gauss-Jordan with partial
    pivoting inside the factorisation kernel, K_t^-1 applied as a block in the solves).  Same pivots as the oracle.  (Before the dense
    tail these bumps overflowed the row slots and fell back to the dense carrier; a fill that outgrows the slots with more than 128
    columns left still does: test_staircase_family_hands_a_filling_bump_over.)"""
    monkeypatch.setenv("MLP_FACTOR", "1")
    monkeypatch.setenv("MLP_FACTOR_J", "7")
    monkeypatch.setenv("MLP_FACTOR_SB_FROM", "2")
    lp = lpgen.gen_sparse_lp(400, 300, 12, 7)
    so, sg = _pair(lp)
    st, sb = sg.stats(), _sb(sg)
    print("pivots", st["iterations"], "largest bump", st["factor_bump_max"], "sparse bump", sb)
    assert st["factor_active"] == 1 and st["factor_bump_max"] >= 20
    assert sb["factorisations"] >= 10 and sb["tail"] >= 20
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())


@pytest.mark.parametrize("knob", ["MLP_FACTOR_FUSE", "MLP_FACTOR_RHO_PART"])
def test_fused_launches_and_left_behind_partial_sums_change_no_bit(knob):
    """The dual iteration on the compact factor with its three single-purpose launches riding inside the solves (U_j stored unscaled with
    its factor beside it) and with the BTRAN leaving the partial sums of V_j . rho for the FTRAN of tau — against the forms they replace
    (knob = 0, read once per process: each form runs in a process of its own): the same pivots, and x, the objective and the dual
    steepest-edge weights BIT for bit.  Transport instance (several levels, 32 pending terms) and config 3 (a bump as a sparse LU)."""
    import subprocess, sys, os, json
    code = r'''
import sys, json, hashlib
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
out = {}
for name, lp in (("transport", lpgen.gen_transport_lp(800, 1000, 4, 11, tight=0.5)), ("config3", lpgen.gen_mixed_lp(6000, 10000, 4, 3))):
    s = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert s.stats()["factor_active"] == 1
    h = hashlib.sha1()
    h.update(np.asarray(s.values(), dtype=np.float64).tobytes())
    h.update(np.asarray(s.state("dual_edge_sq_norms"), dtype=np.float64).tobytes())
    h.update(np.float64(s.objective()).tobytes())
    h.update(repr([t[:5] for t in s.trace()]).encode())
    out[name] = [h.hexdigest(), int(s.stats()["iterations"])]
print("RESULT " + json.dumps(out))
'''
    res = []
    for val in ("1", "0"):
        env = dict(os.environ, MLP_FACTOR="1", MLP_FACTOR_SB_FROM="2")
        env[knob] = val
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        assert r.returncode == 0 and line, (r.returncode, r.stderr[-800:])
        res.append(json.loads(line[-1][7:]))
    print(knob, res)
    assert res[0] == res[1]
    assert res[0]["transport"][1] > 100 and res[0]["config3"][1] > 3000


def test_staircase_family_hands_a_filling_bump_over(monkeypatch):
    """Multi-period production / inventory model (lpgen.gen_staircase_lp, 5 000 rows): with the compact factor forced its bump grows to a
    quarter of the rows and the elimination of it fills in a dense tail (rows of 60-80 entries: tools/experiments/bump_lu.py staircase).
    The sparse LU carries the bump while its rows fit their slots (dozens of factorisations), then raises its flag; beyond 1 024
    columns there is no dense carrier either, so the solve goes back to the explicit nucleus inverse — and reaches the oracle's optimum.
    (On the default path this family is where the explicit inverse shines: 1.0 s against 7.8 s for the oracle, whose LU fills.)"""
    monkeypatch.setenv("MLP_FACTOR", "1")
    lp = lpgen.gen_staircase_lp(40, 100, 150, 25)
    so = lpgen.build_problem(O.Problem, lp).solve()
    sg = lpgen.build_problem(M.Problem, lp).solve()
    st, sb = sg.stats(), _sb(sg)
    print("pivots", st["iterations"], "largest bump", st["factor_bump_max"], "switches", st["factor_switches"], "sparse bump", sb)
    assert sb["factorisations"] >= 10 and sb["fallbacks"] >= 1
    assert st["factor_switches"] >= 2 and st["factor_active"] == 0 and st["factor_bump_max"] > 1024
    assert obj_close(sg.objective(), so.objective())
    check_feasible(lp, sg.values())
