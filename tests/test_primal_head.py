"""GPU: the small-nucleus primal head (csrc/primal_head.inc, k_primal_head) — FTRAN of the entering column (solver.rs:671-677), both
Harris passes (741-853), BTRAN of the leaving row (680-683), v = B^-T alpha_q with the eager eta update of the nucleus inverse
(1114, 1274-1284), the touched-column list of the tableau row (685-692) and the partition change as ONE launch of ONE workgroup,
used by the lazy primal steepest-edge iteration while the nucleus holds a few dozen columns and the F products are pushed (large
models; forced here on small ones with MLP_DETERMINISTIC=0).

Gates: (a) a solve with the head and a solve on the launches it replaces (MLP_PRIMAL_HEAD=0) take the SAME pivots, and both take the
oracle's; objective and values agree to the parity tolerances (the pushed F products are float atomics: bits are not comparable);
(b) the nucleus outgrowing the kernel's slots mid-solve moves the iteration to the multi-launch forms and back never;
(c) a tiny slot bound (MLP_PRIMAL_HEAD_K) makes batches end early and alternate with the other forms from the start;
(d) a config-4-shaped instance (20 000 x 20 000, 50 per row: the grid forms of everything else) against the oracle.
state("primal_head_launches") counts the iterations the kernel carried through: "the path was taken" is checked, not assumed."""
import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, X_ATOL, obj_close

pytestmark = pytest.mark.gpu

CASES = [("sparse", (700, 600, 12, 6)), ("sparse", (1000, 1000, 100, 2)), ("sparse", (3000, 2600, 12, 9)), ("twophase", (600, 600, 12, 44)),
         ("dense", (150, 100, 3))]


def _solve(lp, **kw):
    s = lpgen.build_problem(M.Problem, lp).solve(trace=True, **kw)
    return s, int(s.state("primal_head_launches")[0])


FORMS = {"head + k_row_pull + update": {"MLP_PULL_INSIDE": "0"},
         "head + update that pulls the tableau row": {"MLP_PULL_INSIDE": "1", "MLP_HEAD_APPLY": "0"},
         "head that applies the basic side + update of the non-basic side": {"MLP_PULL_INSIDE": "1", "MLP_HEAD_APPLY": "1"}}


@pytest.mark.parametrize("form", list(FORMS), ids=list(FORMS))
@pytest.mark.parametrize("fam,args", CASES, ids=str)
def test_head_takes_the_pivots_of_the_launches_it_replaces_and_of_the_oracle(monkeypatch, fam, args, form):
    monkeypatch.setenv("MLP_HYPER", "0")
    monkeypatch.setenv("MLP_DETERMINISTIC", "0")   # pushed F products, as on models beyond 2^21 non-zeros
    for k_, v_ in FORMS[form].items():
        monkeypatch.setenv(k_, v_)
    lp = GEN[fam](*args)
    monkeypatch.setenv("MLP_PRIMAL_HEAD", "1")
    s1, n1 = _solve(lp)
    monkeypatch.setenv("MLP_PRIMAL_HEAD", "0")
    s0, n0 = _solve(lp)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    # (the two-phase instance leaves its dual phase with a nucleus beyond the head's slots: it covers "the head is never entered")
    assert (n1 > 0 or fam == "twophase") and n0 == 0, (n1, n0)
    assert [t[:5] for t in s1.trace()] == [t[:5] for t in s0.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(s1.objective(), so.objective()) and obj_close(s1.objective(), s0.objective())
    assert np.abs(np.asarray(s1.values()) - np.asarray(so.values())).max() <= X_ATOL
    assert s1.reinvert() < 1e-8
    print(f"{fam}{args}: {len(s1.trace())} pivots, {n1} through k_primal_head, nucleus {s1.stats()['nucleus_size']}")


def test_nucleus_outgrowing_the_head_moves_on_mid_solve(monkeypatch):
    monkeypatch.setenv("MLP_HYPER", "0")
    monkeypatch.setenv("MLP_DETERMINISTIC", "0")
    lp = lpgen.gen_sparse_lp(2500, 2000, 10, 4)
    sg, n = _solve(lp)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    st = sg.stats()
    assert st["nucleus_size"] > 62 and 0 < n < st["iterations"], (st["nucleus_size"], n, st["iterations"])
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert sg.reinvert() < 1e-8


@pytest.mark.parametrize("kmax", [3, 9])
def test_a_tiny_slot_bound_alternates_with_the_other_forms(monkeypatch, kmax):
    monkeypatch.setenv("MLP_HYPER", "0")
    monkeypatch.setenv("MLP_DETERMINISTIC", "0")
    monkeypatch.setenv("MLP_PRIMAL_HEAD_K", str(kmax))
    lp = lpgen.gen_sparse_lp(1200, 1000, 12, 21)
    sg, n = _solve(lp)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    assert 0 < n < sg.stats()["iterations"]
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())


def test_config4_shape_first_pivots_against_the_oracle():
    """20 000 x 20 000, 50 per row (10^6 non-zeros... below 2^21: pushed products forced; above 16 384 rows: every other kernel of the
    iteration runs in the form config 4 uses), 400 pivots from the slack basis: the head serves the first ~55, the rest follow."""
    import os
    os.environ["MLP_DETERMINISTIC"] = "0"
    try:
        lp = lpgen.gen_sparse_lp(20000, 20000, 50, 4)
        sg = lpgen.build_problem(M.Problem, lp).solve(budget=400, trace=True)
        n = int(sg.state("primal_head_launches")[0])
        so = lpgen.build_problem(O.Problem, lp).solve(budget=400, trace=True)
    finally:
        del os.environ["MLP_DETERMINISTIC"]
    assert n >= 20, n
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())


@pytest.mark.parametrize("form", list(FORMS), ids=list(FORMS))
def test_bound_flips_and_boxed_columns_through_the_head(monkeypatch, form):
    """Finite upper bounds on a third of the columns (tight enough that the Harris test often ends in a BOUND FLIP, solver.rs:846-851,
    1031-1042) and a few columns with free lower bounds: the head's no-winner branch (flip applied by the head itself in the applying
    form: x_B on supp(alpha_q), x_N and the flags of the entering column), entering columns coming from their upper bound
    (entering_diff_sign, solver.rs:741-748) and the leaving variable's at-max flag — pivot for pivot with the oracle."""
    monkeypatch.setenv("MLP_HYPER", "0")
    monkeypatch.setenv("MLP_DETERMINISTIC", "0")
    for k_, v_ in FORMS[form].items():
        monkeypatch.setenv(k_, v_)
    lp = lpgen.gen_sparse_lp(900, 800, 12, 17)
    rng = np.random.default_rng(3)
    hi = lp["hi"].copy()
    boxed = rng.random(len(hi)) < 0.35
    hi[boxed] = rng.uniform(0.02, 0.6, size=int(boxed.sum()))
    lp = dict(lp, hi=hi, name=lp["name"] + "_boxed")
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg, n = _solve(lp)
    st = sg.stats()
    assert n > 0 and st["bound_flips"] > 0, (n, st["bound_flips"])
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(np.asarray(sg.values()) - np.asarray(so.values())).max() <= X_ATOL
    print(f"{form}: {len(sg.trace())} pivots, {st['bound_flips']} bound flips, {n} through k_primal_head")
