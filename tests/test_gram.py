"""GPU: Gram mode of the primal steepest-edge solve (opt-in, MLP_GRAM=1).  The v = B^-T alpha_q of
solver.rs:1114 is read from a resident M = [(B B^T)^-1] on the nucleus rows plus a pass over the few rows of the
nucleus inverse that F^T D^-2 a_S touches, instead of a pass over the whole inverse.  Here the large-nucleus machinery
is forced on small instances (as in test_lowrank.py), where M is accurate, and the path must reproduce the oracle's
pivot sequence exactly like the streaming pass does — through every partition case, folds of M every 16 pivots,
rebuilds of M (capacity growth, re-inversion, a checkpoint, a forced rebuild per batch), the back-off and phase changes."""
import os

import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, X_ATOL, obj_close

pytestmark = pytest.mark.gpu

FORCE = {"MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_BANDED": "1"}


@pytest.fixture(params=[3, 16, 32], ids=lambda j: f"J{j}")
def gram_env(request):
    os.environ.update(FORCE)
    os.environ["MLP_GRAM"] = "1"
    os.environ["MLP_LOWRANK"] = str(request.param)
    yield request.param
    for k in list(FORCE) + ["MLP_LOWRANK", "MLP_GRAM", "MLP_GRAM_TOL", "MLP_GRAM_MIN_GAP"]:
        os.environ.pop(k, None)


def _key(t):
    return t[:5]


CASES = [("sparse", dict(m=200, n=200, k=10, seed=4)), ("sparse", dict(m=700, n=600, k=12, seed=6)),
         ("sparse", dict(m=1500, n=1500, k=20, seed=2)), ("dense", dict(m=150, n=100, seed=3))]


@pytest.mark.parametrize("fam,kw", CASES, ids=lambda v: str(v))
def test_gram_path_matches_oracle_pivot_for_pivot(gram_env, fam, kw):
    lp = GEN[fam](**kw)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    st = sg.stats()
    assert st["gram_pivots"] == st["basis_changes"] > 0          # every basis change of this primal solve took the path
    assert st["gram_err"] < 1e-7                                   # a_q.v against ||alpha_q||^2
    assert [_key(t) for t in sg.trace()] == [_key(t) for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(so.values() - sg.values()).max() <= X_ATOL
    assert sum(st["kase"][1:4]) > 0                                # grow / shrink / column swap all carry M along
    # and the streaming pass it replaces gives the same sequence
    os.environ["MLP_GRAM"] = "0"
    s0 = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    os.environ["MLP_GRAM"] = "1"
    assert s0.stats()["gram_pivots"] == 0
    assert [_key(t) for t in s0.trace()] == [_key(t) for t in sg.trace()]


def test_gram_rebuild_every_batch_and_after_reinversion(gram_env):
    lp = lpgen.gen_sparse_lp(700, 600, 12, 6)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    os.environ["MLP_GRAM_TOL"] = "0"                               # every pivot counts as poor: M is rebuilt after every batch
    os.environ["MLP_GRAM_MIN_GAP"] = "0"                           # ... and the mode never backs off to the streaming pass
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert sg.stats()["gram_rebuilds"] > 10
    assert [_key(t) for t in sg.trace()] == [_key(t) for t in so.trace()]
    os.environ.pop("MLP_GRAM_MIN_GAP")
    # default policy: a rebuild that does not last 4096 pivots makes the mode back off to the streaming pass for a while
    sb = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    st = sb.stats()
    assert st["gram_backoffs"] >= 1 and st["gram_pivots"] < st["basis_changes"]
    assert [_key(t) for t in sb.trace()] == [_key(t) for t in so.trace()]
    os.environ.pop("MLP_GRAM_TOL")
    prob = lpgen.build_problem(M.Problem, lp)
    s = prob.solve(budget=200, trace=True)
    r0 = s.stats()["gram_rebuilds"]
    assert s.reinvert() < 1e-8                                     # fresh nucleus inverse: M no longer trusted
    s.continue_solve(-1)
    assert s.stats()["gram_rebuilds"] > r0
    assert [_key(t) for t in s.trace()] == [_key(t) for t in so.trace()]
    # a checkpoint carries no M: the continued run rebuilds it and follows the same sequence
    s = prob.solve(budget=300, trace=True)
    cut = len(s.trace())
    t = prob.solve_from_basis(s.save_basis(2), trace=True)
    assert t.stats()["gram_rebuilds"] >= 1 and t.stats()["gram_pivots"] > 0
    assert [_key(x) for x in t.trace()] == [_key(x) for x in so.trace()[cut:]]


def test_gram_path_after_a_dual_phase(gram_env):
    lp = lpgen.gen_twophase_lp(400, 400, 14, 35)                   # dual loop first (no M), then the primal loop
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    st = sg.stats()
    assert st["dual_iters"] > 0 and 0 < st["gram_pivots"] <= st["primal_iters"]
    assert [_key(t) for t in sg.trace()] == [_key(t) for t in so.trace()]
    # warm start: the dual re-solve after a cut drops M, a clone starts without one
    x = sg.values()
    top = np.argsort(-x)[:3]
    cut = sg.add_constraint([(int(v), 1.0) for v in top], M.LE, 0.9 * float(x[top].sum()))
    ref = so.add_constraint([(int(v), 1.0) for v in top], O.LE, 0.9 * float(x[top].sum()))
    assert obj_close(cut.objective(), ref.objective())
    assert obj_close(cut.clone().objective(), ref.objective())
