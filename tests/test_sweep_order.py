"""GPU: locality order of the banded tableau-row sweep and its packed non-basic copy (HISTORY.md §7.0).  On config 4
they switch on after 4 096 pivots and are rebuilt every 2 048; here they are forced on from the first pivot and rebuilt
every few pivots on small instances (MLP_BANDED=1, MLP_ORDER_FROM=0, MLP_ORDER_EVERY=7): the pass visits the positions in
a different order and reads a different copy of A, but every tableau row, hence every pivot, must stay the oracle's —
through primal and dual phases, bound flips, cuts appended on the device (which rebuild both copies) and clones."""
import os

import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, X_ATOL, obj_close

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["1", "0"], ids=["packed", "indirect"])
def order_env(request):
    # (MLP_STR_K=0: the sparse tableau row would otherwise serve these small nuclei and the sweep under test never run)
    env = {"MLP_BANDED": "1", "MLP_ORDER_FROM": "0", "MLP_ORDER_EVERY": "7", "MLP_SWEEP_PACKED": request.param, "MLP_STR_K": "0"}
    os.environ.update(env)
    yield request.param
    for k in env:
        os.environ.pop(k, None)


CASES = [("sparse", dict(m=300, n=260, k=10, seed=11)), ("cover", dict(m=250, n=250, k=9, seed=12)),
         ("twophase", dict(m=300, n=300, k=12, seed=13)), ("mixed", dict(m=200, n=260, k=5, seed=14)),
         ("dense", dict(m=60, n=90, seed=15))]


@pytest.mark.parametrize("fam,kw", CASES, ids=lambda v: str(v))
def test_sweep_order_and_packed_copy_keep_the_pivots(order_env, fam, kw):
    lp = GEN[fam](**kw)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert sg.stats()["banded_sweep"] == 1
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(so.values() - sg.values()).max() <= X_ATOL


def test_sweep_order_through_cuts_and_clones(order_env):
    lp = lpgen.gen_sparse_lp(220, 180, 9, 21)
    so, sg = lpgen.build_problem(O.Problem, lp).solve(), lpgen.build_problem(M.Problem, lp).solve()
    x = so.values()
    for step in range(8):
        vars_ = [(5 * step + j) % 180 for j in range(4)]
        lhs = float(sum(x[v] for v in vars_))
        expr = [(v, 1.0) for v in vars_]
        so, sg = so.add_constraint(expr, O.LE, 0.9 * lhs + 0.01), sg.add_constraint(expr, M.LE, 0.9 * lhs + 0.01)
        assert obj_close(sg.objective(), so.objective())
        x = so.values()
    c = sg.clone()
    fx = int(np.argmax(x))
    assert obj_close(c.fix_var(fx, 0.0).objective(), so.fix_var(fx, 0.0).objective())
