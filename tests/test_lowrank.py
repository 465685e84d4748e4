"""GPU: the delayed-update mode of the nucleus inverse (W = W0 + sum_j U_j V_j^T, folded every J
pivots; HISTORY.md §2.1).  It is switched on automatically from capacity 8192; here it is forced on
small instances (MLP_LOWRANK=J is read when a Solution is created) and must reproduce the oracle's
pivot sequence exactly like the in-place mode, through every partition case."""
import os

import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, X_ATOL, check_feasible, obj_close

pytestmark = pytest.mark.gpu


# (J, large-nucleus tiling forced): J = 0 is the in-place mode; the second flag selects the 64-row
# non-temporal tiles and a padded row pitch of W that are otherwise used from capacity 8192 on
@pytest.fixture(params=[(1, 0), (3, 0), (16, 0), (3, 1), (16, 1), (0, 1)], ids=lambda p: f"J{p[0]}-big{p[1]}")
def lowrank(request):
    j, big = request.param
    os.environ["MLP_LOWRANK"] = str(j)
    os.environ["MLP_STR_K"] = "0"   # (the sweeps of the large-nucleus regime are what runs there, not the sparse tableau row)
    if big:
        os.environ["MLP_BIGTILE"] = "1"
        os.environ["MLP_LDPAD"] = "16"
        os.environ["MLP_BANDED"] = "1"   # banded tableau-row sweep (otherwise from 32 768 rows on)
    yield j
    for k in ("MLP_LOWRANK", "MLP_BIGTILE", "MLP_LDPAD", "MLP_BANDED", "MLP_STR_K"):
        os.environ.pop(k, None)


CASES = [("sparse", dict(m=200, n=200, k=10, seed=4)), ("sparse", dict(m=700, n=600, k=12, seed=6)),
         ("dense", dict(m=150, n=100, seed=3)), ("dense", dict(m=64, n=257, seed=4))]


@pytest.mark.parametrize("fam,kw", CASES, ids=lambda v: str(v))
def test_lowrank_mode_matches_oracle_pivot_for_pivot(lowrank, fam, kw):
    lp = GEN[fam](**kw)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(so.values() - sg.values()).max() <= X_ATOL
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert sum(sg.stats()["kase"]) == sg.stats()["basis_changes"]
    assert sg.reinvert() < 1e-8  # fold + compare with a from-scratch inverse


@pytest.mark.parametrize("fam,kw", CASES[:2] + [("sparse", dict(m=1500, n=1400, k=12, seed=9))], ids=lambda v: str(v))
def test_pulled_F_product_matches_oracle_pivot_for_pivot(fam, kw, monkeypatch):
    """Round 6 (csrc/fpull.inc): the F product of the primal FTRAN pulled per singleton row from the packed copy of the nucleus columns,
    entering columns appended to it pivot by pivot, Harris pass 1 in the same launch — the path of the large-nucleus regime (from
    capacity 8 192 on), forced here on small instances (delayed-update mode, blocked-push geometry, the grid forms of the ratio test) so
    that the ORACLE is the other side: the whole solve pivot for pivot, through every partition case.  The solve starts from the slack
    basis: the packed copy is built empty and every nucleus column of the solve gets into it through the append."""
    for k, v in dict(MLP_LOWRANK="3", MLP_STR_K="0", MLP_BIGTILE="1", MLP_LDPAD="16", MLP_BANDED="1", MLP_RATIO_ONE="0", MLP_HYPER="0").items():
        monkeypatch.setenv(k, v)
    lp = GEN[fam](**kw)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    fp = sg.state("fpull")
    assert int(fp[0]) == 1 and int(fp[3]) == 1 and int(fp[1]) >= 1, fp      # the pulled form was the one in use
    assert obj_close(sg.objective(), so.objective())
    assert np.abs(so.values() - sg.values()).max() <= X_ATOL
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert sum(sg.stats()["kase"]) == sg.stats()["basis_changes"]
    assert sg.reinvert() < 1e-8


def test_lowrank_mode_dual_and_warm_start(lowrank):
    lp = lpgen.gen_mixed_lp(300, 400, 8, 4)   # dual simplex, every partition case incl. singleton swaps
    so = lpgen.build_problem(O.Problem, lp).solve()
    sg = lpgen.build_problem(M.Problem, lp).solve()
    assert obj_close(sg.objective(), so.objective())
    check_feasible(lp, sg.values())
    st = sg.stats()
    assert st["kase"][2] + st["kase"][3] > 0
    lp2 = lpgen.gen_sparse_lp(120, 90, 9, 21)
    so, sg = lpgen.build_problem(O.Problem, lp2).solve(), lpgen.build_problem(M.Problem, lp2).solve()
    x = so.values()
    for step in range(6):
        vars_ = [(3 * step + j) % 90 for j in range(4)]
        lhs = float(sum(x[v] for v in vars_))
        expr = [(v, 1.0) for v in vars_]
        so, sg = so.add_constraint(expr, O.LE, 0.9 * lhs + 0.01), sg.add_constraint(expr, M.LE, 0.9 * lhs + 0.01)
        assert obj_close(sg.objective(), so.objective())
        x = so.values()
    c = sg.clone()
    assert obj_close(c.objective(), sg.objective())


def test_lowrank_mode_config4_budget():
    os.environ["MLP_LOWRANK"] = "8"
    try:
        lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
        sg = lpgen.build_problem(M.Problem, lp).solve(budget=300, trace=True)
        so = lpgen.build_problem(O.Problem, lp).solve(budget=300, trace=True)
        assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
        assert obj_close(sg.objective(), so.objective())
        sg.continue_solve(1500)
        assert sg.reinvert() < 1e-6
    finally:
        os.environ.pop("MLP_LOWRANK", None)


def test_engine_level_stepping_with_the_large_model_machinery(lowrank):
    """The host-paced stage API over the banded sweep (separate combine launch), the blocked F push and the
    delayed-update mode: same optimum and, on the sparse family, the same pivots as the oracle."""
    from tests.test_hip_parity import _drive_by_stages
    lp = lpgen.gen_sparse_lp(200, 200, 10, 4)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    sg = lpgen.build_problem(M.Problem, lp).solve(budget=0, trace=True)
    st, pivots = _drive_by_stages(sg)
    assert st == M.api.ITER_OPTIMAL and pivots == len(so.trace())
    assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
    assert obj_close(sg.objective(), so.objective())


@pytest.mark.parametrize("balanced", ["0", "1", "5", "4096"])
def test_stream_strip_geometry_does_not_change_the_pivots(balanced):
    """k_stream_w: fixed 128-row strips (0), one equal tile per co-resident block (1, the default), and forced block
    counts that make a block loop over several tiles (5) or leave most blocks without one (4096)."""
    env = {"MLP_LOWRANK": "16", "MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_BANDED": "1", "MLP_STREAM_BALANCED": balanced}
    os.environ.update(env)
    try:
        lp = lpgen.gen_sparse_lp(1500, 1500, 20, 2)   # a nucleus of several hundred slots
        so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
        sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
        assert sg.stats()["nucleus_size"] > 0
        assert [t[:5] for t in sg.trace()] == [t[:5] for t in so.trace()]
        assert obj_close(sg.objective(), so.objective())
        assert sg.reinvert() < 1e-8
    finally:
        for k in env:
            os.environ.pop(k, None)


def test_ftran_head_inside_the_gather_is_bit_identical_to_the_one_wave_launch(monkeypatch):
    """Round 5: in the delayed-update mode the FTRAN head (entering column's scalars, fold decision, singleton entries, list, c_j = V[j] . list)
    runs inside every block of the gather (k_ftran_gather_lrh) instead of as a one-wave launch in front of it (k_ftran_prep +
    k_ftran_gather<4>).  Same list order, same sums: with the deterministic blocked push the two solves agree bit for bit — trace,
    objective, values — and take the oracle's pivots."""
    for k_, v_ in dict(MLP_LOWRANK="3", MLP_BIGTILE="1", MLP_LDPAD="16", MLP_BANDED="1", MLP_STR_K="0", MLP_PB_DET="1", MLP_HYPER="0").items():
        monkeypatch.setenv(k_, v_)
    lp = lpgen.gen_sparse_lp(900, 800, 12, 5)
    runs = []
    for on in ("1", "0"):
        monkeypatch.setenv("MLP_LR_HEAD_FUSION", on)
        s = lpgen.build_problem(M.Problem, lp).solve(trace=True)
        runs.append((s.trace(), s.objective(), s.values().tobytes()))
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    assert [t[:5] for t in runs[0][0]] == [t[:5] for t in so.trace()]
    assert obj_close(runs[0][1], so.objective())
