"""Per-stage differential tests (pyramid level 2): the engine is stepped stage by stage through the
C ABI (mlp_engine_open / mlp_engine_stage) and after every stage the vector that stage produced is read back and
compared with the oracle's vector of the same pivot:

  FTRAN  -> col_coeffs            alpha_q = B^-1 a_q                 solver.rs:671-677
  BTRAN  -> inv_basis_row_coeffs  rho_r   = B^-T e_r                 solver.rs:680-683
  ROW    -> row_coeffs            alpha_r = rho_r^T N                solver.rs:685-692
  BASIS  -> tau = B^-1 rho_r (solver.rs:1157), v = B^-T alpha_q (solver.rs:1114)
  APPLY  -> sq_norms_update_helper (solver.rs:1126-1132, on the touched columns), then x_B, d, gamma, beta
            (solver.rs:1049-1080, 1140-1150, 1164-1173)

Tolerance: the two sides sum the same terms in a different order, |delta| <= 1e-11 * max(1, |x|_inf)."""
import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN

pytestmark = pytest.mark.gpu
A = M.api
RTOL = 1e-11


def _close(name, it, got, want, mask=None):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, it, got.shape, want.shape)
    if mask is not None:
        got, want = got[mask], want[mask]
    if got.size == 0:
        return 0.0
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= RTOL * scale, f"{name} differs at iteration {it}: max|delta| = {err:.3e} (scale {scale:.3e})"
    return err / scale


def _step_and_compare(lp, max_iters):
    sg = lpgen.build_problem(M.Problem, lp).solve(budget=0, trace=True)
    so = lpgen.build_problem(O.Problem, lp).solve(budget=0, trace=True)
    so.set_capture(True)
    worst, it, phases = {}, 0, set()
    while it < max_iters:
        st, info = sg.engine_open()
        if st in (A.ITER_OPTIMAL, A.ITER_INFEASIBLE, A.ITER_UNBOUNDED):
            break
        if st == A.ITER_FEASIBLE:
            continue
        assert st == A.ITER_PIVOT
        while st in (A.ITER_PIVOT, A.ITER_FLIP) and it < max_iters:
            got = {}
            phase = info["phase"]
            phases.add(phase)
            while True:
                stage = info["next_stage"]
                st, info = sg.engine_stage(stage)
                if stage == A.STAGE_FTRAN:
                    got["col_coeffs"] = sg.state("col_coeffs")
                elif stage == A.STAGE_BTRAN:
                    got["inv_basis_row_coeffs"] = sg.state("inv_basis_row_coeffs")
                elif stage == A.STAGE_ROW:
                    got["row_coeffs"] = sg.state("row_coeffs")
                elif stage == A.STAGE_BASIS:
                    got["tau"] = sg.state("tau")
                    got["v"] = sg.state("v")
                if stage == A.STAGE_APPLY or st not in (A.ITER_PIVOT, A.ITER_FLIP):
                    break
            if stage != A.STAGE_APPLY:
                break
            # the oracle takes the same pivot (a phase switch may spend one budget unit on its decision record)
            n0 = len(so.trace())
            for _ in range(3):
                so.continue_solve(1)
                if len(so.trace()) > n0:
                    break
            tg, to = sg.trace()[-1], so.trace()[-1]
            assert tg[:5] == to[:5], (it, tg, to)
            is_flip = tg[2] < 0
            pse = bool(so.state("flags")[2]) or phase == 0 and bool(sg.state("flags")[2])
            if not is_flip:
                for name in ("col_coeffs", "row_coeffs"):
                    worst[name] = max(worst.get(name, 0.0), _close(name, it, got[name], so.state(name)))
                worst["rho"] = max(worst.get("rho", 0.0), _close("inv_basis_row_coeffs", it, got["inv_basis_row_coeffs"], so.state("inv_basis_row_coeffs")))
                worst["tau"] = max(worst.get("tau", 0.0), _close("tau", it, got["tau"], so.state("dbg_tau"), mask=np.asarray(got["col_coeffs"]) != 0.0))
                if pse and len(so.state("dbg_v")):
                    worst["v"] = max(worst.get("v", 0.0), _close("v", it, got["v"], so.state("dbg_v")))
                    touched = np.asarray(so.state("row_coeffs")) != 0.0
                    worst["helper"] = max(worst.get("helper", 0.0), _close("sq_norms_update_helper", it, sg.state("sq_norms_update_helper"),
                                                                           so.state("sq_norms_update_helper"), mask=touched))
            for name in ("basic_var_vals", "nb_var_obj_coeffs", "nb_var_vals", "dual_edge_sq_norms", "primal_edge_sq_norms"):
                if name == "primal_edge_sq_norms" and not pse:
                    continue
                worst[name] = max(worst.get(name, 0.0), _close(name, it, sg.state(name), so.state(name)))
            assert (np.asarray(sg.state("basic_vars")) == np.asarray(so.state("basic_vars"))).all()
            assert (np.asarray(sg.state("nb_vars")) == np.asarray(so.state("nb_vars"))).all()
            it += 1
    return it, worst, phases


@pytest.mark.parametrize("fam,args,iters", [("sparse", (300, 260, 10, 41), 150), ("dense", (60, 50, 42), 40),
                                            ("cover", (260, 300, 10, 43), 120), ("twophase", (300, 300, 12, 44), 160)], ids=str)
def test_every_stage_matches_the_oracle(fam, args, iters):
    lp = GEN[fam](*args)
    n, worst, phases = _step_and_compare(lp, iters)
    assert n >= min(iters, 10), n
    if fam == "twophase":
        assert phases == {0, 1}       # dual loop on the artificial objective, then the primal loop
    if fam == "cover":
        assert phases == {1}
    print(fam, n, "iterations; worst relative differences:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_every_stage_matches_with_the_large_nucleus_machinery(monkeypatch):
    """Same comparison with the delayed-update mode (J = 3), the 16-row tiles, the padded pitch, the blocked F push
    and the banded sweep forced on: the stages are then served by the other kernel variants."""
    for k, v in dict(MLP_LOWRANK="3", MLP_BIGTILE="1", MLP_LDPAD="16", MLP_BANDED="1", MLP_STR_K="0").items():
        monkeypatch.setenv(k, v)
    lp = GEN["sparse"](300, 260, 10, 41)
    n, worst, _ = _step_and_compare(lp, 120)
    assert n >= 100
