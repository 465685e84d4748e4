"""CPU: the oracle against the committed HiGHS golden objectives, the generators' determinism,
and oracle-internal size-independent properties."""
import numpy as np
import pytest

from minilp_amd import lpgen
from oracle import minilp_oracle as mo
from tests.common import GEN, HIGHS_RTOL, check_feasible, highs_cases, obj_close, objective_of

SMALL = [c for c in highs_cases() if c["args"]["m"] <= 1000 and not (c["family"] == "dense" and c["args"]["m"] >= 1000)]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: c["name"])
def test_oracle_matches_highs_objective(case):
    lp = GEN[case["family"]](**case["args"])
    assert lp["name"] == case["name"]
    s = lpgen.build_problem(mo.Problem, lp).solve()
    assert obj_close(s.objective(), case["objective"], HIGHS_RTOL)
    x = s.values()
    check_feasible(lp, x)
    assert obj_close(objective_of(lp, x), s.objective(), 1e-9)


def test_oracle_config2_dense_1000():  # BASELINE config 2 through the oracle (a few seconds)
    case = [c for c in highs_cases() if c["name"] == "dense_1000x1000_s2"][0]
    lp = lpgen.gen_dense_lp(1000, 1000, 2)
    s = lpgen.build_problem(mo.Problem, lp).solve()
    assert obj_close(s.objective(), case["objective"], HIGHS_RTOL)
    check_feasible(lp, s.values())


def test_generators_are_deterministic():
    a = lpgen.gen_sparse_lp(100, 120, 7, 9)
    b = lpgen.gen_sparse_lp(100, 120, 7, 9)
    for key in ("indices", "data", "rhs", "obj"):
        assert (a[key] == b[key]).all()
    assert (np.diff(a["indices"].reshape(100, 7), axis=1) > 0).all()  # sorted, distinct columns per row
    # SplitMix64 known answers (seed 0): first outputs of the reference implementation
    assert [int(v) for v in lpgen.splitmix64(0, 3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_budget_protocol_resumes_identically():
    """Fixed-pivot-budget runs must not change the pivot sequence."""
    lp = lpgen.gen_sparse_lp(150, 120, 8, 11)
    full = lpgen.build_problem(mo.Problem, lp).solve(trace=True)
    part = lpgen.build_problem(mo.Problem, lp).solve(budget=10, trace=True)
    assert part.budget_exhausted
    while part.budget_exhausted:
        part.continue_solve(7)
    assert [t[:5] for t in part.trace()] == [t[:5] for t in full.trace()]
    assert part.objective() == full.objective()
