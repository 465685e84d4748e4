"""Generates tests/golden/highs_objectives.json in the BUILD container (scipy 1.15.3 bundles HiGHS).
HiGHS is not the reference: it pins the OPTIMUM of the synthetic instances (objective, and x where
the optimum is unique), not minilp's pivot sequence.  Run:  python tests/golden/make_highs_fixtures.py
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
from scipy.optimize import linprog

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from minilp_amd import lpgen  # noqa: E402

CASES = [
    ("sparse", dict(m=50, n=40, k=8, seed=3)),
    ("sparse", dict(m=200, n=200, k=10, seed=4)),
    ("sparse", dict(m=300, n=500, k=20, seed=5)),
    ("sparse", dict(m=2000, n=2000, k=20, seed=4)),
    ("dense", dict(m=60, n=60, seed=2)),
    ("dense", dict(m=150, n=100, seed=3)),
    ("dense", dict(m=1000, n=1000, seed=2)),   # BASELINE config 2
    ("mixed", dict(m=30, n=40, k=5, seed=2)),
    ("mixed", dict(m=100, n=150, k=6, seed=3)),
    ("mixed", dict(m=300, n=400, k=8, seed=4)),
    ("mixed", dict(m=1000, n=1500, k=6, seed=5)),
    ("mixed", dict(m=6000, n=10000, k=4, seed=3)),   # BASELINE config 3 stand-in: ~10k vars, sparse, via MPS
    ("cover", dict(m=300, n=350, k=8, seed=4)),      # dual-only solves (Min c'x, Ax >= b)
    ("cover", dict(m=1200, n=1000, k=10, seed=6)),
    ("twophase", dict(m=300, n=350, k=8, seed=6)),   # dual loop on the artificial objective, then primal
    ("twophase", dict(m=1200, n=1000, k=10, seed=6)),
]
GEN = {"sparse": lpgen.gen_sparse_lp, "dense": lpgen.gen_dense_lp, "mixed": lpgen.gen_mixed_lp, "cover": lpgen.gen_cover_lp, "twophase": lpgen.gen_twophase_lp}


def solve_highs(lp):
    A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(lp["m"], lp["n"]))
    c = lp["obj"] if lp["direction"] == lpgen.MINIMIZE else -lp["obj"]
    ops = lp["ops"]
    le, ge, eq = ops == lpgen.LE, ops == lpgen.GE, ops == lpgen.EQ
    A_ub = sp.vstack([A[le], -A[ge]]) if (le.any() or ge.any()) else None
    b_ub = np.concatenate([lp["rhs"][le], -lp["rhs"][ge]]) if A_ub is not None else None
    A_eq = A[eq] if eq.any() else None
    b_eq = lp["rhs"][eq] if eq.any() else None
    bounds = [(None if lo == -np.inf else lo, None if hi == np.inf else hi) for lo, hi in zip(lp["lo"], lp["hi"])]
    res = linprog(c, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds, method="highs-ds",
                  options=dict(primal_feasibility_tolerance=1e-9, dual_feasibility_tolerance=1e-9))
    assert res.status == 0, res.message
    obj = res.fun if lp["direction"] == lpgen.MINIMIZE else -res.fun
    return float(obj), res.x, int(res.nit)


def main():
    out = []
    for fam, kw in CASES:
        lp = GEN[fam](**kw)
        obj, x, nit = solve_highs(lp)
        out.append(dict(family=fam, args=kw, name=lp["name"], objective=obj, highs_iterations=nit,
                        x_head=[float(v) for v in x[:8]]))
        print(lp["name"], obj, nit)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "highs_objectives.json"), "w") as f:
        json.dump(dict(generator="scipy.optimize.linprog(method='highs-ds'), scipy 1.15.3", cases=out), f, indent=1)


if __name__ == "__main__":
    main()
