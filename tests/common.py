"""Shared helpers for the parity tests."""
import json
import math
import os

import numpy as np

from minilp_amd import lpgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = {"sparse": lpgen.gen_sparse_lp, "dense": lpgen.gen_dense_lp, "mixed": lpgen.gen_mixed_lp, "cover": lpgen.gen_cover_lp, "twophase": lpgen.gen_twophase_lp}

# Parity contract (SURVEY.md §7 hard part 5): f64 everywhere,
#   |dobj| <= OBJ_RTOL * max(1, |obj|);  |dx|inf <= X_ATOL on instances with a unique optimum.
OBJ_RTOL = 1e-9
X_ATOL = 1e-7
# HiGHS fixtures pin the optimum only to HiGHS' own tolerances.
HIGHS_RTOL = 1e-7


def highs_cases():
    with open(os.path.join(ROOT, "tests", "golden", "highs_objectives.json")) as f:
        return json.load(f)["cases"]


def obj_close(a, b, rtol=OBJ_RTOL):
    return abs(a - b) <= rtol * max(1.0, abs(a), abs(b))


def check_feasible(lp, x, tol=1e-7):
    """Primal feasibility of x for an lpgen instance: bounds and rows (size-independent property)."""
    x = np.asarray(x)
    assert (x >= lp["lo"] - tol).all() and (x <= lp["hi"] + tol).all()
    lhs = np.add.reduceat(lp["data"] * x[lp["indices"]], lp["indptr"][:-1])
    scale = np.maximum(1.0, np.abs(lp["rhs"]))
    ops, rhs = lp["ops"], lp["rhs"]
    viol = np.where(ops == lpgen.LE, lhs - rhs, np.where(ops == lpgen.GE, rhs - lhs, np.abs(lhs - rhs)))
    assert (viol <= tol * scale).all(), float(viol.max())


def objective_of(lp, x):
    return float(np.dot(lp["obj"], x))
