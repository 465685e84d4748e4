"""Differential fuzz: small random LPs with every kind of bound / operator / direction through the
oracle and the HIP engine; statuses must agree and objectives match to the parity tolerance."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minilp_amd as M
from oracle import minilp_oracle as O

INF = math.inf


def gen(rng):
    n = int(rng.integers(1, 9))
    m = int(rng.integers(0, 9))
    direction = int(rng.integers(0, 2))
    vars_ = []
    for _ in range(n):
        kind = rng.integers(0, 7)
        c = float(rng.integers(-4, 5))
        lo, hi = {0: (0.0, INF), 1: (0.0, float(rng.integers(1, 6))), 2: (-INF, INF), 3: (float(rng.integers(-3, 1)), INF),
                  4: (-INF, float(rng.integers(0, 4))), 5: (2.0, 2.0), 6: (float(rng.integers(-2, 2)), float(rng.integers(2, 5)))}[int(kind)]
        vars_.append((c, lo, hi))
    cons = []
    for _ in range(m):
        k = int(rng.integers(0, min(n, 4) + 1))
        idx = rng.choice(n, size=k, replace=False) if k else np.array([], dtype=int)
        coef = rng.integers(-3, 4, size=k).astype(float)
        coef[coef == 0] = 1.0
        cons.append((idx.tolist(), coef.tolist(), int(rng.integers(0, 3)), float(rng.integers(-6, 10))))
    return direction, vars_, cons


def build(B, inst):
    direction, vars_, cons = inst
    p = B.Problem(direction)
    for c, lo, hi in vars_:
        p.add_var(c, (lo, hi))
    for idx, coef, op, rhs in cons:
        p.add_constraint(list(zip(idx, coef)), op, rhs)
    return p


def outcome(B, inst):
    try:
        s = build(B, inst).solve()
        return "ok", s.objective(), s
    except (B.Infeasible,):
        return "infeasible", None, None
    except (B.Unbounded,):
        return "unbounded", None, None


def main(n_cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    counts = {}
    for i in range(n_cases):
        inst = gen(rng)
        if os.environ.get("FUZZ_VERBOSE"):
            print("case", i, inst, flush=True)
        so, oo, sol_o = outcome(O, inst)
        sg, og, sol_g = outcome(M, inst)
        counts[so] = counts.get(so, 0) + 1
        # where the reference algorithm itself ends in a non-finite objective (inf - inf territory) the
        # engine must too; which non-finite value comes out depends on the order of the arithmetic
        same_nan = so == "ok" and sg == "ok" and not math.isfinite(oo) and not math.isfinite(og)
        ok = same_nan or (so == sg and (so != "ok" or oo == og or abs(oo - og) <= 1e-9 * max(1.0, abs(oo))))
        if same_nan:
            counts["nan"] = counts.get("nan", 0) + 1
            continue
        if ok and so == "ok" and i % 3 == 0 and len(inst[1]) >= 2:
            # warm start: an extra row, then fix/unfix
            v = int(rng.integers(0, len(inst[1])))
            xo_v, xg_v = float(sol_o[v]), float(sol_g[v])  # read before the mutators consume the solutions
            try:
                a = sol_o.add_constraint([(v, 1.0)], O.LE, xo_v - 0.5)
                ra = ("ok", a.objective())
            except O.Infeasible:
                ra = ("infeasible", None)
            try:
                b = sol_g.add_constraint([(v, 1.0)], M.LE, xg_v - 0.5)
                rb = ("ok", b.objective())
            except M.Infeasible:
                rb = ("infeasible", None)
            if abs(xo_v - xg_v) < 1e-9:  # same vertex => same cut => comparable
                ok = ra[0] == rb[0] and (ra[0] != "ok" or ra[1] == rb[1] or (ra[1] != ra[1] and rb[1] != rb[1]) or abs(ra[1] - rb[1]) <= 1e-8 * max(1.0, abs(ra[1])))
        if not ok:
            bad += 1
            print("MISMATCH case", i, inst, (so, oo), (sg, og), flush=True)
            if bad > 5:
                break
    print("cases", n_cases, "outcomes", counts, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
