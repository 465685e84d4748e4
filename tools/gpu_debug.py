"""Ad-hoc GPU bring-up script: run product vs oracle on a ladder of cases and report the first divergence."""
import math
import sys
import time
import traceback

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O

INF = math.inf


def compare(lp, budget=-1, label=""):
    po = lpgen.build_problem(O.Problem, lp)
    pg = lpgen.build_problem(M.Problem, lp)
    t = time.time()
    so = po.solve(budget=budget, trace=True)
    to = time.time() - t
    t = time.time()
    sg = pg.solve(budget=budget, trace=True)
    tg = time.time() - t
    tro, trg = so.trace(), sg.trace()
    ndiff = None
    for i, (a, b) in enumerate(zip(tro, trg)):
        if a[:5] != b[:5]:
            ndiff = i
            break
    oo, og = so.objective(), sg.objective()
    xo, xg = so.values(), sg.values()
    print(f"[{label}] oracle obj={oo:.12g} ({len(tro)} piv, {to:.2f}s) gpu obj={og:.12g} ({len(trg)} piv, {tg:.2f}s) "
          f"|dobj|={abs(oo-og):.3e} |dx|inf={np.abs(xo-xg).max():.3e} first trace diff={ndiff} stats={sg.stats()['nucleus_size']}")
    if ndiff is not None:
        print("   oracle:", tro[max(0, ndiff-1):ndiff+2])
        print("   gpu   :", trg[max(0, ndiff-1):ndiff+2])
    return so, sg


def main():
    print("devices", M.device_count())
    p = M.Problem(M.MAXIMIZE)
    x = p.add_var(1.0, (0.0, INF))
    y = p.add_var(2.0, (0.0, 3.0))
    p.add_constraint([(x, 1.0), (y, 1.0)], M.LE, 4.0)
    p.add_constraint([(x, 2.0), (y, 1.0)], M.GE, 2.0)
    s = p.solve(trace=True)
    print("toy:", s.objective(), s[x], s[y], s.trace(), s.stats())
    for (m, n, k, seed) in [(5, 5, 3, 1), (20, 20, 5, 2), (50, 40, 8, 3), (200, 200, 10, 4), (300, 500, 20, 5)]:
        try:
            compare(lpgen.gen_sparse_lp(m, n, k, seed), label=f"sparse {m}x{n} k{k}")
        except Exception:
            traceback.print_exc()
    for (m, n, seed) in [(10, 10, 1), (60, 60, 2), (150, 100, 3)]:
        try:
            so, sg = compare(lpgen.gen_dense_lp(m, n, seed), label=f"dense {m}x{n}")
            print("   reinvert diff:", sg.reinvert())
        except Exception:
            traceback.print_exc()
    for (m, n, k, seed) in [(8, 8, 3, 1), (30, 40, 5, 2), (100, 150, 6, 3), (300, 400, 8, 4)]:
        try:
            compare(lpgen.gen_mixed_lp(m, n, k, seed), label=f"mixed {m}x{n} k{k}")
        except Exception:
            traceback.print_exc()


if __name__ == "__main__":
    main()
