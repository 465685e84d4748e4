"""Cost of Solution::add_constraint on a config-4-sized matrix (100 000 x 100 000, 10.1 M non-zeros, banded sweep on):
the costs are negated so that x = 0 is optimal at once, then K rows are appended one by one (each followed by its dual
re-solve, solver.rs:549-634).  Reports the wall time per call and the share spent in the dual pivots."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen

m, n, k = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (100000, 100000, 100)))
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 12
lp = lpgen.gen_sparse_lp(m, n, k, 4)
lp = dict(lp, obj=-lp["obj"])
t = time.time(); s = lpgen.build_problem(M.Problem, lp).solve(); print("solve %.3fs pivots=%d banded=%d" % (time.time() - t, s.stats()["iterations"], s.stats()["banded_sweep"]), flush=True)
rng = np.random.default_rng(1)
for i in range(rows):
    idx = np.sort(rng.choice(n, size=20, replace=False))
    it0 = s.stats()["iterations"]; w0 = s.stats()["solve_wall_s"]
    t = time.perf_counter()
    s = s.add_constraint([(int(j), 1.0) for j in idx], M.GE, 0.25 + 0.01 * i)
    dt = time.perf_counter() - t
    st = s.stats()
    print("add_constraint %2d: %.3f ms (pivots %d, objective %.9f)" % (i, dt * 1e3, st["iterations"] - it0, s.objective()), flush=True)
