#!/usr/bin/env python3
"""us per pivot against the nucleus size over ONE solve of a gen_sparse_lp instance, in chunks of pivots (the regime map of
DESIGN.md §7: sparse tableau row -> eager dense update -> delayed update with folds).
  python tools/regime_curve.py 50000,20000,4,7 [chunk]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

args = tuple(int(x) for x in sys.argv[1].split(","))
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
lp = lpgen.gen_sparse_lp(*args)
if np.bincount(lp["indices"], minlength=lp["n"]).min() == 0:
    lp["hi"] = np.full(lp["n"], 5.0)
prob = lpgen.build_problem(M.Problem, lp)
t0 = time.perf_counter()
s = prob.solve(budget=chunk)
done = 0
while True:
    dt = time.perf_counter() - t0
    st = s.stats()
    n = st["iterations"] - done
    done = st["iterations"]
    print(f"pivots {done:8d}  nucleus {st['nucleus_size']:6d} (capacity {st['nucleus_capacity']:6d})  {dt * 1e6 / max(1, n):8.1f} us/pivot  "
          f"reinversions {st['reinversions']}  objective {s.objective():.9g}", flush=True)
    if not s.budget_exhausted:
        break
    t0 = time.perf_counter()
    s.continue_solve(chunk)
print("status: done,", done, "pivots")
