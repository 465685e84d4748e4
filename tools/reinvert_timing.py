#!/usr/bin/env python3
"""Wall time of the blocked re-inversion of the nucleus (csrc/inverse.inc) at config-4 size: loading a committed basis =
classify the columns + assemble K + invert it in place.  python tools/reinvert_timing.py [mid|late] [reps]"""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "late"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read()
for i in range(reps):
    t0 = time.perf_counter()
    s = prob.solve_from_basis(blob, budget=0)
    t1 = time.perf_counter()
    s.continue_solve(64)
    st = s.stats()
    print(f"{which} load {i}: {t1 - t0:.3f} s (k = {st['nucleus_size']}), objective after 64 pivots {s.objective():.10f}, max_pivot_err {st['max_pivot_err']:.2e}", flush=True)
    del s
