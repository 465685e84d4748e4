#!/usr/bin/env python3
"""A/B of the pulled F product (fpull.inc) against the blocked push, one window of config 4 from a committed basis.

  python tools/fpull_ab.py {mid|late} PIVOTS

Runs the window twice in one process (MLP_FPULL=1, then 0: the knob is read when a Solution is created), prints us per pivot, whether the
pull was in use, and compares the pivot sequences and objectives of the two runs."""
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
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read()
res = {}
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for on in ("1", "0"):
        os.environ["MLP_FPULL"] = on
        s = prob.solve_from_basis(blob, budget=0, trace=True)
        s.continue_solve(32)
        t0 = time.perf_counter()
        s.continue_solve(pivots)
        dt = time.perf_counter() - t0
        fp = s.state("fpull")
        res[on] = ([t[:5] for t in s.trace()], s.objective())
        x1 = s.values().copy()
        s.recompute_basic_values()   # x_B = B^-1 (b - N x_N) from scratch: a wrong alpha_q in any pivot shows as a gap to the maintained values
        x2 = s.values()
        import numpy as np
        print(f"   max |x maintained - x recomputed| = {np.abs(x1 - x2).max():.3e} (max |x| {np.abs(x2).max():.3e})", flush=True)
        print(f"{which} MLP_FPULL={on}: {dt * 1e6 / pivots:.1f} us/pivot, fpull state {fp.tolist()}, k = {s.stats()['nucleus_size']}, max_pivot_err {s.stats()['max_pivot_err']:.2e}", flush=True)
        del s
ta, tb = res["1"][0], res["0"][0]
same = sum(1 for a, b in zip(ta, tb) if a == b)
first = next((i for i, (a, b) in enumerate(zip(ta, tb)) if a != b), None)
print(f"pivots identical: {same} of {len(ta)} (first difference at {first}); objectives {res['1'][1]:.12g} / {res['0'][1]:.12g}", flush=True)
