#!/usr/bin/env python3
"""MLP_KPROF=1 diagnostics: wall-clock timeline of the LAST iteration of a batch (kernel entries, marks inside the primal
ratio test and the update kernel), for the early window of config 4 and for the late basis.
  python tools/kernel_timeline.py [early|late|mid] [pivots]"""
import gzip
import os
import sys

os.environ["MLP_KPROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

NAMES = ["ftran (head) entry", "ratio entry", "ratio: loads done", "ratio: step bound", "ratio: winner", "ratio: decision", "ratio: head+plan done",
         "btran entry", "W pass entry", "v tail entry", "row touch entry", "row pull entry", "update entry", "update: block 0 done",
         "update: last block reduced", "update: next iteration open", "update: last ticket taken", "update: partials read",
         "ftran gather entry", "push stage 1 entry", "push combine entry", "stream pass entry", "banded sweep entry", "-"]
which = sys.argv[1] if len(sys.argv) > 1 else "early"
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
if which == "early":
    s = prob.solve(budget=0)
else:
    s = prob.solve_from_basis(gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read(), budget=0)
s.continue_solve(pivots)
for rep in range(3):
    s.continue_solve(7)
    tl = s.state("kernel_timeline")
    newest = max(tl)
    order = sorted((t, i) for i, t in enumerate(tl) if t >= 0 and newest - t < 5000.0)  # (marks of kernels this regime does not run are stale)
    print(f"{which}, k = {s.stats()['nucleus_size']}: " + " | ".join(f"{NAMES[i]} {t:.2f}" for t, i in order), flush=True)
