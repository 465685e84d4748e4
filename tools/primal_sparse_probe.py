#!/usr/bin/env python3
"""Hypersparse instances that run the PRIMAL loop (Max c'x, Ax <= b, x >= 0 from the slack basis; 4 non-zeros per row):
engine against the single-threaded restatement of the reference on the same box, us per pivot."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402
from oracle import minilp_oracle as O  # noqa: E402

cases = [(16000, 6000, 4, 4), (6000, 3000, 12, 4), (10000, 6000, 4, 4)]
no_cpu = "--no-cpu" in sys.argv  # (the restatement of the reference takes minutes on these: 2-10 ms per pivot once the basis has filled in)
argv = [a for a in sys.argv[1:] if a != "--no-cpu"]
if argv:
    cases = [tuple(int(x) for x in a.split(",")) for a in argv]
for args in cases:
    lp = lpgen.gen_sparse_lp(*args)
    if np.bincount(lp["indices"], minlength=lp["n"]).min() == 0:
        lp["hi"] = np.full(lp["n"], 5.0)  # empty columns: bound every variable (the run then has bound flips too)
    pg = lpgen.build_problem(M.Problem, lp)
    po = lpgen.build_problem(O.Problem, lp)
    pg.solve()
    bg = bo = 1e9
    for _ in range(2):
        t = time.perf_counter(); sg = pg.solve(); bg = min(bg, time.perf_counter() - t)
    so = sg
    if not no_cpu:
        t = time.perf_counter(); so = po.solve(); bo = time.perf_counter() - t
    st = sg.stats()
    it = st["iterations"]
    print(f"{args}: GPU {bg * 1e3:.1f} ms, {it} pivots, {bg * 1e6 / it:.1f} us/pivot (hypersparse {st['hyper_iters']}, handed back {st['hyper_bails']}, "
          f"nucleus {st['nucleus_size']}) | CPU {bo * 1e3:.1f} ms, {bo * 1e6 / it:.1f} us/pivot | objective {sg.objective():.9g} vs {so.objective():.9g}", flush=True)
