#!/usr/bin/env python3
"""The driver-timed window of bench.py (config 4 from the slack basis: W warm-up pivots, then K timed) under the forms of the small-nucleus
primal iteration, in ONE process: five launches (MLP_PRIMAL_HEAD=0), primal head + k_row_pull + update (MLP_PULL_INSIDE=0), primal
head + update with the tableau row pulled inside (default).  Each form: R fresh solves, pivots/s of each, and the traces compared.
  python tools/early_ab.py [K=20] [W=5] [R=7]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
R = int(sys.argv[3]) if len(sys.argv) > 3 else 7
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
FORMS = [("five launches", {"MLP_PRIMAL_HEAD": "0"}), ("head + pull + update", {"MLP_PRIMAL_HEAD": "1", "MLP_PULL_INSIDE": "0"}),
         ("head + update(pull inside)", {"MLP_PRIMAL_HEAD": "1", "MLP_PULL_INSIDE": "1", "MLP_HEAD_APPLY": "0"}),
         ("head(applies) + update(n side)", {"MLP_PRIMAL_HEAD": "1", "MLP_PULL_INSIDE": "1", "MLP_HEAD_APPLY": "1"})]
traces = {}
for name, env in FORMS:
    for k_, v_ in env.items():
        os.environ[k_] = v_
    rates = []
    for rep in range(R):
        s = prob.solve(budget=0, trace=True)
        s.continue_solve(W)
        t0 = time.perf_counter()
        s.continue_solve(K)
        dt = time.perf_counter() - t0
        rates.append(K / dt)
        if rep == 0:
            s.continue_solve(200)   # the traces are compared further than the timed window
            traces[name] = [t[:5] for t in s.trace()]
            heads = int(s.state("primal_head_launches")[0])
        del s
    print(f"{name:32s}: median {statistics.median(rates):9.1f} pivots/s ({1e6 / statistics.median(rates):6.1f} us/pivot)  best {max(rates):9.1f}  "
          f"all {[round(r) for r in rates]}  head iterations in {W + K + 200} pivots: {heads}", flush=True)
    for k_ in env:
        del os.environ[k_]
names = list(traces)
for nm in names[1:]:
    print(f"trace of '{nm}' == trace of '{names[0]}': {traces[nm] == traces[names[0]]} ({len(traces[nm])} pivots)")
