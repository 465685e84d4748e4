#!/usr/bin/env python3
"""us per pivot of the first pivots of config 4 from the slack basis, in chunks, against the nucleus size — once with the one-launch
form of BTRAN + pass + v tail + touch at every size the sparse tableau row allows (MLP_SMALL_BASIS_K=256) and once with the three
launches (MLP_SMALL_BASIS=0): where the two curves cross is the default of MLP_SMALL_BASIS_K.  python tools/small_basis_curve.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time
sys.path.insert(0, %r)
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
prob.solve(budget=40)
best = {}
for rep in range(3):
    s = prob.solve(budget=8)
    for i in range(10):
        t0 = time.perf_counter(); s.continue_solve(16); dt = time.perf_counter() - t0
        k = s.stats()["nucleus_size"]
        best[i] = min(best.get(i, (1e9, 0)), (dt * 1e6 / 16, k))
print(" ".join("%%d:%%.1f" %% (k, us) for us, k in (best[i] for i in sorted(best))), "| launches", int(s.state("small_basis_launches")[0]))
''' % ROOT
for cfg in ("MLP_SMALL_BASIS_K=256", "MLP_SMALL_BASIS=0", "MLP_SMALL_BASIS_K=128"):
    env = dict(os.environ)
    k, v = cfg.split("=")
    env[k] = v
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("%-24s %s %s" % (cfg, out.stdout.strip(), out.stderr.strip()[-300:]), flush=True)
