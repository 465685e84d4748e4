#!/usr/bin/env python3
"""A/B of the fold kernel on the late window of config 4 (k = 20 493): the two forms of the fold kernel — U through the
scalar unit (default), U staged in LDS (MLP_FOLD_SCALAR=0).  Prints us/pivot and the sampled fold kernel time of each."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import gzip, sys, time
sys.path.insert(0, %r)
import bench, minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(bench.LATE_BASIS, "rb").read()
s = prob.solve_from_basis(blob, budget=0)
s.set_sampling(1)
s.continue_solve(64)
t0 = time.perf_counter(); s.continue_solve(256); dt = time.perf_counter() - t0
st = s.stats()
print("sampled: fold %%.1f us x %%d, %%.1f us/pivot" %% (1e3 * st["fold_ms"] / max(1, st["fold_launches"]), st["fold_launches"], dt * 1e6 / 256))
s = prob.solve_from_basis(blob, budget=0)
s.continue_solve(64)
t0 = time.perf_counter(); s.continue_solve(512); dt = time.perf_counter() - t0
print("graph:   %%.1f us/pivot, objective %%.12g" %% (dt * 1e6 / 512, s.objective()))
''' % ROOT
for val in ("scalar", "lds", "scalar", "lds"):
    env = dict(os.environ, MLP_FOLD_SCALAR="0" if val == "lds" else "1")
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("fold = " + val + ":", out.stdout.strip().replace("\n", " | "), out.stderr.strip()[-300:], flush=True)
