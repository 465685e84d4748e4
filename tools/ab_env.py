#!/usr/bin/env python3
"""Generic A/B of environment knobs on one window of config 4.

    python tools/ab_env.py late "MLP_PB_DET=0" "MLP_PB_DET=1" [--reps 2] [--pivots 512]

window: early (pivots 5..25 x 200 fresh solves is too noisy: uses pivots 200..2200 from the slack basis), mid (k = 9 999) or
late (k = 20 493).  Each configuration runs in its own process (the knobs are read once per process), alternating, REPS times;
prints us per pivot of the graph-replayed run and the objective reached (so that a change of the pivot path is visible)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import gzip, sys, time
sys.path.insert(0, %r)
import bench, minilp_amd as M
from minilp_amd import lpgen
window, pivots = sys.argv[1], int(sys.argv[2])
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
if window == "early":
    s = prob.solve(budget=200)
else:
    blob = gzip.open(bench.LATE_BASIS if window == "late" else bench.MID_BASIS, "rb").read()
    s = prob.solve_from_basis(blob, budget=0)
    s.continue_solve(64)
t0 = time.perf_counter(); s.continue_solve(pivots); dt = time.perf_counter() - t0
print("%%.1f us/pivot, objective %%.12g, nucleus %%d" %% (dt * 1e6 / pivots, s.objective(), s.stats()["nucleus_size"]))
''' % ROOT

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
    pivots = int(sys.argv[sys.argv.index("--pivots") + 1]) if "--pivots" in sys.argv else 0
    args = [a for a in args if not a.isdigit()]
    window, cfgs = args[0], args[1:]
    if not pivots:
        pivots = {"early": 2000, "mid": 1024, "late": 512}[window]
    for _ in range(reps):
        for cfg in cfgs:
            env = dict(os.environ)
            for kv in cfg.split(","):
                if "=" in kv:
                    k, v = kv.split("=", 1)
                    env[k] = v
            out = subprocess.run([sys.executable, "-c", CODE, window, str(pivots)], env=env, capture_output=True, text=True)
            print("%-40s %s %s" % (cfg, out.stdout.strip().replace("\n", " | "), out.stderr.strip()[-300:]), flush=True)
