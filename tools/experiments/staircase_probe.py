"""The staircase family (lpgen.gen_staircase_lp: multi-period production / inventory) on the default path and with the compact factor forced:
its bump grows to half the rows and its elimination fills in a dense tail (rows of 60-80 entries: tools/experiments/bump_lu.py staircase …) —
the sparse LU of the bump must hand over (dense inverse up to 1 024 columns, explicit nucleus inverse beyond) and the optimum must stand."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
T, R, P = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (40, 100, 150)))
lp = lpgen.gen_staircase_lp(T, R, P, max(1, R // 4))
print("rows", lp["m"], "cols", lp["n"], "nnz", len(lp["data"]), flush=True)
if os.environ.get("ORACLE", "1") != "0":
    t = time.perf_counter(); so = lpgen.build_problem(O.Problem, lp).solve(); to = time.perf_counter() - t
    print(f"oracle: {to:.2f} s, {so.stats()['pivots']} pivots, obj {so.objective():.9f}", flush=True)
for name, env in (("default", {}), ("factor forced", {"MLP_FACTOR": "1"}), ("factor forced, sparse bump from 2", {"MLP_FACTOR": "1", "MLP_FACTOR_SB_FROM": "2"})):
    for k in ("MLP_FACTOR", "MLP_FACTOR_SB_FROM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = lpgen.build_problem(M.Problem, lp)
    p.solve()
    t = time.perf_counter(); s = p.solve(); dt = time.perf_counter() - t
    st = s.stats()
    print(f"{name:36s} {dt:.3f} s, {st['iterations']} pivots ({dt * 1e6 / max(1, st['iterations']):.0f} us/pivot), obj {s.objective():.9f}, nucleus {st['nucleus_size']}, factor {st['factor_active']} "
          f"switches {st['factor_switches']} bump max {st['factor_bump_max']} sb {s.state('factor_sb').astype(int).tolist()}", flush=True)
