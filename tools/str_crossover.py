"""Crossover of the sparse tableau row against the sweep over all of A on config 4: us per pivot in chunks of 64 pivots
from the slack basis, with MLP_STR_K = 100000 (always sparse) and 0 (never)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
res = {}
for mode in ("100000", "0"):
    os.environ["MLP_STR_K"] = mode
    s = prob.solve(budget=0)
    s.continue_solve(8)
    rows = []
    for chunk in range(16):
        t = time.perf_counter(); s.continue_solve(64); dt = time.perf_counter() - t
        rows.append((int(s.stats()["nucleus_size"]), dt * 1e6 / 64))
    res[mode] = rows
    del s
for a, b in zip(res["100000"], res["0"]):
    print(f"k = {a[0]:5d}: sparse row {a[1]:7.1f} us/pivot | sweep {b[1]:7.1f} us/pivot")
