"""Diagnostic: the same solve with different continue_solve() chunk sizes (wall time, full statistics)."""
import sys, time, json
sys.path.insert(0, ".")
from minilp_amd import lpgen
import minilp_amd as B
m, n, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lp = lpgen.gen_mixed_lp(m, n, 4, 3)
prob = lpgen.build_problem(B.Problem, lp)
t0 = time.perf_counter()
s = prob.solve(budget=0)
while True:
    t1 = time.perf_counter()
    s.continue_solve(chunk)
    st = s.stats()
    print("pivots %d  %.2f s  nucleus %d factor %d" % (st["iterations"], time.perf_counter() - t1, st["nucleus_size"], st["factor_active"]), flush=True)
    if not s.budget_exhausted:
        break
print("chunk", chunk, "wall %.2f s" % (time.perf_counter() - t0), s.objective())
print(json.dumps({k: (int(v) if float(v).is_integer() else round(float(v), 3)) for k, v in s.stats().items()}))
