"""A compact-factor solve in-process (for rocprofv3): family transport|mixed, S D deg, pivot limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get('MLP_IMPORT_TORCH'):
    import torch  # noqa: F401
import minilp_amd as M
from minilp_amd import lpgen
fam, S, D, deg, limit = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 20000
lp = lpgen.gen_mixed_lp(S, D, deg, 3) if fam == "mixed" else lpgen.gen_transport_lp(S, D, deg, tight=1.0)
prob = lpgen.build_problem(M.Problem, lp)
s = prob.solve(budget=0)
done = 0
t0 = time.perf_counter()
while True:
    t1 = time.perf_counter(); s.continue_solve(chunk); dt = time.perf_counter() - t1
    st = s.stats(); piv = int(st["iterations"])
    print(f"pivots {piv} us/pivot {dt*1e6/max(1,piv-done):.1f} nucleus {int(st['nucleus_size'])} cap {int(st['nucleus_capacity'])} factor {int(st['factor_active'])} levels {int(st['factor_levels'])} "
          f"refactors {int(st['factor_refactors'])} bump {int(st['factor_bump'])} switches {int(st['factor_switches'])} sb {s.state('factor_sb').astype(int).tolist()} obj {s.objective():.9f}", flush=True)
    done = piv
    if not s.budget_exhausted or (limit > 0 and piv >= limit):
        break
print(f"total {time.perf_counter()-t0:.2f} s, finished {not s.budget_exhausted}")
