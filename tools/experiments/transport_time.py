"""Wall time of the 200 000-row transport solve on the compact factor (bench.py's factor_transport leg without the oracle): first 20 000 pivots, then the rest."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("MLP_IMPORT_TORCH"):  # rocprofv3 crashes inside graph capture with the system HIP runtime; torch's bundled one works
    import torch  # noqa: F401
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_transport_lp(100000, 100000, 4, tight=0.4)
p = lpgen.build_problem(M.Problem, lp)
t0 = time.perf_counter()
s = p.solve(budget=0)
s.continue_solve(20000)
t1 = time.perf_counter()
s.continue_solve(-1)
t2 = time.perf_counter()
st = s.stats()
n = int(st["iterations"])
print("first 20000: %.3f s (%.1f us/pivot)   whole: %.3f s, %d pivots (%.1f us/pivot)  objective %.10g" % (t1 - t0, (t1 - t0) / 20000 * 1e6, t2 - t0, n, (t2 - t0) / n * 1e6, s.objective()))
