"""In-process run of the transport family on the compact factor (for rocprofv3: the evidence tool runs its solves in subprocesses)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("MLP_IMPORT_TORCH"):
    import torch  # noqa: F401  (rocprofv3 + graph capture: torch's bundled HIP runtime is the one that works)
import minilp_amd as M
from minilp_amd import lpgen
S, D, deg, tight, skip, run = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
lp = lpgen.gen_mixed_lp(S, D, deg, 3) if os.environ.get("FAMILY") == "mixed" else lpgen.gen_transport_lp(S, D, deg, tight=tight)
s = lpgen.build_problem(M.Problem, lp).solve(budget=skip)
t0 = time.perf_counter(); s.continue_solve(run); dt = time.perf_counter() - t0
st = s.stats()
print("pivots %d..%d: %.1f us/pivot, levels %d, bump %d, refactorisations %d, factor %d" % (skip, skip + run, dt * 1e6 / run, st["factor_levels"], st["factor_bump"], st["factor_refactors"], st["factor_active"]))
