"""Drift of the Gram matrix M over a long stretch of pivots without rebuilds: the monitor
|a_q.v - ||alpha_q||^2| / (1 + ||alpha_q||^2), maximum per chunk.  usage: gram_drift.py [late|mid] [chunks] [pivots per chunk]"""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MLP_GRAM_TOL", "1e9")
import bench
import minilp_amd as M
from minilp_amd import lpgen
which = sys.argv[1] if len(sys.argv) > 1 else "late"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
per = int(sys.argv[3]) if len(sys.argv) > 3 else 256
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read()
s = prob.solve_from_basis(blob, budget=0)
for i in range(chunks):
    s.reset_stats()
    t0 = time.perf_counter()
    s.continue_solve(per)
    dt = time.perf_counter() - t0
    st = s.stats()
    print(f"chunk {i}: {dt * 1e6 / per:.1f} us/pivot, k = {st['nucleus_size']}, gram_err {st['gram_err']:.2e}, rebuilds {st['gram_rebuilds']}, "
          f"pivot_err {st['max_pivot_err']:.2e}, obj {s.objective():.9f}", flush=True)
