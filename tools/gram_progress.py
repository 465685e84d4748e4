"""Objective progress per pivot from a saved basis of config 4, Gram mode against the streaming pass: the pricing
weights are only as good as v.  usage: gram_progress.py [mid|late] [pivots]   (MLP_GRAM etc. from the environment)"""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import minilp_amd as M
from minilp_amd import lpgen
which = sys.argv[1] if len(sys.argv) > 1 else "mid"
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read()
s = prob.solve_from_basis(blob, budget=0)
o0 = s.objective()
t0 = time.perf_counter()
for i in range(4):
    s.continue_solve(pivots // 4)
    st = s.stats()
    print(f"  after {st['iterations']} pivots: obj {s.objective():.6f} (+{s.objective() - o0:.3f}), k {st['nucleus_size']}, "
          f"monitor {st['gram_err']:.1e}, rebuilds {st['gram_rebuilds']}, {(time.perf_counter() - t0):.1f}s", flush=True)
