#!/usr/bin/env python3
"""One window of bench.py on its own, for rocprofv3: load a committed mid-solve basis of config 4 and run P pivots.

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_late -- python tools/window_profile.py late 256

The kernel statistics of that run are the per-kernel picture of the regime (the rocSOLVER kernels of the one
re-inversion at load time appear in the list too)."""
import gzip
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("MLP_IMPORT_TORCH"):
    import torch  # noqa: F401  (rocprofv3 + graph capture: torch's bundled HIP runtime is the one that works)
import bench  # noqa: E402
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "late"
pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 256
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 32
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
prob = lpgen.build_problem(M.Problem, lp)
if which == "early":
    s = prob.solve(budget=0)
else:
    blob = gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read()
    s = prob.solve_from_basis(blob, budget=0)
s.continue_solve(warm)
t0 = time.perf_counter()
s.continue_solve(pivots)
dt = time.perf_counter() - t0
st = s.stats()
print(f"{which}: {pivots} pivots in {dt:.4f}s = {pivots / dt:.1f} pivots/s, {dt * 1e6 / pivots:.1f} us/pivot, k = {st['nucleus_size']}, cap = {st['nucleus_capacity']}", flush=True)
