"""GPU perf probe: run config-4-family instance for chunks of pivots, print pivots/s and kernel stats."""
import sys
import time

import numpy as np

import os
if os.environ.get("MLP_IMPORT_TORCH"):  # rocprofv3 crashes inside graph capture with the system HIP runtime; torch's bundled one works
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minilp_amd as M
from minilp_amd import lpgen

m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
chunk, nchunks = int(sys.argv[4]), int(sys.argv[5])
profile = len(sys.argv) > 6 and sys.argv[6] == "prof"
t = time.time(); lp = lpgen.gen_sparse_lp(m, n, k, 4); print("gen %.2fs" % (time.time() - t), flush=True)
t = time.time(); p = lpgen.build_problem(M.Problem, lp); print("build %.2fs" % (time.time() - t), flush=True)
t = time.time(); s = p.solve(budget=0, profile=profile); print("init %.2fs" % (time.time() - t), flush=True)
for c in range(nchunks):
    s.reset_stats()
    t = time.time(); s.continue_solve(chunk); dt = time.time() - t
    st = s.stats()
    its = st["iterations"]
    line = f"chunk {c}: {dt:.3f}s its={its} piv/s={its/max(dt,1e-9):.0f} k={st['nucleus_size']} cap={st['nucleus_capacity']} obj={s.objective():.6f}"
    if profile and st["fused_launches"]:
        line += f" fused: {st['fused_ms']/st['fused_launches']*1e3:.1f}us/launch {st['fused_bytes']/st['fused_ms']/1e6:.1f}GB/s"
    if profile and st["sweep_launches"]:
        line += f" sweep: {st['sweep_ms']/st['sweep_launches']*1e3:.1f}us/launch {st['sweep_bytes']/st['sweep_ms']/1e6:.1f}GB/s"
    line += f" kase={st['kase']} piverr={st['max_pivot_err']:.2e} reinv={st['reinversions']}"
    print(line, flush=True)
    if not s.budget_exhausted:
        break
x = np.asarray(s.values())
print("objective accumulated %.9f  recomputed c.x %.9f  status %s" % (s.objective(), float(np.dot(lp["obj"], x)), "optimal" if not s.budget_exhausted else "budget"), flush=True)  # obj_from_x
if not os.environ.get("MLP_NO_REINV"):
    print("reinvert diff", s.reinvert())
