#!/usr/bin/env python3
"""Config 3 (mixed 6000 x 10000, 4 per row, through the MPS reader) on the MULTI-KERNEL path only, for rocprofv3:
  MLP_HYPER=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -- python tools/cfg3_profile.py
(the per-kernel picture of the dual iteration at a small size: the dense tail of config 3 runs there)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("MLP_IMPORT_TORCH"):
    import torch  # noqa: F401
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

lp = lpgen.gen_mixed_lp(6000, 10000, 4, 3)
pg = M.MpsFile(lpgen.to_mps(lp), lp["direction"]).problem
pg.solve()
t = time.perf_counter(); s = pg.solve(); dt = time.perf_counter() - t
st = s.stats()
print(f"solve {dt * 1e3:.1f} ms, {st['iterations']} pivots, {dt * 1e6 / st['iterations']:.1f} us/pivot, hyper {st['hyper_iters']}, nucleus {st['nucleus_size']}")
