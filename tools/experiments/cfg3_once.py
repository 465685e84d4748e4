"""Config 3 through the MPS reader, three solves on the default (hypersparse + multi-kernel) path: for a kernel-trace of the dense tail."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get('MLP_IMPORT_TORCH'):
    import torch  # noqa: F401  (rocprofv3 + graph capture: torch's bundled HIP runtime is the one that works)
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_mixed_lp(6000, 10000, 4, 3)
pg = M.MpsFile(lpgen.to_mps(lp), lp["direction"]).problem
for _ in range(3):
    t = time.perf_counter(); s = pg.solve(); dt = time.perf_counter() - t
    st = s.stats()
    print(f"solve {dt*1e3:.1f} ms pivots {st['iterations']} hyper {st['hyper_iters']} bails {st['hyper_bails']} nucleus {st['nucleus_size']}", flush=True)
