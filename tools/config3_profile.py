"""Config 3 (mixed 6000 x 10000, 4 nnz/row, through the MPS reader) on the GPU engine: wall time, pivots, nucleus size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("MLP_IMPORT_TORCH"):
    import torch  # noqa: F401
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_mixed_lp(6000, 10000, 4, 3)
text = lpgen.to_mps(lp)
for rep in range(3):
    p = M.MpsFile(text, lp["direction"]).problem
    t = time.time(); s = p.solve(); dt = time.time() - t
    st = s.stats()
    print(f"rep {rep}: {dt:.4f}s pivots={st['iterations']} primal={st['primal_iters']} dual={st['dual_iters']} k={st['nucleus_size']} cap={st['nucleus_capacity']} "
          f"us/pivot={dt*1e6/st['iterations']:.1f} obj={s.objective():.9f} kase={st['kase']}", flush=True)
