"""Debug: accuracy of the reduced costs of the UNSHARDED solve after P pivots from the late basis under environment toggles."""
import gzip, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import minilp_amd as M
from minilp_amd import lpgen
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
p = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(os.path.join(ROOT, "tests/golden/cfg4_basis_p240000.bin.gz"), "rb").read()
s = p.solve_from_basis(blob, budget=0)
s.continue_solve(P)
rd = s.state("nb_var_obj_coeffs")
b0 = s.save_basis(0)
obj = s.objective()
del s
for k in ("MLP_SWEEP_LOCALITY", "MLP_RATIO_TWO_KERNELS", "MLP_BANDED", "MLP_SWEEP_PACKED"):
    os.environ.pop(k, None)
fresh = p.solve_from_basis(b0, budget=0)
e = np.abs(rd - fresh.state("nb_var_obj_coeffs"))
print("unsharded %d pivots objective %.6f: d max err %.2e (median %.1e), |d| median %.2e" % (P, obj, e.max(), float(np.median(e)), float(np.median(np.abs(rd)))), flush=True)
