"""Debug: step one pivot from the late basis and report non-finite entries of every stage vector."""
import gzip, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import minilp_amd as M
from minilp_amd import api as A, lpgen
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
p = lpgen.build_problem(M.Problem, lp)
blob = gzip.open(os.path.join(ROOT, "tests/golden/cfg4_basis_p240000.bin.gz"), "rb").read()
s = p.solve_from_basis(blob, budget=0)
st, info = s.engine_open()
names = {A.STAGE_FTRAN: ["col_coeffs"], A.STAGE_BTRAN: ["inv_basis_row_coeffs"], A.STAGE_BASIS: ["tau", "v"], A.STAGE_ROW: ["row_coeffs", "sq_norms_update_helper"]}
for it in range(2):
    while True:
        stage = info["next_stage"]
        st, info = s.engine_stage(stage)
        for nm in names.get(stage, []):
            a = s.state(nm)
            bad = ~np.isfinite(a)
            print("iter", it, "stage", stage, nm, "non-finite:", int(bad.sum()), "first:", np.nonzero(bad)[0][:8].tolist(), "max|x| %.3e" % np.abs(a[~bad]).max(), flush=True)
        if stage == A.STAGE_APPLY or st not in (A.ITER_PIVOT, A.ITER_FLIP):
            break
print("pb_det env:", os.environ.get("MLP_PB_DET"))
