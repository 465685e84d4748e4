"""Oracle trace fixture for a two-phase instance at scale (gen_twophase_lp 10 000 x 10 000, 30 nnz/row, every 40th row a >= row): dual loop
on the artificial objective, recalc_obj_coeffs, then the primal loop.  python tests/golden/make_twophase_trace.py 6000"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minilp_amd import lpgen  # noqa: E402
from oracle import minilp_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
lp = lpgen.gen_twophase_lp(10000, 10000, 30, 9, ge_every=40)
t = time.time()
s = lpgen.build_problem(O.Problem, lp).solve(budget=N, trace=True)
tr = s.trace()
st = s.stats()
print("oracle", len(tr), "pivots in %.1f s; obj %r; primal %d dual %d" % (time.time() - t, s.objective(), st["primal_iters"], st["dual_iters"]), flush=True)
arr = np.array([[t_[0], t_[1], t_[2], t_[3], t_[4]] for t_ in tr], dtype=np.int32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "twophase10k_oracle_trace.npz"), trace=arr,
                    objective=np.array(s.objective()), pivots=np.array(len(tr)), dual_iters=np.array(st["dual_iters"]))
