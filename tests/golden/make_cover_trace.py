"""Oracle trace fixture for a large dual-only instance (cover family, 40 000 x 40 000, 60 nnz/row): the first N
pivots of the dual loop.  Run in the build container:  python tests/golden/make_cover_trace.py 1500"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minilp_amd import lpgen  # noqa: E402
from oracle import minilp_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
lp = lpgen.gen_cover_lp(40000, 40000, 60, 8)
t = time.time()
s = lpgen.build_problem(O.Problem, lp).solve(budget=N, trace=True)
tr = s.trace()
print("oracle", len(tr), "pivots in %.1f s; obj %r; stats %s" % (time.time() - t, s.objective(), {k: s.stats()[k] for k in ("primal_iters", "dual_iters")}), flush=True)
arr = np.array([[t_[0], t_[1], t_[2], t_[3], t_[4]] for t_ in tr], dtype=np.int32)
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "cover40k_oracle_trace.npz")
np.savez_compressed(OUT, trace=arr,
                    objective=np.array(s.objective()), pivots=np.array(len(tr)))
