import sys, time, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from minilp_amd import lpgen
from oracle import minilp_oracle as O
N = int(sys.argv[1])
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
t = time.time()
s = lpgen.build_problem(O.Problem, lp).solve(budget=N, trace=True)
tr = s.trace()
print("oracle", len(tr), "pivots in", time.time() - t, "s; obj", s.objective(), flush=True)
arr = np.array([[t_[0], t_[1], t_[2], t_[3], t_[4]] for t_ in tr], dtype=np.int32)
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg4_oracle_trace.npz")
np.savez_compressed(OUT, trace=arr, objective=np.array(s.objective()), pivots=np.array(len(tr)))
