"""Dump the dual steepest-edge weights and the basis after N pivots of the cover40k instance (analysis aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
N = int(sys.argv[1])
lp = lpgen.gen_cover_lp(40000, 40000, 60, 8)
s = lpgen.build_problem(M.Problem, lp).solve(budget=N)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/dse_gpu_%d.npz" % N, beta=s.state("dual_edge_sq_norms"), basic_vars=s.state("basic_vars").astype(np.int64),
                    xB=s.state("basic_var_vals"), lo=s.state("basic_var_mins"), hi=s.state("basic_var_maxs"))
print("dumped", N, s.stats()["nucleus_size"])
