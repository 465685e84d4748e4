"""Dump the primal steepest-edge weights and the basis after N pivots of config 4 (analysis aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
N = int(sys.argv[1])
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
s = lpgen.build_problem(M.Problem, lp).solve(budget=N)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/pse_gpu_%d.npz" % N, gamma=s.state("primal_edge_sq_norms"), nb_vars=s.state("nb_vars").astype(np.int64),
                    basic_vars=s.state("basic_vars").astype(np.int64), d=s.state("nb_var_obj_coeffs"))
print("dumped", N, s.stats()["nucleus_size"])
