"""Accuracy of the dual steepest-edge weights after N pivots of the cover40k instance: the engine's weights
(dumped on the GPU box by tools/dump_dse.py N -> gpurun_out/dse_gpu_N.npz) and the oracle's, both against the
exact values ||B^-T e_i||^2 computed from the same basis with a sparse LU (scipy).  CPU only.
usage: python tools/dse_accuracy.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from minilp_amd import lpgen
from oracle import minilp_oracle as O

N = int(sys.argv[1])
lp = lpgen.gen_cover_lp(40000, 40000, 60, 8)
m, n = lp["m"], lp["n"]
A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n))
Afull = sp.hstack([A, sp.identity(m, format="csr")], format="csc")
s = lpgen.build_problem(O.Problem, lp).solve(budget=N)
bo, bvo = s.state("dual_edge_sq_norms"), s.state("basic_vars").astype(np.int64)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dse_gpu_%d.npz" % N))
bg, bvg = g["beta"], g["basic_vars"]
print("same basis after %d pivots:" % N, bool((bvo == bvg).all()))
lu = spl.splu(Afull[:, bvo].tocsc())
rel_o, rel_g = [], []
for i in np.random.default_rng(1).choice(m, 200, replace=False):
    e = np.zeros(m); e[i] = 1.0
    rho = lu.solve(e, trans="T")
    t = float(rho @ rho)
    rel_o.append(abs(bo[i] - t) / t); rel_g.append(abs(bg[i] - t) / t)
print("200 random rows, relative error of the stored weight against ||B^-T e_i||^2:")
print("  oracle (reference algorithm): median %.2e  max %.2e" % (np.median(rel_o), max(rel_o)))
print("  GPU engine                  : median %.2e  max %.2e" % (np.median(rel_g), max(rel_g)))
