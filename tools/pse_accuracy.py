"""Accuracy of the primal steepest-edge weights after N pivots of config 4: the engine's (tools/dump_pse.py N ->
gpurun_out/pse_gpu_N.npz) and the oracle's, against the exact 1 + ||B^-1 a_j||^2 from a sparse LU of the same basis.
CPU only (the oracle needs ~35 min for N = 9653).  usage: python tools/pse_accuracy.py N [position ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from minilp_amd import lpgen
from oracle import minilp_oracle as O

N = int(sys.argv[1])
extra = [int(a) for a in sys.argv[2:]]
lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
m, n = lp["m"], lp["n"]
A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n))
Afull = sp.hstack([A, sp.identity(m, format="csr")], format="csc")
t = time.time()
s = lpgen.build_problem(O.Problem, lp).solve(budget=N)
print("oracle: %d pivots in %.0f s" % (N, time.time() - t), flush=True)
go, nbo, bvo = s.state("primal_edge_sq_norms"), s.state("nb_vars").astype(np.int64), s.state("basic_vars").astype(np.int64)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pse_gpu_%d.npz" % N))
gg, nbg, bvg = g["gamma"], g["nb_vars"], g["basic_vars"]
print("same basis:", bool((bvo == bvg).all()), " same non-basic order:", bool((nbo == nbg).all()))
lu = spl.splu(Afull[:, bvo].tocsc())
def exact(pos):
    col = Afull[:, nbo[pos]].toarray().ravel()
    a = lu.solve(col)
    return 1.0 + float(a @ a)
for pos in extra:
    t_ = exact(pos)
    print("position %d: exact %.9g oracle %.9g (rel %.2e) gpu %.9g (rel %.2e)" % (pos, t_, go[pos], (go[pos] - t_) / t_, gg[pos], (gg[pos] - t_) / t_))
rel_o, rel_g = [], []
for pos in np.random.default_rng(2).choice(n, 150, replace=False):
    t_ = exact(int(pos))
    rel_o.append(abs(go[pos] - t_) / t_); rel_g.append(abs(gg[pos] - t_) / t_)
print("150 random non-basic columns, relative error of the stored weight against 1 + ||B^-1 a_j||^2:")
print("  oracle (reference algorithm): median %.2e  max %.2e" % (np.median(rel_o), max(rel_o)))
print("  GPU engine                  : median %.2e  max %.2e" % (np.median(rel_g), max(rel_g)))
