"""Solve config 4 to optimality on the GPU and extract an optimality CERTIFICATE that anyone can check on
a CPU without any LP solver: a primal point x and a dual point y with
    A x <= b, x >= 0        (primal feasible)
    A^T y >= c, y >= 0      (dual feasible)
    c.x == b.y              (no duality gap)
Weak duality then proves both optimal.  y comes from the reduced costs of the non-basic slack columns
(d_slack_i = -pi_i, solver.rs:1199-1231); basic slacks have y_i = 0.  Writes gpurun_out/cfg4_certificate.npz.
usage: [CERTIFY_CHUNK=25000] certify_cfg4.py [rows cols nnz_per_row seed]   (CERTIFY_CHUNK: also write the
wall-time curve of the solve, one point per chunk, to gpurun_out/cfg4_solve_curve.json)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import minilp_amd as M
from minilp_amd import lpgen

m, n, k, seed = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (100000, 100000, 100, 4)))
lp = lpgen.gen_sparse_lp(m, n, k, seed)
os.makedirs("gpurun_out", exist_ok=True)
chunk = int(os.environ.get("CERTIFY_CHUNK", "0"))  # > 0: solve in chunks of that many pivots and record the curve
curve = []
t = time.time()
if chunk > 0:
    s = lpgen.build_problem(M.Problem, lp).solve(budget=chunk)
    while True:
        st = s.stats()
        curve.append((int(st["iterations"]), time.time() - t, int(st["nucleus_size"])))
        with open("gpurun_out/cfg4_solve_progress.log", "a") as f:  # survives a run that is cut short
            f.write("pivots %d wall %.1f k %d obj %.9f\n" % (curve[-1][0], curve[-1][1], curve[-1][2], s.objective()))
        if not s.budget_exhausted:
            break
        s.continue_solve(chunk)
else:
    s = lpgen.build_problem(M.Problem, lp).solve()
wall = time.time() - t
st = s.stats()
x = np.asarray(s.values())
nb_vars = s.state("nb_vars").astype(np.int64)
d = s.state("nb_var_obj_coeffs")
y = np.zeros(m)
slack = nb_vars >= n
y[nb_vars[slack] - n] = d[slack]
A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n))
c, b = lp["obj"], lp["rhs"]
cert = dict(rows=m, cols=n, nnz_per_row=k, seed=seed, pivots=int(st["iterations"]), solve_wall_s=wall,
            objective_accumulated=s.objective(), primal_objective=float(c @ x), dual_objective=float(b @ y),
            max_primal_violation=float(max((A @ x - b).max(), (-x).max(), 0.0)),
            max_dual_violation=float(max((c - A.T @ y).max(), (-y).max(), 0.0)),
            nucleus_size=int(st["nucleus_size"]), max_pivot_err=st["max_pivot_err"])
cert["relative_gap"] = abs(cert["primal_objective"] - cert["dual_objective"]) / max(1.0, abs(cert["primal_objective"]))
print(json.dumps(cert, indent=1), flush=True)
if curve:  # (pivots, wall seconds, nucleus size) at every chunk boundary, and the chunk's microseconds per pivot
    rows, prev = [], (0, 0.0, 0)
    for c3 in curve:
        dp = max(c3[0] - prev[0], 1)
        rows.append(dict(pivots=c3[0], wall_s=round(c3[1], 3), nucleus=c3[2], us_per_pivot=round((c3[1] - prev[1]) / dp * 1e6, 1)))
        prev = c3
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(chunk=chunk, solve_wall_s=wall, curve=rows), open("gpurun_out/cfg4_solve_curve.json", "w"), indent=0)
os.makedirs("gpurun_out", exist_ok=True)
xi = np.nonzero(x)[0]; yi = np.nonzero(y)[0]
np.savez_compressed("gpurun_out/cfg%d_certificate.npz" % (4 if m == 100000 else 0), x_idx=xi.astype(np.int32), x_val=x[xi], y_idx=yi.astype(np.int32), y_val=y[yi],
                    meta=np.array(json.dumps(cert)))
