"""Debug: accuracy of the primal steepest-edge weights and reduced costs after P pivots from the late basis of config 4,
sharded (2 ranks) and unsharded, against exact values from a sparse LU of each run's own basis (scipy on the box)."""
import gzip, os, sys, time
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def check(tag, lp, bv, nbv, gamma, d, xn_unused=None):
    import scipy.sparse as sp, scipy.sparse.linalg as spl
    m, n = lp["m"], lp["n"]
    A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n))
    Afull = sp.hstack([A, sp.identity(m, format="csr")], format="csc")
    c = np.concatenate([-lp["obj"], np.zeros(m)])   # minimised form of Max c'x
    t = time.time()
    lu = spl.splu(Afull[:, bv].tocsc())
    y = lu.solve(c[bv], trans="T")
    rel_g, err_d = [], []
    for pos in np.random.default_rng(2).choice(n, 120, replace=False):
        col = Afull[:, nbv[pos]].toarray().ravel()
        a = lu.solve(col)
        ex = 1.0 + float(a @ a)
        rel_g.append(abs(gamma[pos] - ex) / ex)
        err_d.append(abs(d[pos] - (c[nbv[pos]] - float(col @ y))))
    print("%s: 120 random columns: gamma rel err median %.2e max %.2e | d abs err median %.2e max %.2e  (LU + solves %.0f s)" % (
        tag, np.median(rel_g), max(rel_g), np.median(err_d), max(err_d), time.time() - t), flush=True)

def worker(rank, world, port, pivots, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import minilp_amd as M
    from minilp_amd import dist as md, lpgen
    lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
    p = lpgen.build_problem(M.Problem, lp)
    blob = gzip.open(os.path.join(ROOT, "tests/golden/cfg4_basis_p240000.bin.gz"), "rb").read()
    s = p.solve_from_basis(blob, budget=0)
    box = md.setup_sharding(s, dist)
    dist.barrier()
    s.continue_solve(pivots)
    n = lp["n"]; lo, hi = n * rank // world, n * (rank + 1) // world
    g = s.state("primal_edge_sq_norms"); d = s.state("nb_var_obj_coeffs")
    res = dict(lo=lo, hi=hi, g=g[lo:hi].copy(), d=d[lo:hi].copy(), bv=s.state("basic_vars").astype(np.int64), nbv=s.state("nb_vars").astype(np.int64), obj=s.objective())
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        md.remove_mailbox(box)
        del s
        gam = np.concatenate([x["g"] for x in gathered]); dd = np.concatenate([x["d"] for x in gathered])
        print("sharded objective", gathered[0]["obj"], "ranks agree on the basis:", bool((gathered[0]["bv"] == gathered[1]["bv"]).all()))
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "shard_state_%d.npz" % pivots), bv=gathered[0]["bv"], nbv=gathered[0]["nbv"], gamma=gam, d=dd)
        ref = p.solve_from_basis(blob, budget=0)
        ref.continue_solve(pivots)
        print("unsharded objective", ref.objective())
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "unshard_state_%d.npz" % pivots), bv=ref.state("basic_vars").astype(np.int64),
                            nbv=ref.state("nb_vars").astype(np.int64), gamma=ref.state("primal_edge_sq_norms"), d=ref.state("nb_var_obj_coeffs"))
        out.put(True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "check":   # offline (CPU): python shard_gamma_check.py check FILE.npz
    from minilp_amd import lpgen
    z = np.load(sys.argv[2])
    check(os.path.basename(sys.argv[2]), lpgen.gen_sparse_lp(100000, 100000, 100, 4), z["bv"], z["nbv"], z["gamma"], z["d"])
    sys.exit(0)
if __name__ == "__main__":
    pivots = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29579, pivots, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(1500) for p in procs]
