"""Debug: sharded (2 ranks) against unsharded state after P pivots from the late basis of config 4: reduced costs and primal
steepest-edge weights of each rank's own column block against the unsharded run's."""
import gzip, os, sys
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

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
    n = lp["n"]; lo, hi = n * rank // world, n * (rank + 1) // world
    step, count = int(os.environ.get("CP_STEP", "128")), int(os.environ.get("CP_COUNT", "12"))
    for i in range(1, count + 1):
        s.continue_solve(step)
        d = s.state("nb_var_obj_coeffs"); g = s.state("primal_edge_sq_norms")
        blob0 = s.save_basis(0)
        gathered = [None] * world
        dist.all_gather_object(gathered, dict(rank=rank, lo=lo, hi=hi, d=d[lo:hi].copy(), g=g[lo:hi].copy(), obj=s.objective()))
        if rank == 0:
            fresh = p.solve_from_basis(blob0, budget=0)   # reduced costs and weights recomputed from the basis by an unsharded engine
            td = fresh.state("nb_var_obj_coeffs"); tg = fresh.state("primal_edge_sq_norms")
            del fresh
            msg = "pivots %5d objective %.9f:" % (i * step, gathered[0]["obj"])
            for x in gathered:
                e = np.abs(x["d"] - td[x["lo"]:x["hi"]])
                msg += " rank %d: d max err %.2e at %d (median %.1e, |d| median %.2e)" % (x["rank"], e.max(), x["lo"] + int(e.argmax()), float(np.median(e)), float(np.median(np.abs(x["d"]))))
            print(msg, flush=True)
        dist.barrier()
    if rank == 0:
        md.remove_mailbox(box)
        del s
        ref = p.solve_from_basis(blob, budget=0)      # the unsharded solve, same pivot counts, against its own recomputed values
        for i in range(1, (0 if os.environ.get("SHARD_ONLY") else count) + 1):
            ref.continue_solve(step)
            rd = ref.state("nb_var_obj_coeffs")
            fresh = p.solve_from_basis(ref.save_basis(0), budget=0)
            e = np.abs(rd - fresh.state("nb_var_obj_coeffs"))
            del fresh
            print("unsharded pivots %5d objective %.9f: d max err %.2e (median %.1e), |d| median %.2e" % (i * step, ref.objective(), e.max(), float(np.median(e)), float(np.median(np.abs(rd)))), flush=True)
        out.put(True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    pivots = int(sys.argv[1]) if len(sys.argv) > 1 else 2418
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29577, pivots, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(1200) for p in procs]
