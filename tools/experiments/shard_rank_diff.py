"""Debug: per-iteration quantities of the two ranks of a sharded solve compared (MLP_KPROF=2..5 puts one into the records)."""
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
    s = p.solve_from_basis(blob, budget=0, trace=True)
    box = md.setup_sharding(s, dist)
    dist.barrier()
    s.continue_solve(pivots)
    vals = np.array([t[5] for t in s.trace()])
    gathered = [None] * world
    dist.all_gather_object(gathered, vals)
    if rank == 0:
        md.remove_mailbox(box)
        a, b = gathered[0], gathered[1]
        rel = np.abs(a - b) / np.maximum(1e-300, np.abs(a))
        bad = np.nonzero(rel > 1e-9)[0]
        print("MLP_KPROF=%s: %d pivots, rank 0 vs rank 1: max rel diff %.2e, %d pivots differ by > 1e-9, first at %s" % (
            os.environ.get("MLP_KPROF"), len(a), rel.max(), len(bad), bad[:8].tolist()), flush=True)
        for i in bad[:6]:
            print("   pivot %d: %.15g / %.15g" % (i, a[i], b[i]), flush=True)
        out.put(True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    pivots = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29589, pivots, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(1200) for p in procs]
