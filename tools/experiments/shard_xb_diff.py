"""Debug: first pivot at which the ranks of a default (row-sharded stream) 2-rank run hold different x_B, and where."""
import gzip, os, sys
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
STEP, COUNT = int(os.environ.get("CP_STEP", "1")), int(os.environ.get("CP_COUNT", "48"))

def worker(rank, world, port):
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
    for i in range(COUNT):
        s.continue_solve(STEP)
        xb = s.state("basic_var_vals").copy()
        bv = s.state("basic_vars").astype(np.int64)
        g = [None] * world
        dist.all_gather_object(g, xb)
        if rank == 0:
            d = np.abs(g[0] - g[1])
            nz = np.nonzero(d)[0]
            tr = s.trace()[-1]
            line = "pivot %4d (q %d r %d): x_B differs at %d positions" % ((i + 1) * STEP, tr[1], tr[2], len(nz))
            if len(nz):
                struct = bv[nz] < lp["n"]
                line += " (max %.3e at pos %d; %d structural, %d slack; leaving row among them: %s; first: %s)" % (
                    d.max(), int(d.argmax()), int(struct.sum()), int((~struct).sum()), bool(tr[2] in nz), nz[:8].tolist())
            print(line, flush=True)
    if rank == 0:
        md.remove_mailbox(box)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, 29611)) for r in range(2)]
    [p.start() for p in procs]; [p.join(900) for p in procs]
