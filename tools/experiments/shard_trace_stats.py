"""Debug: what the pivots of a sharded (2 ranks) solve look like against the unsharded one, pivots A..B from the late basis."""
import gzip, os, sys
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def describe(tag, s, skip):
    tr = s.trace()[skip:]
    obj = np.array([-t[6] for t in tr])
    dobj = np.diff(obj)
    ev = np.array([t[3] for t in tr]); lv = np.array([t[4] for t in tr])
    pc = np.abs(np.array([t[5] for t in tr]))
    print("%s: |pivot element| p1 %.2e p10 %.2e median %.2e p90 %.2e ; share below 1e-6: %.2f %%, below 1e-4: %.2f %%" % (
        tag, np.percentile(pc, 1), np.percentile(pc, 10), np.median(pc), np.percentile(pc, 90), 100.0 * float((pc < 1e-6).mean()), 100.0 * float((pc < 1e-4).mean())), flush=True)
    q = np.array([t[1] for t in tr])
    print("%s: entering position in block 0: %.1f %%, block 1: %.1f %%" % (tag, 100.0 * float((q < 50000).mean()), 100.0 * float((q >= 50000).mean())), flush=True)
    st = s.stats()
    print("%s: %d pivots | objective %.6f -> %.6f | zero-progress pivots (|dobj| < 1e-9): %.1f %% | median dobj %.3e | entering slack %.1f %% leaving slack %.1f %% | cases %s | flips %d" % (
        tag, len(tr), obj[0], obj[-1], 100.0 * float((np.abs(dobj) < 1e-9).mean()), float(np.median(dobj)), 100.0 * float((ev >= 100000).mean()),
        100.0 * float((lv >= 100000).mean()), st["kase"], st["bound_flips"]), flush=True)

def worker(rank, world, port, a, b, out):
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
    s.continue_solve(b)
    if rank == 0:
        describe("sharded  ", s, a)
    dist.barrier()
    if rank == 0:
        md.remove_mailbox(box)
        del s
        ref = p.solve_from_basis(blob, budget=0, trace=True)
        ref.continue_solve(b)
        describe("unsharded", ref, a)
        out.put(True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29583, a, b, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(2000) for p in procs]
