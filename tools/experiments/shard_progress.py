"""Debug: objective progress of a sharded (2 ranks) and the unsharded solve over many pivots from the late basis of config 4.
  python shard_progress.py CHUNK COUNT [nounsharded]"""
import gzip, os, sys, time
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def line(tag, s, t0):
    st = s.stats()
    print("%s %7d pivots  objective %.6f  nucleus %d  reinversions %d  max_pivot_err %.1e  %.0f s" % (
        tag, st["iterations"], s.objective(), st["nucleus_size"], st["reinversions"], st["max_pivot_err"], time.time() - t0), flush=True)

def worker(rank, world, port, chunk, count, unsh, out):
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
    t0 = time.time()
    for i in range(count):
        s.continue_solve(chunk)
        if rank == 0:
            line("sharded  ", s, t0)
    dist.barrier()
    if rank == 0:
        md.remove_mailbox(box)
        del s
        if unsh:
            ref = p.solve_from_basis(blob, budget=0)
            t0 = time.time()
            for i in range(count):
                ref.continue_solve(chunk)
                line("unsharded", ref, t0)
        out.put(True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    chunk, count = int(sys.argv[1]), int(sys.argv[2])
    unsh = not (len(sys.argv) > 3 and sys.argv[3] == "nounsharded")
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29581, chunk, count, unsh, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(3000) for p in procs]
