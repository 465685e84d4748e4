import os, sys
os.environ.setdefault("MLP_SHARD_DEFER", "0")
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def worker(rank, world, port, pivots):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import minilp_amd as M
    from minilp_amd import dist as md, lpgen
    M.set_device(0)
    lp = lpgen.gen_sparse_lp(4000, 3500, 12, 4)
    p = lpgen.build_problem(M.Problem, lp)
    s = p.solve(budget=0, trace=True)
    box = md.setup_sharding(s, dist)
    dist.barrier()
    for i in range(pivots):
        try:
            s.continue_solve(1)
        except Exception as e:
            print(rank, "EXC at pivot", i, e, "trace", [t[:5] for t in s.trace()][-3:], "fpull", s.state("fpull").tolist(), "k", s.stats()["nucleus_size"], flush=True)
            break
        aq = None
    print(rank, "done", len(s.trace()), [t[:5] for t in s.trace()][:6], flush=True)
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    pivots = int(sys.argv[1])
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, 29571, pivots)) for r in range(2)]
    [p.start() for p in procs]; [p.join(300) for p in procs]
