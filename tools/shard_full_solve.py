"""Config 4 solved to the certified optimum by a SHARDED solve: WORLD ranks (one per GPU when the box has that many, otherwise
all on GPU 0 — the device mailboxes are mapped through HIP IPC either way), column-block pricing, row-sharded streaming pass,
every per-pivot exchange of the protocol over ~1.09 M pivots.  Each rank continues in the same chunks; rank 0 then loads the
final basis into a fresh unsharded engine, which must find it optimal as it stands, and takes the duality certificate there.
  python tools/shard_full_solve.py [world] [rows cols nnz_per_row]"""
import json
import os
os.environ.setdefault("MLP_SHARD_DEFER", "0")  # protocol tools: sharded from the first pivot unless the caller asks for the default deferral (engine.h)
import sys
import time
import types

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, rows, cols, nnz, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import minilp_amd as M
    from minilp_amd import dist as md
    from minilp_amd import lpgen
    ndev = M.device_count()
    M.set_device(rank if ndev >= world else 0)
    lp = lpgen.gen_sparse_lp(rows, cols, nnz, 4)
    p = lpgen.build_problem(M.Problem, lp)
    s = p.solve(budget=0)
    box = md.setup_sharding(s, dist)
    dist.barrier()
    t0 = time.perf_counter()
    chunks = 0
    s.continue_solve(50000)
    while s.budget_exhausted:          # (the pivot sequence is identical on every rank: they all leave the loop together)
        chunks += 1
        if rank == 0 and chunks % 4 == 0:
            print(f"[{time.perf_counter() - t0:7.1f} s] {s.stats()['iterations']} pivots, nucleus {s.stats()['nucleus_size']}", flush=True)
        s.continue_solve(50000)
    wall = time.perf_counter() - t0
    st = s.stats()
    mine = dict(pivots=int(st["iterations"]), objective=s.objective())
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    transport = s.transport()
    blob = s.save_basis(0) if rank == 0 else None   # (mode 0: sets, flags, x_N — what every rank of a sharded solve holds in full)
    if rank == 0:
        md.remove_mailbox(box)
    del s
    dist.barrier()
    if rank == 0:
        # the final basis of the sharded solve in a fresh UNSHARDED engine: it must be optimal as it stands (no further pivot),
        # and the duality certificate is taken there (the reduced costs of a sharded solve live per column block)
        ref = p.solve_from_basis(blob, budget=0)
        ref.continue_solve(1000)
        a = types.SimpleNamespace(chunk=50000, wall_guard=1e9)
        rec = bench.full_solve_live(ref, lp, a, time.perf_counter(), 0.0, 0)
        rec.pop("curve", None)
        out_rec = dict(world=world, devices_visible=ndev, transport=transport, sharded_wall_s=round(wall, 1), all_ranks=gathered,
                       reloaded_unsharded=dict(further_pivots=int(ref.stats()["iterations"]), optimal_as_loaded=not ref.budget_exhausted,
                                               objective=ref.objective(), certificate=rec.get("certificate")))
        print(json.dumps(out_rec), flush=True)
        out.put(bool(out_rec["reloaded_unsharded"]["optimal_as_loaded"]) and out_rec["reloaded_unsharded"]["further_pivots"] == 0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rows, cols, nnz = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (100000, 100000, 100)))
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, 29561 + world, rows, cols, nnz, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(5400)
    ok = out.get(timeout=5) if not out.empty() else False
    sys.exit(0 if ok else 1)
