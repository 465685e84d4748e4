"""Sharded-pricing protocol test: WORLD processes, one per GPU when the box has that many, otherwise all on
GPU 0 (the device mailboxes are mapped across processes through HIP IPC either way, so the exchange protocol
is exercised even on a single-GPU box); gloo for the control plane."""
import os
os.environ.setdefault("MLP_SHARD_DEFER", "0")  # protocol tools: sharded from the first pivot unless the caller asks for the default deferral (engine.h)
import sys
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, m, n, k, pivots, out, family="sparse"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import minilp_amd as M
    from minilp_amd import dist as md
    from minilp_amd import lpgen
    ndev = M.device_count()
    M.set_device(rank if ndev >= world else 0)   # distinct devices whenever the box has them
    if family == "transport":   # network with gains (m supplies, n demands, k arcs per demand node): the compact factor's family
        lp = lpgen.gen_transport_lp(m, n, k, 4, tight=0.5)
    else:
        lp = lpgen.gen_cover_lp(m, n, k, 4) if family == "cover" else lpgen.gen_sparse_lp(m, n, k, 4)
    p = lpgen.build_problem(M.Problem, lp)
    blob = None
    if family.startswith("basis="):   # continue from a committed mid-solve basis (config-4 size: tests/golden/cfg4_basis_p*.bin.gz)
        import gzip
        with gzip.open(family[6:], "rb") as f:
            blob = f.read()
    s = p.solve_from_basis(blob, budget=0, trace=True) if blob else p.solve(budget=0, trace=True)
    box = md.setup_sharding(s, dist)
    dist.barrier()
    t0 = time.time()
    probe = int(os.environ.get("SHARD_TEST_PROBE", "0"))   # pivots after which the "sharding is live" flag is read once (deferred sharding)
    live_probe = None
    if 0 < probe < pivots:
        s.continue_solve(probe)
        live_probe = int(s.state("shard_live")[0])
        if s.budget_exhausted:
            s.continue_solve(pivots - probe)
    else:
        s.continue_solve(pivots)
    live_end = int(s.state("shard_live")[0])
    golive = int(s.state("golive_checks")[0])   # fingerprint comparisons passed at the go-live point of the deferred sharding
    dt = time.time() - t0
    tr = [t[:5] for t in s.trace()]
    transport = s.transport()
    res = dict(rank=rank, n=len(tr), obj=s.objective(), dt=dt, trace=tr, done=not s.budget_exhausted, factor=int(s.stats()["factor_active"]))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        if blob:
            del s   # (its copy of the nucleus inverse)
            ref = p.solve_from_basis(blob, budget=0, trace=True)
            ref.continue_solve(pivots)
        else:
            ref = p.solve(budget=pivots, trace=True)
        rtr = [t[:5] for t in ref.trace()]
        ok = all(g["trace"] == rtr for g in gathered)
        print("sharding live after the probe / at the end:", live_probe, "/", live_end, flush=True)
        print("go-live fingerprint checks passed on rank 0:", golive, flush=True)
        print("transport:", transport, "| devices visible:", ndev, "| factor active on rank 0:", int(ref.stats()["factor_active"]), flush=True)
        print("compact factor active on the sharded ranks:", [g["factor"] for g in gathered], flush=True)
        print("sharded world=%d: pivots=%s obj=%s dt=%s | unsharded pivots=%d obj=%.12g | traces identical: %s" % (
            world, [g["n"] for g in gathered], ["%.12g" % g["obj"] for g in gathered], ["%.3f" % g["dt"] for g in gathered],
            len(rtr), ref.objective(), ok), flush=True)
        if not ok:
            for i, (a, b) in enumerate(zip(gathered[0]["trace"], rtr)):
                if a != b:
                    print("first diff at", i, a, b)
                    break
        md.remove_mailbox(box)
        out.put(ok)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    m, n, k, pivots = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (3000, 3000, 12, 400)))
    family = sys.argv[6] if len(sys.argv) > 6 else "sparse"
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, 29533 + world, m, n, k, pivots, out, family)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1500)
    ok = out.get(timeout=5) if not out.empty() else False
    sys.exit(0 if ok else 1)
