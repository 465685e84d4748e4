"""Replica-exactness gate of the sharded path (tests/test_dist_gpu.py).

WORLD ranks (one process each; distinct GPUs when the box has them, otherwise all on GPU 0) continue config 4 from the
committed late basis (nucleus 20 493) for PIVOTS pivots and stop every STEP pivots.  At every checkpoint each rank reports
SHA-1 digests of what it holds — x_B, the basic / non-basic sets, its block of d and of gamma, an order-independent checksum
of the nucleus inverse and its slot maps — and, at the last one, of the dual steepest-edge weights rebuilt from the
inverse.  Rank 0 then runs the UNSHARDED solve through the same checkpoints.

mode "default"   : row-sharded streaming pass (v_K is summed over the ranks' partials, i.e. in another order than
                   unsharded): the ranks must be BIT-IDENTICAL replicas of the basic side at every checkpoint, and the
                   objective must keep the unsharded run's pace.
mode "replicated": MLP_NO_WSHARD=1, the pulled F product and rho_K in a BTRAN launch on both sides: the sharded arithmetic then equals
                   the unsharded run's operation for operation, so the UNION of the ranks' d / gamma blocks and x_B must
                   equal the unsharded vectors bit for bit as well.

Prints one JSON line; exit code 0 when every assertion of the mode holds."""
import gzip
import hashlib
import json
import os
os.environ.setdefault("MLP_SHARD_DEFER", "0")  # protocol tools: sharded from the first pivot unless the caller asks for the default deferral (engine.h)
import sys
import time

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BASIS = os.path.join(ROOT, "tests", "golden", "cfg4_basis_p240000.bin.gz")


def digest(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def snapshot(s, lo, hi, last):
    d = s.state("nb_var_obj_coeffs")
    g = s.state("primal_edge_sq_norms")
    cs = s.state("w_checksum")
    out = dict(obj=s.objective(), xB=digest(s.state("basic_var_vals")), basic=digest(s.state("basic_vars")),
               nb=digest(s.state("nb_vars")), d=digest(d[lo:hi]), gamma=digest(g[lo:hi]),
               W="%08x%08x" % (int(cs[1]), int(cs[0])), k=int(cs[2]), pivots=int(s.stats()["iterations"]))
    if last:
        out["beta"] = digest(s.state("dual_edge_sq_norms"))
    return out


def worker(rank, world, port, pivots, step, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "replicated":
        os.environ["MLP_NO_WSHARD"] = "1"
        # (round 6: the F product of the FTRAN is PULLED on both sides — a fixed summation order, the same on every rank and in the unsharded
        # run; a sharded solve forms rho_K in a BTRAN launch of its own, so the unsharded side is told to do the same)
        os.environ["MLP_RK_RIDE"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import minilp_amd as M
    from minilp_amd import dist as md, lpgen
    ndev = M.device_count()
    M.set_device(rank if ndev >= world else 0)
    lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
    p = lpgen.build_problem(M.Problem, lp)
    with gzip.open(BASIS, "rb") as f:
        blob = f.read()
    s = p.solve_from_basis(blob, budget=0)
    box = md.setup_sharding(s, dist)
    dist.barrier()
    n = lp["n"]
    lo, hi = md.shard_range(n, rank, world)
    count = pivots // step
    snaps = []
    t0 = time.time()
    for i in range(count):
        s.continue_solve(step)
        snaps.append(snapshot(s, lo, hi, i == count - 1))
    dt = time.time() - t0
    transport = s.transport()
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, lo=lo, hi=hi, snaps=snaps, dt=dt))
    if rank == 0:
        md.remove_mailbox(box)
        del s
        os.environ.pop("MLP_NO_WSHARD", None)
        ref = p.solve_from_basis(blob, budget=0)
        t0 = time.time()
        rsnaps = []
        blocks = []
        for i in range(count):
            ref.continue_solve(step)
            rsnaps.append(snapshot(ref, 0, n, i == count - 1))
            d = ref.state("nb_var_obj_coeffs")
            g = ref.state("primal_edge_sq_norms")
            blocks.append([(digest(d[x["lo"]:x["hi"]]), digest(g[x["lo"]:x["hi"]])) for x in gathered])
        rdt = time.time() - t0
        rec = dict(mode=mode, world=world, pivots=count * step, step=step, transport=transport, devices_visible=ndev,
                   sharded_s=[round(x["dt"], 2) for x in gathered], unsharded_s=round(rdt, 2), checkpoints=[])
        ok_replicas = True
        ok_pace = True
        ok_union = True
        for i in range(count):
            a = gathered[0]["snaps"][i]
            keys = ("xB", "basic", "nb", "W", "k", "pivots") + (("beta",) if "beta" in a else ())
            differing = sorted({key for x in gathered for key in keys if x["snaps"][i][key] != a[key]})
            same = not differing
            r = rsnaps[i]
            pace = abs(a["obj"] - r["obj"]) / max(1.0, abs(r["obj"]))
            sets_equal = a["basic"] == r["basic"] and a["nb"] == r["nb"]
            union = all(x["snaps"][i]["d"] == blocks[i][j][0] and x["snaps"][i]["gamma"] == blocks[i][j][1] for j, x in enumerate(gathered)) \
                and a["xB"] == r["xB"] and a["W"] == r["W"]
            ok_replicas &= same
            ok_pace &= pace <= 1e-9
            ok_union &= union
            rec["checkpoints"].append(dict(pivots=a["pivots"], k=a["k"], objective=a["obj"], unsharded_objective=r["obj"], replicas_identical=same,
                                           objective_rel_diff=pace, same_basis_as_unsharded=sets_equal, bitwise_equal_to_unsharded=union,
                                           replicas_differ_in=differing,
                                           vs_unsharded=dict(xB=a["xB"] == r["xB"], W=a["W"] == r["W"],
                                                             d=[x["snaps"][i]["d"] == blocks[i][j][0] for j, x in enumerate(gathered)],
                                                             gamma=[x["snaps"][i]["gamma"] == blocks[i][j][1] for j, x in enumerate(gathered)])))
        rec["replicas_bit_identical"] = ok_replicas
        rec["objective_pace_within_1e-9"] = ok_pace
        rec["union_bitwise_equal_to_unsharded"] = ok_union
        rec["ok"] = bool(ok_replicas and ok_pace and (ok_union or mode != "replicated"))
        print(json.dumps(rec), flush=True)
        out.put(rec["ok"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    step = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    mode = sys.argv[4] if len(sys.argv) > 4 else "default"
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, 29601 + world, pivots, step, mode, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(2400)
    ok = out.get(timeout=5) if not out.empty() else False
    sys.exit(0 if ok else 1)
