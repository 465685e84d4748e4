"""Debug: first checkpoint at which a 2-rank sharded run (default: replicated streaming pass, MLP_NO_WSHARD=1, whose arithmetic
should equal the unsharded run's bit for bit) differs from the unsharded run, from the late basis of config 4."""
import gzip, os, sys
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
STEP, COUNT = int(os.environ.get("CP_STEP", "128")), int(os.environ.get("CP_COUNT", "24"))

def snap(s, lo, hi):
    return dict(d=s.state("nb_var_obj_coeffs")[lo:hi].copy(), g=s.state("primal_edge_sq_norms")[lo:hi].copy(), xb=s.state("basic_var_vals").copy(),
                nbv=s.state("nb_vars").copy(), obj=s.objective())

def worker(rank, world, port, out):
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
    snaps = []
    for i in range(COUNT):
        s.continue_solve(STEP)
        snaps.append(snap(s, lo, hi))
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(rank=rank, lo=lo, hi=hi, snaps=snaps))
    if rank == 0:
        md.remove_mailbox(box)
        del s
        for k in ("MLP_NO_WSHARD",):
            os.environ.pop(k, None)
        ref = p.solve_from_basis(blob, budget=0)
        nbv0 = ref.state("nb_vars").copy()
        for i in range(COUNT):
            ref.continue_solve(STEP)
            r = snap(ref, 0, n)
            line = "pivots %5d obj %.9f / %.9f" % ((i + 1) * STEP, gathered[0]["snaps"][i]["obj"], r["obj"])
            for x in gathered:
                sn = x["snaps"][i]; a, b = x["lo"], x["hi"]
                line += " | rank %d: sets %s d %.1e gamma %.1e (rel) x_B %.1e" % (x["rank"], "=" if (sn["nbv"] == r["nbv"]).all() else "DIFFER",
                    np.abs(sn["d"] - r["d"][a:b]).max(), (np.abs(sn["g"] - r["g"][a:b]) / np.maximum(1.0, np.abs(r["g"][a:b]))).max(), np.abs(sn["xb"] - r["xb"]).max())
            print(line, flush=True)
            for x in gathered:
                sn = x["snaps"][i]; a, b = x["lo"], x["hi"]
                dd = np.abs(sn["d"] - r["d"][a:b]); gg = np.abs(sn["g"] - r["g"][a:b]); xx = np.abs(sn["xb"] - r["xb"])
                if dd.max() > 0 or xx.max() > 0:
                    print("      rank %d: d differs at %d positions (max %.2e at %d), gamma differs at %d, x_B differs at %d positions (max %.2e at %d)" % (
                        x["rank"], int((dd > 0).sum()), dd.max(), a + int(dd.argmax()), int((gg > 0).sum()), int((xx > 0).sum()), xx.max(), int(xx.argmax())), flush=True)
            if os.environ.get("BRIEF"):
                continue
            x = gathered[1]; sn = x["snaps"][i]; a, b = x["lo"], x["hi"]
            rel = np.abs(sn["g"] - r["g"][a:b]) / np.maximum(1.0, np.abs(r["g"][a:b]))
            changed = sn["nbv"][a:b] != nbv0[a:b]
            bad = rel > 1e-7
            print("      rank 1: %d positions with gamma rel diff > 1e-7, of them %d hold a variable that entered the non-basic set since the load (%d such positions in the block)" % (
                int(bad.sum()), int((bad & changed).sum()), int(changed.sum())), flush=True)
            for j in np.argsort(-rel)[:4]:
                print("        pos %d var %d (was %d): gamma %.10g / %.10g  d %.6g / %.6g" % (a + j, sn["nbv"][a + j], nbv0[a + j], sn["g"][j], r["g"][a + j], sn["d"][j], r["d"][a + j]), flush=True)
        out.put(True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29587, out)) for r in range(2)]
    [p.start() for p in procs]; [p.join(1500) for p in procs]
