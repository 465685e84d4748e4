#!/usr/bin/env python3
"""bench.py — simplex pivots/sec on BASELINE.json's headline workload (config 4: random LP,
100 000 vars x 100 000 constraints, 100 nnz/row), fixed-pivot-budget protocol.

A "step" is ONE simplex iteration (pricing -> FTRAN -> ratio test -> BTRAN -> tableau row ->
basis-inverse update -> x_B/d/gamma/beta updates) of the device-resident solver.  W warm-up
pivots from the slack basis, then exactly K timed pivots bracketed by barrier + synchronize.
N > 1: one process per GPU (torch.distributed, RCCL); see DESIGN.md §6 for what is sharded.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--cols", type=int, default=100000)
    ap.add_argument("--nnz-per-row", type=int, default=100)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--cpu-pivots", type=int, default=500, help="bounded CPU-baseline sample (pivots after warm-up; ~20 s of CPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--independent", action="store_true",
                    help="N > 1: one independent LP per rank (weak scaling) instead of column-block sharded pricing of ONE LP")
    return ap.parse_args()


def pmc_traffic(a, which):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes of this workload, corrected as
    MI355X_MICROARCH.md prescribes and as tools/pmc_calib.sh confirms).  PMC counters cannot be
    collected from inside the timed run, so the figure is the committed per-launch average; null when
    the workload differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        doc = json.load(open(path))
    except OSError:
        return None
    w = doc.get("workload", {})
    if (w.get("rows"), w.get("cols"), w.get("nnz_per_row"), w.get("seed")) != (a.rows, a.cols, a.nnz_per_row, a.seed):
        return None
    k = doc.get("kernels", {}).get(which)
    return k["hbm_bytes_per_launch"] if k else None


def cpu_baseline(lp, warmup, sample):
    """The oracle (single-threaded C++ restatement of minilp 0.2.2) timed on this box's host cores:
    same instance, same warm-up, then `sample` timed pivots.  kind = "port" (the Rust reference
    cannot be built here)."""
    from minilp_amd import lpgen
    from oracle import minilp_oracle as O
    s = lpgen.build_problem(O.Problem, lp).solve(budget=warmup)
    it0 = s.stats()
    t0 = time.perf_counter()
    s.continue_solve(sample)
    dt = time.perf_counter() - t0
    it1 = s.stats()
    n = (it1["primal_iters"] + it1["dual_iters"]) - (it0["primal_iters"] + it0["dual_iters"])
    return dict(value=n / dt, unit="pivots/s", cores=1, kind="port",
                sample=f"oracle (C++ restatement of minilp 0.2.2, 1 thread) on the same instance: pivots "
                       f"{warmup}..{warmup + n} from the slack basis in {dt:.2f}s (its fastest stretch; "
                       f"the GPU figure covers pivots {warmup}..{warmup}+steps)",
                host_cpus=os.cpu_count())


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    oversub = world > 1 and torch.cuda.device_count() < world  # test rigs with fewer GPUs than ranks
    dev_index = 0 if oversub else (local_rank if world > 1 else 0)
    if world > 1:
        torch.cuda.set_device(dev_index)
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    red_dev = "cpu" if oversub else "cuda"
    import minilp_amd as M
    from minilp_amd import lpgen
    M.set_device(dev_index)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1 (default): ONE LP, its pricing path (tableau-row sweep K4, d/gamma update K8, pricing scan
    # K1) sharded over disjoint blocks of non-basic positions, candidates exchanged once per pivot
    #  => strong scaling.  --independent: one LP per rank, no data-path exchange => weak.
    from minilp_amd import dist as mdist

    def all_ok(ok):  # every rank must take the same branch
        if world == 1:
            return bool(ok)
        t_ok = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        return int(t_ok.item()) == 1

    def run(sharded):
        lp = lpgen.gen_sparse_lp(a.rows, a.cols, a.nnz_per_row, a.seed + (0 if sharded or world == 1 else rank))
        s = lpgen.build_problem(M.Problem, lp).solve(budget=0, profile=True)
        mailbox = None
        try:
            if sharded:
                mailbox = mdist.setup_sharding(s, dist)
            s.continue_solve(a.warmup)       # W untimed warm-up pivots
            s.reset_stats()
            barrier()
            t0 = time.perf_counter()
            s.continue_solve(a.steps)        # exactly K timed pivots
            barrier()
            return lp, s, mailbox, time.perf_counter() - t0, None
        except Exception as e:               # sharded mode: a failed exchange fails on every rank (bounded waits)
            if not sharded:
                raise
            return lp, s, mailbox, 0.0, e

    sharded = world > 1 and not a.independent
    note = None
    lp, s, mailbox, dt, err = run(sharded)
    if sharded and not all_ok(err is None):
        # the per-pivot exchange could not be set up / timed out on this node: report the same
        # workload as independent replicas (one LP of the family per rank) instead of nothing
        print(f"[rank {rank}] sharded pricing failed ({err}); falling back to independent LPs", file=sys.stderr, flush=True)
        note = f"sharded pricing failed on this node ({err if err else 'on a peer'}); independent LPs reported"
        if mailbox and rank == 0:
            mdist.remove_mailbox(mailbox)
        del s
        sharded = False
        lp, s, mailbox, dt, err = run(False)
    st = s.stats()
    done = st["iterations"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([float(done)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total = float(c.item()) / world if sharded else float(c.item())   # sharded: every rank counts the same pivots
    else:
        total = float(done)
    if rank == 0:
        kern = {}
        for name in ("fused", "sweep"):
            n_l = st[name + "_launches"]
            if n_l:
                us = st[name + "_ms"] * 1e3 / n_l
                gbs = st[name + "_bytes"] / (st[name + "_ms"] * 1e-3) / 1e9
                kern[name] = dict(launches=n_l, avg_us=us, algorithmic_bytes_per_launch=st[name + "_bytes"] / n_l, gbs=gbs,
                                  total_ms=st[name + "_ms"])
        dom = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
        # the pricing path that shards over column blocks = tableau-row sweep + d/gamma update + pricing scan
        pricing_us = None
        if st["update_launches"] and "sweep" in kern:
            pricing_us = kern["sweep"]["avg_us"] + st["update_ms"] * 1e3 / st["update_launches"]
        roofline = None
        if dom:
            kname = {"fused": "k_fused_w (tau=W*rho, v=W^T*t, eta update of the nucleus inverse)",
                     "sweep": ("k_sweep_band (tableau row rho^T N [+ PSE helper]: band-major copy of A, the band of (rho, v) "
                               "held in LDS; the per-band partials are summed in band order by k_update_pivot)"
                               if st.get("banded_sweep") else
                               "k_sweep (tableau row rho^T N [+ PSE helper] as a CSC pull over A)")}[dom]
            roofline = dict(bound="hbm", kernel=kname, achieved=kern[dom]["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=kern[dom]["gbs"] / HBM_PEAK_GBS, traffic=pmc_traffic(a, dom),
                            traffic_unit="HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_traffic.json)",
                            avg_launch_us=kern[dom]["avg_us"], launches=kern[dom]["launches"],
                            algorithmic_bytes_per_launch=kern[dom]["algorithmic_bytes_per_launch"],
                            other_kernels={k: v for k, v in kern.items() if k != dom})
        out = dict(metric="simplex pivots/sec", value=total / dt, unit="pivots/s", n_gpus=world, steps=a.steps,
                   warmup=a.warmup, ms_per_step=dt * 1e3 / max(done, 1), higher_is_better=True,
                   scaling=("strong" if sharded else "weak"),
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=f"config 4: random LP {a.rows} vars x {a.cols} constraints, {a.nnz_per_row} nnz/row "
                                        f"(0.1% fill), Max c'x, Ax<=b, x>=0; primal simplex with PSE+DSE from the slack basis; "
                                        f"timed pivots {a.warmup}..{a.warmup + a.steps}",
                               rows=a.rows, cols=a.cols, nnz=int(st["nnz"]), seed=a.seed,
                               parallelism=("1 GPU" if world == 1 else
                                            (f"{world} GPUs: one LP, pricing path sharded over {world} column blocks, "
                                             f"per-pivot candidate exchange through a host-mapped mailbox; FTRAN/BTRAN/W replicated"
                                             if sharded else f"{world} GPUs, one independent LP of the family per rank")) + (" [oversubscribed test rig: all ranks on one GPU]" if oversub else ""),
                               nucleus_size_at_end=int(st["nucleus_size"]), objective_at_end=s.objective(),
                               completed_steps=int(done), bound_flips=int(st["bound_flips"]),
                               pricing_path_us_per_pivot=pricing_us, note=note),
                   roofline=roofline)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lp, a.warmup, a.cpu_pivots)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if mailbox and rank == 0:
            mdist.remove_mailbox(mailbox)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
