#!/usr/bin/env python3
"""bench.py — simplex pivots/sec on BASELINE.json's headline workload (config 4: random LP,
100 000 vars x 100 000 constraints, 100 nnz/row), fixed-pivot-budget protocol.

A "step" is ONE simplex iteration (pricing -> FTRAN -> ratio test -> BTRAN -> tableau row ->
basis-inverse update -> x_B/d/gamma/beta updates) of the device-resident solver.

`value` (the driver-timed figure): W warm-up pivots from the slack basis, then exactly K timed pivots
bracketed by barrier + synchronize.  That window is the cheapest stretch of a 1.09 M-pivot solve, so the
same JSON line also carries (N = 1, rank 0, outside the timed region; every object says whether it was measured
LIVE by this run or read from a COMMITTED file):
  * `windows.mid` / `windows.late` (live): a fixed number of pivots from the two committed MID-SOLVE bases of the same
    instance (tests/golden/cfg4_basis_p*.bin.gz, nucleus ~10 000 and ~20 000: where the solve spends its time),
    each with pivots/s, us per pivot and per-kernel bytes / GB/s / fraction of the HBM roofline;
  * `roofline.ftran` (live): the column FTRAN alpha_q = B^-1 a_q in us and bytes touched (a latency-bound gather of a few
    columns of the explicit nucleus inverse: a bandwidth fraction is the wrong yardstick for it), and the one
    FTRAN-shaped solve that IS a stream: the dense-rhs x_B = B^-1 (b - N x_N) of the polish step
    (mlp_solution_recompute_basic_values), one read of the nucleus inverse through k_stream_w's tau side;
  * `full_solve` (live): BASELINE.json's "total solve wall-time" — the timed solve is continued to optimality in chunks
    of 50 000 pivots under a wall guard, and the optimum is certified on the box by weak duality (scipy mat-vecs);
  * `cpu_baseline` (live): the oracle timed on the host cores over the SAME pivots as `value` (and over pivots
    200..700, its fastest sustained stretch).
N > 1: one process per GPU (torch.distributed, RCCL); ONE LP, pricing path sharded; a failure of
the sharded set-up is an error, never a silent fall-back to replicas.

Prints ONE JSON line on rank 0.
"""
import argparse
import datetime
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MID_BASIS = os.path.join(ROOT, "tests", "golden", "cfg4_basis_p45000.bin.gz")
LATE_BASIS = os.path.join(ROOT, "tests", "golden", "cfg4_basis_p240000.bin.gz")


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
    ap.add_argument("--no-windows", action="store_true", help="skip the mid / late windows from the saved bases")
    ap.add_argument("--window-steps", type=int, nargs=2, default=[512, 256], metavar=("MID", "LATE"))
    ap.add_argument("--samples", type=int, default=32, help="pivots of the event-bracketed sampling pass after a timed region")
    ap.add_argument("--no-full-solve", action="store_true", help="do not continue the timed solve to optimality (total solve wall time)")
    ap.add_argument("--wall-guard", type=float, default=1300.0,
                    help="full solve: stop (complete = false) once the whole bench run has lasted this many seconds")
    ap.add_argument("--chunk", type=int, default=50000, help="full solve: pivots per continue call")
    ap.add_argument("--no-factor-transport", action="store_true", help="skip the 200 000-row transport solve on the compact factor (row f3)")
    ap.add_argument("--independent", action="store_true",
                    help="N > 1: one independent LP per rank (weak scaling) instead of column-block sharded pricing of ONE LP")
    return ap.parse_args()


def pmc_traffic(a, which):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes of this workload, corrected as
    MI355X_MICROARCH.md prescribes and as tools/pmc_calib.sh confirms).  PMC counters cannot be
    collected from inside the timed run, so the figure is the committed per-launch average; null when
    the workload differs from the profiled one."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        w = doc.get("workload", {})
        if (w.get("rows"), w.get("cols"), w.get("nnz_per_row"), w.get("seed")) != (a.rows, a.cols, a.nnz_per_row, a.seed):
            return None, None
        k = doc.get("kernels", {}).get(which)
        # (launches that exit at once — the streaming pass of a folding pivot is skipped — are left out of the average when the file says so)
        return (k.get("hbm_bytes_per_working_launch", k["hbm_bytes_per_launch"]), name) if k else (None, None)
    return None, None


def cpu_baseline(lp, warmup, steps, cap):
    """One core, pinned (minilp is single-threaded: SURVEY §8d); the affinity mask is restored afterwards."""
    old_mask = pinned = None
    try:
        old_mask = os.sched_getaffinity(0)
        pinned = max(old_mask)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        old_mask = pinned = None
    try:
        return cpu_baseline_pinned(lp, warmup, steps, cap, pinned)
    finally:
        if old_mask is not None:
            try:
                os.sched_setaffinity(0, old_mask)
            except OSError:
                pass


def cpu_baseline_pinned(lp, warmup, steps, cap, pinned):
    """The oracle (single-threaded C++ restatement of minilp 0.2.2) timed on this box's host cores on the same
    instance.  `value`: exactly the pivots the GPU figure was timed on, warmup..warmup+steps (bounded by `cap` pivots so
    that a long timed region costs ~20 s of CPU at most); `window_200_700`: pivots 200..700, its fastest sustained
    stretch (it slows to ~7 pivots/s by pivot 2 500).  kind = "port" (the Rust reference cannot be built here)."""
    from minilp_amd import lpgen
    from oracle import minilp_oracle as O

    def count(st):
        return st["primal_iters"] + st["dual_iters"]
    s0 = lpgen.build_problem(O.Problem, lp).solve(budget=warmup)
    n_same = min(steps, cap)
    # the same-window sample is short when the window is (20 pivots ~ 14 ms): timed REPEATS times on clones of the
    # warm state, median reported (a single read of a 14 ms interval is noise)
    reps = 7 if n_same <= 200 else 1
    samples = []
    for i in range(reps):
        s = s0.clone() if i + 1 < reps else s0
        it0 = count(s.stats())
        t0 = time.perf_counter()
        s.continue_solve(n_same)
        samples.append((time.perf_counter() - t0, count(s.stats()) - it0))
        if i + 1 < reps:
            del s
    samples_sorted = sorted(samples, key=lambda x: x[0] / max(x[1], 1))
    dt, n = samples_sorted[len(samples_sorted) // 2]
    out = dict(value=n / dt, unit="pivots/s", cores=1, kind="port", measured="live",
               sample=f"oracle (C++ restatement of minilp 0.2.2, 1 thread pinned to core {pinned}), same instance, pivots {warmup}..{warmup + n} "
                      f"from the slack basis: median of {reps} repeats, {dt * 1e3:.1f} ms" + ("" if n_same == steps else f" (the first {n_same} of the {steps} timed pivots)"),
               host_cpus=os.cpu_count(), cpu_model=cpu_model(), pinned_core=pinned, repeats=reps,
               repeat_pivots_per_s=[round(n_ / t_, 1) for t_, n_ in samples])
    done = warmup + n
    if done <= 200:   # the 200..700 figure quoted in DESIGN.md, separately
        s.continue_solve(200 - done)
        it0 = count(s.stats())
        t0 = time.perf_counter()
        s.continue_solve(500)
        dt2 = time.perf_counter() - t0
        n2 = count(s.stats()) - it0
        out["window_200_700"] = dict(value=n2 / dt2, pivots=int(n2), seconds=dt2)
    return out


def kernel_report(st):
    """Per-kernel averages of the event-bracketed (sampled) iterations: algorithmic bytes, GB/s, roofline fraction."""
    out = {}
    for name in ("fused", "sweep", "ftran", "fold"):
        n_l = st[name + "_launches"]
        if n_l and st[name + "_ms"] > 0:
            gbs = st[name + "_bytes"] / (st[name + "_ms"] * 1e-3) / 1e9
            out[name] = dict(launches=int(n_l), avg_us=st[name + "_ms"] * 1e3 / n_l,
                             algorithmic_bytes_per_launch=st[name + "_bytes"] / n_l, gbs=gbs, frac=gbs / HBM_PEAK_GBS,
                             total_ms=st[name + "_ms"])
    if st.get("str_launches"):
        out["row_sparse"] = dict(launches=int(st["str_launches"]), avg_us=st["str_ms"] * 1e3 / st["str_launches"])
    if st["update_launches"]:
        out["update"] = dict(launches=int(st["update_launches"]), avg_us=st["update_ms"] * 1e3 / st["update_launches"])
    if st["iter_samples"]:
        out["iteration"] = dict(samples=int(st["iter_samples"]), avg_us=st["iter_ms"] * 1e3 / st["iter_samples"])
    return out


KERNEL_NAMES = {
    "fused": ("pass over the nucleus inverse of a primal pivot: v_K = W^T t_K, the BTRAN-shaped dense solve v = B^-T alpha_q of primal "
              "steepest edge (solver.rs:1114).  Nucleus of a few dozen columns: k_small_basis (BTRAN, the pass with the eta update, the v "
              "tail and the touched-column list in one launch); small nucleus: k_fused_w (one read + one write, the eta update rides along); large "
              "nucleus: k_stream_w (read-only).  tau = B^-1 rho (solver.rs:1157) is skipped in primal pivots (lazy dual steepest edge)"),
    "primal_head": ("k_primal_head (small nucleus: FTRAN + Harris test + BTRAN + v = B^-T alpha_q with the eta update of W + touched columns + "
                    "partition change + basic side of the pivot in ONE workgroup; a latency chain, not a stream: the byte figure of this entry is "
                    "the 16 k^2 of the W walk it contains)"),
    "fold": "k_fold_w (fold of the pending rank-1 terms into the nucleus inverse, every 32 pivots: read + write)",
    "dense_ftran": ("dense-rhs FTRAN x_B = B^-1 (b - N x_N) of the polish step (solver.rs:1177-1197): one streaming read of the nucleus "
                    "inverse through k_stream_w's tau side, x_K = W r_K"),
    "sweep_band": ("k_sweep_band (tableau row rho^T N [+ PSE helper]: band-major copy of A, the band of (rho, v) held in "
                   "LDS; per-band partials summed in band order by k_update_pivot)"),
    "sweep": "k_sweep (tableau row rho^T N [+ PSE helper] as a CSC pull over A)",
    "ftran": ("alpha_q = B^-1 a_q: the listed columns of the nucleus inverse (k_ftran_gather_lrh, head inside the gather) + the F product of the "
              "singleton rows — round 6: PULLED per row from the packed copy of the nucleus columns (k_fpull_p1, which also runs Harris pass 1) — in "
              "the large-nucleus windows; in the driver-timed window the FTRAN is the first stages of k_primal_head and this entry times the whole head"),
}


def window_from_basis(M, prob, path, warm, steps, samples, shard=None, dense_ftran=True):
    """`steps` timed pivots from a committed mid-solve basis (after `warm` untimed ones), then an event-bracketed
    sampling pass of `samples` pivots for the per-kernel figures.  shard = (dist, mdist, barrier, max_over_ranks): every
    rank loads the same basis and the solve continues SHARDED (pricing path over column blocks, streaming pass of the
    nucleus inverse over row strips); the time is the maximum over the ranks."""
    import torch
    with gzip.open(path, "rb") as f:
        blob = f.read()
    t0 = time.perf_counter()
    s = prob.solve_from_basis(blob, budget=0, profile=True)   # device re-inversion + x_B, d recomputed from the basis
    load_s = time.perf_counter() - t0
    k0 = int(s.stats()["nucleus_size"])
    mailbox = None
    err = None
    if shard:
        mailbox = shard[1].setup_sharding(s, shard[0])   # raises on every rank if any rank cannot join

    def guarded(budget):  # a failed exchange is a bounded wait on every rank; the barriers below must still be reached
        nonlocal err
        try:
            if err is None:
                s.continue_solve(budget)
        except Exception as e:
            err = str(e)
    guarded(warm)
    s.reset_stats()
    s.set_sampling(None)   # (no instrumented iteration inside the timed pivots; the sampling pass follows)
    if shard:
        shard[2]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    guarded(steps)
    if shard:
        shard[2]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if shard:
        dt = shard[3](dt)
        if shard[3](1.0 if err else 0.0) > 0.0:   # any rank failed: report it, every rank leaves the same way
            shard[2]()
            if mailbox and shard[0].get_rank() == 0:
                shard[1].remove_mailbox(mailbox)
            del s
            return dict(error=err or "a peer rank failed", basis=os.path.basename(path))
    st = s.stats()
    done = int(st["iterations"])
    s.set_sampling(True)
    s.continue_solve(samples)
    st2 = s.stats()
    kern = kernel_report(st2)
    k1 = int(st2["nucleus_size"])
    if not shard and dense_ftran:
        # the FTRAN-shaped stream: x_B = B^-1 (b - N x_N), three dense-rhs solves (the solve + two refinement steps)
        s.recompute_basic_values()
        st3 = s.stats()
        if st3["dense_ftran_launches"]:
            ms, nl, by = st3["dense_ftran_ms"], st3["dense_ftran_launches"], st3["dense_ftran_bytes"]
            gbs = by / (ms * 1e-3) / 1e9
            kern["dense_ftran"] = dict(launches=int(nl), avg_us=ms * 1e3 / nl, algorithmic_bytes_per_launch=by / nl, gbs=gbs,
                                       frac=gbs / HBM_PEAK_GBS, total_ms=ms)
    out = dict(measured="live", basis=os.path.basename(path), nucleus_size_at_start=k0, nucleus_size_at_end=k1, warmup=warm, steps=done,
               pivots_per_s=done / dt, us_per_pivot=dt * 1e6 / max(done, 1), load_basis_s=load_s,
               objective_at_end=s.objective(), max_pivot_err=st2["max_pivot_err"], kernels=kern,
               sampling=f"{samples}-pivot event-bracketed pass right after the timed pivots (eager launches, "
                        f"{int(st2['iter_samples'])} iterations sampled); the timed pivots run as graph replays")
    if shard:
        out["transport"] = s.transport()
        shard[2]()
        if mailbox and shard[0].get_rank() == 0:
            shard[1].remove_mailbox(mailbox)
    del s
    return out


def full_solve_live(s, lp, a, t_start, spent_s, pivots_done):
    """BASELINE.json's `total solve wall-time`, measured live: the solve that was just timed (slack basis, warm-up, K timed
    pivots, sampling pass: `spent_s` seconds, `pivots_done` pivots) is continued to optimality in chunks of a.chunk pivots.
    A wall guard stops it (complete = false) once the whole bench run has lasted a.wall_guard seconds.  When it completes,
    the optimum is certified on the box without a solver: x and the dual point y read off the reduced costs of the
    non-basic slacks must satisfy A x <= b, x >= 0, A^T y >= c, y >= 0 and c.x = b.y (weak duality)."""
    import numpy as np
    import scipy.sparse as sp
    wall = spent_s
    curve = []
    complete = False
    error = None
    try:
        while True:
            if not s.budget_exhausted:
                complete = True
                break
            if time.perf_counter() - t_start > a.wall_guard:
                break
            t0 = time.perf_counter()
            s.continue_solve(a.chunk)
            wall += time.perf_counter() - t0
            st = s.stats()
            curve.append((pivots_done + int(st["iterations"]), round(wall, 2), int(st["nucleus_size"])))
    except Exception as e:  # a failed solve is reported, it does not void the timed figure
        error = str(e)
    st = s.stats()
    pivots = pivots_done + int(st["iterations"])
    out = dict(measured="live", complete=bool(complete), pivots=pivots, total_solve_wall_s=wall, avg_pivots_per_s=pivots / max(wall, 1e-9),
               objective=s.objective(), nucleus_size_at_end=int(st["nucleus_size"]), max_pivot_err=st["max_pivot_err"],
               final_refreshes=int(st["final_refreshes"]), reinversions=int(st["reinversions"]), wall_guard_s=a.wall_guard, chunk=a.chunk,
               wall_definition="seconds inside mlp_problem_solve_ex / mlp_solution_continue from the slack basis (set-up of the device "
                               "state included, problem assembly through add_var / add_constraint excluded), incl. the event-bracketed sampling pass",
               curve=[dict(pivots=p, wall_s=w, nucleus=k) for p, w, k in curve])
    if error:
        out["error"] = error[:300]
    if complete and not error:
        m, n = lp["m"], lp["n"]
        x = np.asarray(s.values())
        nb_vars = s.state("nb_vars").astype(np.int64)
        d = s.state("nb_var_obj_coeffs")
        y = np.zeros(m)
        slack = nb_vars >= n
        y[nb_vars[slack] - n] = d[slack]   # d_slack_i = -pi_i for the minimised form of Max c'x (solver.rs:1199-1231)
        A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n))
        c, b = lp["obj"], lp["rhs"]
        po, do = float(c @ x), float(b @ y)
        out["certificate"] = dict(primal_objective=po, dual_objective=do, relative_gap=abs(po - do) / max(1.0, abs(po)),
                                  max_primal_violation=float(max((A @ x - b).max(), (-x).max(), 0.0)),
                                  max_dual_violation=float(max((c - A.T @ y).max(), (-y).max(), 0.0)),
                                  checked="on the box with scipy mat-vecs (weak duality), no solver")
    return out


def factor_transport(a):
    """SURVEY §8 row f3, measured live: the 200 000-row network-with-gains instance (lpgen.gen_transport_lp(100000, 100000, 4, tight=0.4):
    m = 198 079, n = 400 000) solved to optimality on the COMPACT FACTOR of the basis (auto-selected when the explicit nucleus
    inverse would pass 8 192 slots), with the oracle timed on the same box over the first 20 000 pivots (its whole solve, 92 s on
    this box class, is the committed figure of profiles/r04_transport_200k_evidence.json)."""
    import minilp_amd as M
    from minilp_amd import lpgen
    from oracle import minilp_oracle as O
    lp = lpgen.gen_transport_lp(100000, 100000, 4, tight=0.4)
    out = dict(measured="live", family="network with gains (lpgen.gen_transport_lp 100000 x 100000, 4 arcs per demand node, tight 0.4)",
               rows=int(lp["m"]), cols=int(lp["n"]), nnz=int(len(lp["data"])))
    prob = lpgen.build_problem(M.Problem, lp)
    t0 = time.perf_counter()
    s = prob.solve(budget=0)
    s.continue_solve(20000)
    t_first = time.perf_counter() - t0
    s.continue_solve(-1)
    wall = time.perf_counter() - t0
    st = s.stats()
    out.update(gpu_wall_s=wall, pivots=int(st["iterations"]), gpu_us_per_pivot=wall * 1e6 / max(1, int(st["iterations"])),
               gpu_first_20000_us_per_pivot=t_first * 1e6 / 20000, objective=s.objective(), factor_active=int(st["factor_active"]),
               levels_at_end=int(st["factor_levels"]), refactorisations=int(st["factor_refactors"]), largest_bump=int(st["factor_bump_max"]),
               max_pivot_err=float(st["max_pivot_err"]))
    del s
    if not a.no_cpu_baseline:
        so = lpgen.build_problem(O.Problem, lp).solve(budget=0)
        t0 = time.perf_counter()
        so.continue_solve(20000)
        dt = time.perf_counter() - t0
        out["oracle_first_20000_us_per_pivot"] = dt * 1e6 / 20000
        out["gpu_over_oracle_first_20000"] = out["oracle_first_20000_us_per_pivot"] / out["gpu_first_20000_us_per_pivot"]
        del so
    try:
        ev = json.load(open(os.path.join(ROOT, "profiles", "r04_transport_200k_evidence.json")))
        orc, den = ev["runs"].get("oracle", {}), ev["runs"].get("dense", {})
        out["committed"] = dict(source="profiles/r04_transport_200k_evidence.json", oracle_full_solve_s=orc.get("wall_s"), oracle_pivots=orc.get("pivots"),
                                oracle_objective=orc.get("objective"),
                                explicit_inverse_us_per_pivot_at_60000=(den.get("chunks") or [{}])[-1].get("us_per_pivot"))
        if orc.get("wall_s"):
            out["gpu_over_oracle_full_solve"] = orc["wall_s"] / wall
            out["objective_matches_oracle"] = bool(abs(orc["objective"] - out["objective"]) <= 1e-9 * abs(orc["objective"]))
    except (OSError, KeyError, ValueError):
        pass
    try:  # (committed, not re-run here: the oracle alone needs two minutes on it) the bump of the factor as a sparse LU with fill, DESIGN 2.8
        ev = json.load(open(os.path.join(ROOT, "profiles", "r05_mixed100k_sparse_bump_evidence.json")))
        f, d, o = (ev["runs"].get(k_, {}) for k_ in ("factor", "factor_dense_bump", "oracle"))
        out["sparse_bump_committed"] = dict(
            source="profiles/r05_mixed100k_sparse_bump_evidence.json", family="config-3 generator at 100000 x 160000",
            gpu_wall_s=f.get("wall_s"), pivots=f.get("pivots"), largest_bump=max((c_.get("bump_max", 0) for c_ in f.get("chunks", [])), default=None),
            oracle_wall_s=o.get("wall_s"), gpu_over_oracle=(o.get("wall_s") / f["wall_s"]) if f.get("wall_s") and o.get("wall_s") else None,
            dense_bump_carrier=dict(finished=d.get("finished"), pivots=d.get("pivots"), wall_s=d.get("wall_s"),
                                    last_us_per_pivot=(d.get("chunks") or [{}])[-1].get("us_per_pivot")))
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return out


def compact_line(out):
    """The ONE JSON line printed on stdout: every contract field, numbers rounded, free text kept short (the driver
    reads the tail of stdout: the round-1 line was 2.7 KB).  The full record (kernel descriptions, sampling notes, every
    per-kernel figure) is written next to it as gpurun_out/bench_detail_n<N>.json."""
    def r(x, n=4):
        return round(x, n) if isinstance(x, float) else x

    NAMES = {"fused": "w_pass_v"}  # the pass over the nucleus inverse of a primal pivot computes v = B^-T alpha_q (BTRAN-shaped)

    def kern(k):  # per-kernel digest: average launch time, GB/s, fraction of the HBM peak
        return {NAMES.get(n, n): {"us": r(v.get("avg_us"), 1), "gbs": r(v.get("gbs"), 0), "frac": r(v.get("frac"), 3)} if "gbs" in v
                else {"us": r(v.get("avg_us"), 1)} for n, v in k.items()}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["value"] = r(line["value"], 2)
    line["ms_per_step"] = r(line["ms_per_step"], 5)
    c = out["config"]
    line["config"] = dict(workload=c["workload"], parallelism=c["parallelism"], nucleus_size_at_end=c["nucleus_size_at_end"],
                          completed_steps=c["completed_steps"], pricing_path_us_per_pivot=r(c["pricing_path_us_per_pivot"], 1))
    rf = out.get("roofline")
    if rf:
        line["roofline"] = dict(bound=rf["bound"], kernel=rf["kernel"].split(" (")[0], achieved=r(rf["achieved"], 1), peak=rf["peak"],
                                unit=rf["unit"], frac=r(rf["frac"], 4), traffic=r(rf["traffic"], 0) if rf["traffic"] else None,
                                avg_launch_us=r(rf["avg_launch_us"], 2), launches=rf["launches"],
                                algorithmic_bytes_per_launch=r(rf["algorithmic_bytes_per_launch"], 0))
        line["roofline"]["measured"] = dict(achieved="live", traffic=rf.get("traffic_measured"))
        if rf.get("pivot_level"):
            pl = rf["pivot_level"]
            line["roofline"]["pivot_level"] = dict(bytes_per_pivot=r(pl["bytes_per_pivot"], 0), us_per_pivot=r(pl["us_per_pivot"], 1),
                                                   achieved=r(pl["achieved"], 1), frac=r(pl["frac"], 4))
        if rf.get("window"):
            line["roofline"]["window"] = rf["window"][:120]
        tw = rf.get("timed_window")
        if tw:  # the driver-timed window's own largest kernel (no bandwidth-bound kernel there: a latency chain)
            line["roofline"]["timed_window"] = dict(kernel=tw["kernel"].split(" (")[0][:40], us=r(tw["avg_launch_us"], 1), frac=r(tw["frac"], 4))
        ft = rf.get("ftran")
        if ft:  # north_star's FTRAN entries: us and bytes touched of the column FTRAN per window; the dense-rhs FTRAN as a stream
            def colft(x):
                return dict(us=r(x["avg_us"], 1), bytes=r(x["algorithmic_bytes_per_launch"], 0)) if x and "avg_us" in x else None
            line["roofline"]["ftran"] = dict(column={w: colft(x) for w, x in ft.get("column", {}).items()})
            if ft.get("dense_rhs_late"):
                line["roofline"]["ftran"]["dense_rhs_late"] = kern({"x": ft["dense_rhs_late"]})["x"]
    w = out.get("windows")
    if w:
        line["windows"] = {}
        for name, x in w.items():
            if not x:
                line["windows"][name] = None
            elif "error" in x:
                line["windows"][name] = dict(error=x["error"][:160])
            else:
                line["windows"][name] = dict(measured=x.get("measured", "live"), k=x["nucleus_size_at_start"], steps=x["steps"], pivots_per_s=r(x["pivots_per_s"], 1),
                                             us_per_pivot=r(x["us_per_pivot"], 1), kernels=kern(x["kernels"]))
    fs = out.get("full_solve")
    if fs:
        line["full_solve"] = dict(measured=fs["measured"], complete=fs["complete"], total_solve_wall_s=r(fs["total_solve_wall_s"], 1),
                                  pivots=fs["pivots"], avg_pivots_per_s=r(fs["avg_pivots_per_s"], 1), objective=r(fs["objective"], 6))
        if fs.get("certificate"):
            ce = fs["certificate"]
            line["full_solve"]["certificate"] = dict(relative_gap=float("%.1e" % ce["relative_gap"]),
                                                     primal_violation=float("%.1e" % ce["max_primal_violation"]),
                                                     dual_violation=float("%.1e" % ce["max_dual_violation"]))
        if fs.get("error"):
            line["full_solve"]["error"] = fs["error"][:160]
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(value=r(cb["value"], 2), unit=cb["unit"], cores=cb["cores"], kind=cb["kind"], measured=cb.get("measured", "live"),
                                    sample=cb["sample"][:240], host_cpus=cb.get("host_cpus"), cpu_model=cb.get("cpu_model"),
                                    repeats=cb.get("repeats"))
        if cb.get("window_200_700"):
            line["cpu_baseline"]["pivots_200_700"] = r(cb["window_200_700"]["value"], 2)
        if cb.get("gpu_over_cpu_same_window"):
            line["cpu_baseline"]["gpu_over_cpu_same_window"] = r(cb["gpu_over_cpu_same_window"], 2)
    if out.get("parity"):
        line["parity"] = dict(oracle_identical_through_pivot=out["parity"]["oracle_identical_through_pivot"], beyond="defining equations vs A at k = 9 999 / 20 493 + live duality certificate")
    for key in ("ranks", "value_vs_1gpu", "pricing_speedup_vs_1gpu", "late_sharded", "unsharded_same_run"):
        if out.get(key) is not None:
            line[key] = ({k_: r(v_, 1) for k_, v_ in out[key].items()} if key == "late_sharded" else out[key]) if isinstance(out[key], dict) else r(out[key], 3)
    if out.get("factor_transport"):
        ft_ = out["factor_transport"]
        line["factor_transport"] = {k_: (r(v_, 2) if isinstance(v_, float) else v_) for k_, v_ in ft_.items() if k_ not in ("chunks", "note")}
        sbc = line["factor_transport"].get("sparse_bump_committed")
        if isinstance(sbc, dict):  # (the detail file keeps the whole record)
            line["factor_transport"]["sparse_bump_committed"] = {k_: (r(v_, 2) if isinstance(v_, float) else v_) for k_, v_ in sbc.items()
                                                                 if k_ in ("source", "gpu_wall_s", "oracle_wall_s", "gpu_over_oracle", "largest_bump")}
    return line


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks here, one process per
    GPU, exactly as the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` would, hand their stdout
    through (rank 0 prints the ONE line) and leave with the launcher's return code.  Fewer visible GPUs than ranks is an ERROR
    (rc 2) unless MLP_OVERSUBSCRIBE=1 (test rigs: every rank on the one device)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and os.environ.get("MLP_OVERSUBSCRIBE") != "1":
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but {ndev} GPU(s) visible; refusing to run {a.gpus} ranks on fewer devices "
                         f"(set MLP_OVERSUBSCRIBE=1 for an oversubscribed test rig)\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MLP_BENCH_SELF_LAUNCHED="1")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    T_START = time.perf_counter()
    a = parse()
    if a.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:   # a world that is not what --gpus says is a mis-launch, not something to measure under another name
        if int(os.environ.get("RANK", "0")) == 0:
            sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The contract is ONE JSON line on stdout.  Native libraries (Gloo's "Rank 0 is connected ...", RCCL warnings, the HIP
    # runtime) write to file descriptor 1 directly, so fd 1 is pointed at stderr for the whole run and the line goes to
    # a private copy of the original stdout.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the peer mailboxes need it on this driver
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    oversub = world > 1 and torch.cuda.device_count() < world  # test rigs with fewer GPUs than ranks
    if oversub and os.environ.get("MLP_OVERSUBSCRIBE") != "1":
        if rank == 0:
            sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible; refusing (MLP_OVERSUBSCRIBE=1 allows "
                             f"an oversubscribed test rig)\n")
        sys.exit(2)
    dev_index = 0 if oversub else (local_rank if world > 1 else 0)
    if world > 1:
        torch.cuda.set_device(dev_index)
        if oversub:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=datetime.timedelta(seconds=600))
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
    sharded = world > 1 and not a.independent
    lp = lpgen.gen_sparse_lp(a.rows, a.cols, a.nnz_per_row, a.seed + (0 if sharded or world == 1 else rank))
    prob = lpgen.build_problem(M.Problem, lp)
    solve_s = 0.0   # seconds inside solve / continue calls of THIS solve (total solve wall time, N = 1)
    t0 = time.perf_counter()
    s = prob.solve(budget=0, profile=True)
    solve_s += time.perf_counter() - t0
    mailbox = None
    if sharded:
        # raises on EVERY rank when any rank cannot join (setup_sharding all-gathers the errors): a sharded run
        # that cannot be set up is a failed run, not a run of something else
        mailbox = mdist.setup_sharding(s, dist)
    ranks_info = None
    if world > 1:
        # what actually ran: every rank reports itself (rank, device index, device uuid-ish name, pid); rank 0 puts it into the line
        mine = dict(rank=rank, device=dev_index, device_name=torch.cuda.get_device_name(dev_index), pid=os.getpid(),
                    transport=(s.transport() if sharded else "none"))
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ranks_info = dict(ranks_seen=len({r_["rank"] for r_ in allr}), devices=[r_["device"] for r_ in allr],
                          distinct_devices=len({r_["device"] for r_ in allr}), pids_distinct=len({r_["pid"] for r_ in allr}) == world,
                          transport=sorted({r_["transport"] for r_ in allr}),
                          process_group=dict(backend=dist.get_backend(), size=dist.get_world_size()),
                          self_launched=os.environ.get("MLP_BENCH_SELF_LAUNCHED") == "1", oversubscribed=bool(oversub))
    t0 = time.perf_counter()
    s.continue_solve(a.warmup)       # W untimed warm-up pivots
    solve_s += time.perf_counter() - t0
    pivots_before = int(s.stats()["iterations"])
    s.reset_stats()
    s.set_sampling(None)   # the timed region is the production path: graph replays only, no event-bracketed (eager) iteration inside it
    barrier()
    t0 = time.perf_counter()
    s.continue_solve(a.steps)        # exactly K timed pivots
    barrier()
    dt = time.perf_counter() - t0
    solve_s += dt
    st_timed = s.stats()
    done = st_timed["iterations"]
    try:
        live_timed = int(s.state("shard_live")[0]) if world > 1 else None
    except Exception:
        live_timed = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([float(done)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total = float(c.item()) / world if sharded else float(c.item())   # sharded: every rank counts the same pivots
    else:
        total = float(done)
    # event-bracketed sampling pass (outside the timed region, same solve, the pivots that follow): >= 8 samples
    # of every kernel whatever K is; all ranks run it (the sharded exchange needs every rank)
    samples_in_region = int(st_timed["sweep_launches"])
    s.set_sampling(True)
    t0 = time.perf_counter()
    s.continue_solve(a.samples)
    solve_s += time.perf_counter() - t0
    s.set_sampling(False)
    st = s.stats()
    if rank == 0:
        kern = kernel_report(st)
        dom = max((k for k in ("fused", "sweep") if k in kern), key=lambda k: kern[k]["total_ms"], default=None)
        # the pricing path that shards over column blocks = tableau-row sweep + d/gamma update + pricing scan
        pricing_us = None
        if "update" in kern and ("sweep" in kern or "row_sparse" in kern):
            pricing_us = kern["sweep" if "sweep" in kern else "row_sparse"]["avg_us"] + kern["update"]["avg_us"]
        roofline = None
        try:
            head_iters = int(s.state("primal_head_launches")[0])
        except Exception:
            head_iters = 0
        if dom:
            which = ("sweep_band" if st.get("banded_sweep") else "sweep") if dom == "sweep" else dom
            if dom == "fused" and head_iters > 0:
                which = "primal_head"
            traffic, traffic_src = pmc_traffic(a, dom) if world == 1 else (None, None)  # the PMC figure is the unsharded kernel's
            roofline = dict(bound="hbm", kernel=KERNEL_NAMES[which], achieved=kern[dom]["gbs"], peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=kern[dom]["frac"], traffic=traffic, traffic_measured=("committed:profiles/" + traffic_src) if traffic else None,
                            traffic_unit=f"HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/{traffic_src})" if traffic else None,
                            avg_launch_us=kern[dom]["avg_us"], launches=kern[dom]["launches"],
                            algorithmic_bytes_per_launch=kern[dom]["algorithmic_bytes_per_launch"],
                            samples=f"a {a.samples}-pivot event-bracketed pass right after the timed region (HIP events stamped by the kernels on the "
                                    f"launch stream; {samples_in_region} instrumented iterations inside the timed region itself: it runs graph replays only)",
                            other_kernels={k: v for k, v in kern.items() if k not in (dom, "ftran")},
                            ftran=dict(kernel=KERNEL_NAMES["ftran"], column=dict(early=kern.get("ftran"))))
        out = dict(metric="simplex pivots/sec", value=total / dt, unit="pivots/s", n_gpus=world, steps=a.steps,
                   warmup=a.warmup, ms_per_step=dt * 1e3 / max(done, 1), higher_is_better=True,
                   # N > 1, one LP: "strong" only if the column-block sharding was LIVE in the timed pivots; with the deferred sharding
                   #  the driver's early window runs as bit-identical replicas — every GPU doing the whole pivot — which is
                   # neither: the label says so, value_vs_1gpu is then null, and the sharded figures are `late_sharded` / `pricing_speedup_vs_1gpu`
                   scaling=("weak" if (world > 1 and not sharded) else ("strong" if (world == 1 or live_timed) else "replicated")),
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=f"config 4: random LP {a.rows} vars x {a.cols} constraints, {a.nnz_per_row} nnz/row "
                                        f"(0.1% fill), Max c'x, Ax<=b, x>=0; primal simplex with PSE+DSE from the slack basis; "
                                        f"timed pivots {a.warmup}..{a.warmup + a.steps}",
                               rows=a.rows, cols=a.cols, nnz=int(st["nnz"]), seed=a.seed,
                               parallelism=("1 GPU" if world == 1 else
                                            (f"{world} GPUs: one LP; while the nucleus is small the ranks run as identical replicas (deferred sharding), then the "
                                             f"pricing path is sharded over {world} column blocks (and, from a "
                                             f"nucleus of 8 192 on, the streaming pass of the nucleus inverse over row strips), "
                                             f"per-pivot exchanges through {mdist.transport_name(s)}; FTRAN/BTRAN heads and the fold replicated"
                                             if sharded else f"{world} GPUs, one independent LP of the family per rank")) + (" [oversubscribed test rig: all ranks on one GPU]" if oversub else ""),
                               nucleus_size_at_end=int(st_timed["nucleus_size"]), objective_at_end=s.objective(),
                               completed_steps=int(done), bound_flips=int(st_timed["bound_flips"]),
                               pricing_path_us_per_pivot=pricing_us,
                               vs_baseline_note="BASELINE.md §1: the reference publishes no number for this metric"),
                   roofline=roofline)
        if ranks_info:
            # deferred sharding: in the small-nucleus regime the ranks run as bit-identical replicas without exchanges; the
            # line says whether the column-block sharding was live in the timed pivots (it is in the late window)
            ranks_info["sharding_live_in_timed_window"] = bool(live_timed) if live_timed is not None else None
            out["ranks"] = ranks_info
        out["provenance"] = dict(value="live", roofline_achieved="live (HIP events stamped by the kernels, this run)",
                                 roofline_traffic="committed (PMC passes cannot run inside the timed run)", windows="live",
                                 full_solve="live", cpu_baseline="live")
        if world == 1:
            cfg4 = (a.rows, a.cols, a.nnz_per_row, a.seed) == (100000, 100000, 100, 4)
            if not a.no_windows:
                windows = {}
                for name, path, warm, steps in (("mid", MID_BASIS, 64, a.window_steps[0]), ("late", LATE_BASIS, 32, a.window_steps[1])):
                    if not cfg4 or not os.path.exists(path):
                        windows[name] = None
                        continue
                    windows[name] = window_from_basis(M, prob, path, warm, steps, min(a.samples, 16))
                out["windows"] = windows
                late = windows.get("late")
                if late and "fused" in late.get("kernels", {}):
                    # The roofline object describes the kernel the SOLVE is bound by: the pass over the nucleus inverse in the
                    # large-nucleus regime, where > 90 % of the wall time of the full solve is spent (measured live in the late
                    # window).  In the driver-timed window itself (a nucleus of a few dozen columns) no kernel is bandwidth-bound:
                    # the tableau row only touches the columns that meet supp(rho); its kernels are listed under `timed_window`.
                    lk = late["kernels"]["fused"]
                    traffic, traffic_src = pmc_traffic(a, "stream_late")
                    timed = {k_: v_ for k_, v_ in (roofline or {}).items() if k_ != "ftran"} or None
                    ftran_obj = (roofline or {}).get("ftran") or dict(kernel=KERNEL_NAMES["ftran"], column=dict(early=kern.get("ftran")))
                    roofline = dict(bound="hbm", kernel="k_stream_w (" + KERNEL_NAMES["fused"] + ")", achieved=lk["gbs"], peak=HBM_PEAK_GBS,
                                    unit="GB/s", frac=lk["frac"], traffic=traffic,
                                    traffic_measured=("committed:profiles/" + traffic_src) if traffic else None,
                                    traffic_unit=f"HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/{traffic_src})" if traffic else None,
                                    avg_launch_us=lk["avg_us"], launches=lk["launches"],
                                    algorithmic_bytes_per_launch=lk["algorithmic_bytes_per_launch"],
                                    window=f"late window: {late['steps']} pivots from the committed basis with a nucleus of "
                                           f"{late['nucleus_size_at_start']} columns (the regime of > 90 % of the solve's wall time)",
                                    samples=late["sampling"], timed_window=timed,
                                    ftran=ftran_obj)
                    # the same window at PIVOT level: the bytes a late pivot must move (the 8 k^2-byte pass, the fold's 16 k^2 every
                    # 32nd pivot net of the pass it replaces) over the whole pivot's time — what the solve actually achieves per pivot
                    kk = float(late["nucleus_size_at_start"])
                    pivot_bytes = 8.0 * kk * kk + (16.0 * kk * kk - 8.0 * kk * kk) / 32.0
                    roofline["pivot_level"] = dict(bytes_per_pivot=pivot_bytes, us_per_pivot=late["us_per_pivot"],
                                                   achieved=pivot_bytes / (late["us_per_pivot"] * 1e-6) / 1e9, peak=HBM_PEAK_GBS,
                                                   frac=pivot_bytes / (late["us_per_pivot"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                   note="late window: (8 k^2 + 8 k^2 / 32) bytes per pivot / us per pivot / peak — the kernel-level "
                                                        "fraction above times the share of a pivot the pass takes")
                    out["roofline"] = roofline
                if roofline:
                    for name in ("mid", "late"):
                        if windows.get(name) and "ftran" in windows[name].get("kernels", {}):
                            roofline["ftran"]["column"][name] = windows[name]["kernels"]["ftran"]
                    if windows.get("late") and "dense_ftran" in windows["late"].get("kernels", {}):
                        roofline["ftran"]["dense_rhs_late"] = dict(kernel=KERNEL_NAMES["dense_ftran"], window="late window (saved basis)",
                                                                   **windows["late"]["kernels"]["dense_ftran"])
                    roofline["ftran"]["note"] = (
                        "north_star's FTRAN target names the solve against the basis factor.  Here B^-1 is the explicit nucleus inverse: "
                        "the column FTRAN alpha_q = B^-1 a_q reads |list| columns of it plus the nucleus columns of A (a latency-bound "
                        "gather: reported in us and bytes, a bandwidth fraction says nothing about it).  The FTRAN that is a stream is "
                        "the dense-rhs one, x_B = B^-1 (b - N x_N) (`dense_rhs_late`, k_stream_w's tau side).  The pass of every primal "
                        "pivot (`w_pass_v` in the windows) computes v = B^-T alpha_q: BTRAN-shaped, not an FTRAN")
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(lp, a.warmup, a.steps, a.cpu_pivots)
                cbv = out["cpu_baseline"]
                cbv["gpu_over_cpu_same_window"] = out["value"] / cbv["value"] if cbv.get("value") else None
            # how far "same pivots as the reference algorithm" is pinned on this workload, and what covers the rest
            out["parity"] = dict(oracle_identical_through_pivot=9652,
                                 pinned_by="tests/golden/cfg4_oracle_trace.npz (first 8 000 pivots, driver-run test); the sequences part at pivot 9 653 "
                                           "because the REFERENCE algorithm's steepest-edge weights drift (DESIGN.md §8)",
                                 beyond="defining equations of every solve against A at nuclei of 9 999 and 20 493 (tests/test_late_regime.py), "
                                        "duality certificate of the finished solve (full_solve.certificate, checked on the box)")
            if not a.no_full_solve:
                s.set_sampling(None)   # no event-bracketed iterations in the long run
                out["full_solve"] = full_solve_live(s, lp, a, T_START, solve_s, pivots_before)
            del s
            s = None
            if not a.no_factor_transport:
                try:
                    out["factor_transport"] = factor_transport(a)
                except Exception as e:   # reported, never voids the line
                    out["factor_transport"] = dict(error=str(e)[:200])
    # N > 1, sharded: the late window as well (all ranks take part): this is where the row-sharded streaming pass of
    # the nucleus inverse pays; a failure here is reported inside the line, it does not void the timed figure above
    late_sharded = None
    if world > 1:
        dist.barrier()
        if mailbox and rank == 0:
            mdist.remove_mailbox(mailbox)
    if world > 1 and sharded and not a.no_windows and os.path.exists(LATE_BASIS) and \
            (a.rows, a.cols, a.nnz_per_row, a.seed) == (100000, 100000, 100, 4):
        def max_over_ranks(x):
            t = torch.tensor([x], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        s = None  # free the early-window solver (its device buffers) before the 8.6 GB nucleus inverse of the late basis
        try:
            late_sharded = window_from_basis(M, prob, LATE_BASIS, 32, a.window_steps[1], min(a.samples, 16),
                                             shard=(dist, mdist, barrier, max_over_ranks))
        except Exception as e:  # every rank leaves setup_sharding through the same door; later failures are bounded waits
            late_sharded = dict(error=str(e))
    # N > 1: the same two windows UNSHARDED, measured by rank 0 in this very run (the other ranks wait at the barrier), so that
    # the line carries its own 1-GPU reference: value_vs_1gpu for the timed window, and the ratio of the pricing path
    # (tableau-row sweep + d / gamma update + pricing scan: what north_star's ">= 3x at 8 GPUs" is about) for the late window
    if world > 1 and sharded:
        if rank == 0:
            try:
                ref = {}
                s1 = prob.solve(budget=0, profile=True)
                s1.continue_solve(a.warmup)
                s1.reset_stats()
                s1.set_sampling(None)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s1.continue_solve(a.steps)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t0
                ref["timed_window_pivots_per_s"] = float(s1.stats()["iterations"]) / dt1
                del s1
                # (replicas in the timed window: the ratio would compare one GPU with one GPU — null)
                out["value_vs_1gpu"] = (out["value"] / ref["timed_window_pivots_per_s"]) if live_timed else None
                if late_sharded and "error" not in late_sharded and not a.no_windows:
                    lu = window_from_basis(M, prob, LATE_BASIS, 32, a.window_steps[1], min(a.samples, 16), dense_ftran=False)
                    def pricing(kk_):
                        return (kk_.get("sweep", {}).get("avg_us") or 0.0) + (kk_.get("update", {}).get("avg_us") or 0.0)
                    pu, ps = pricing(lu["kernels"]), pricing(late_sharded["kernels"])
                    ref["late_us_per_pivot"] = lu["us_per_pivot"]
                    ref["late_pricing_path_us"] = pu
                    late_sharded["pricing_path_us"] = ps
                    if pu > 0 and ps > 0:
                        out["pricing_speedup_vs_1gpu"] = pu / ps
                    ref["late_us_per_pivot_sharded"] = late_sharded["us_per_pivot"]
                out["unsharded_same_run"] = {k_: round(v_, 2) for k_, v_ in ref.items()}
            except Exception as e:
                out["unsharded_same_run"] = dict(error=str(e)[:200])
        dist.barrier()
    if rank == 0:
        if late_sharded is not None:
            out.setdefault("windows", {})["late_sharded"] = late_sharded
            # top level of an N > 1 line: `value` is a replica window by construction (deferred sharding), the scaling evidence is here
            if "error" not in late_sharded:
                out["late_sharded"] = dict(us_per_pivot=late_sharded.get("us_per_pivot"), pivots_per_s=late_sharded.get("pivots_per_s"),
                                           pricing_path_us=late_sharded.get("pricing_path_us"), k=late_sharded.get("nucleus_size_at_start"),
                                           unsharded_us_per_pivot=(out.get("unsharded_same_run") or {}).get("late_us_per_pivot"))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"bench_detail_n{world}.json"), "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
        real_stdout.write(json.dumps(compact_line(out)) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
