#!/usr/bin/env python3
"""Hit rate of a speculative multi-candidate pass.

The streaming pass over the nucleus inverse exists to form v = B^-T alpha_q (solver.rs:1114) for the entering column q
that pricing chose.  For a column c that pricing did NOT choose, B'^-T alpha'_c = B^-T alpha_c + s1 rho + s2 v after the
pivot, so extra right-hand sides carried through THIS pivot's read of W0 would let the NEXT pivot skip its pass — if
its entering column was among them.  This tool measures how often that would happen: at every pivot it ranks the
eligible non-basic columns by the pricing score d_j^2 / gamma_j (solver.rs:696-739) and asks which rank, among the
runners-up of pivot i, the entering variable of pivot i + 1 had.

  python tools/spec_hit_rate.py {mid|late} PIVOTS [out.json]

Reads d, gamma, the bound flags and nb_vars through mlp_solution_state after every pivot (host-paced: a measurement
tool, not a timed path)."""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import minilp_amd as M  # noqa: E402
from minilp_amd import lpgen  # noqa: E402

EPS = 1e-8  # solver.rs:11
TOP = 32


def ranking(s):
    d = s.state("nb_var_obj_coeffs")
    g = s.state("primal_edge_sq_norms")
    fl = s.state("nb_flags").astype(np.int64)
    nbv = s.state("nb_vars").astype(np.int64)
    at_min = (fl & 1) != 0
    at_max = (fl & 2) != 0
    skip = (at_min & (d > -EPS)) | (at_max & (d < EPS))
    score = np.where(skip, -np.inf, d * d / g)
    n_elig = int((~skip).sum())
    top = np.argpartition(-score, TOP)[:TOP]
    top = top[np.argsort(-score[top], kind="stable")]
    return [int(nbv[c]) for c in top if np.isfinite(score[c])], [float(score[c]) for c in top], n_elig


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "late"
    pivots = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    out = sys.argv[3] if len(sys.argv) > 3 else None
    lp = lpgen.gen_sparse_lp(100000, 100000, 100, 4)
    prob = lpgen.build_problem(M.Problem, lp)
    if which == "early":
        s = prob.solve(budget=0, trace=True)
        s.continue_solve(int(os.environ.get("SPEC_EARLY_SKIP", "2000")))
    else:
        blob = gzip.open(bench.MID_BASIS if which == "mid" else bench.LATE_BASIS, "rb").read()
        s = prob.solve_from_basis(blob, budget=0, trace=True)
        s.continue_solve(8)
    hist = np.zeros(TOP + 1, dtype=np.int64)  # hist[r] = next entering variable was the r-th runner-up (r = 1..TOP-1); hist[0] = beyond
    flips = 0
    gaps = []
    rankings, entering = [], []  # per pivot: the top-TOP variables BEFORE it, and the variable that entered
    prev_rank, prev_scores, n_elig = ranking(s)
    k0 = s.stats()["nucleus_size"]
    for i in range(pivots):
        n0 = len(s.trace())
        s.continue_solve(1)
        tr = s.trace()
        if len(tr) == n0:
            break
        ph, col, row, ev, lv, pc, ob = tr[-1]
        assert prev_rank[0] == ev, (i, prev_rank[:4], ev)  # the ranking restates the engine's pricing decision
        rankings.append(prev_rank)
        entering.append(int(ev))
        if row < 0:
            flips += 1
        cur_rank, cur_scores, n_elig = ranking(s)
        nxt = cur_rank[0]
        r = prev_rank.index(nxt) if nxt in prev_rank else 0
        hist[r] += 1
        gaps.append(prev_scores[1] / prev_scores[0] if prev_scores[0] > 0 else 0.0)
        prev_rank, prev_scores = cur_rank, cur_scores
    total = int(hist.sum())
    cum = {f"top{t}": float(hist[1:t + 1].sum()) / max(total, 1) for t in (1, 2, 3, 4, 8, 16, 31)}
    # Persistent candidate set (the form a build would take): z_c = B^-T alpha_c of a candidate survives a pivot through
    # z_c' = z_c - theta (v - rho) - sigma rho, so a set carried through ONE pass serves every later pivot whose entering
    # column is still in it; a miss costs one pass with T right-hand sides (the missed column + the T - 1 best runners-up not
    # yet held); the set holds at most M columns (the lowest-ranked leave first).
    policy = {}
    for T in (2, 4, 8, 16):
        for Mcap in (T, 2 * T, 32):
            S, passes = [], 0
            for R, e in zip(rankings, entering):
                if e in S:
                    S.remove(e)
                    continue
                passes += 1
                fresh = [v for v in R[1:] if v not in S][:T - 1]
                S = S + fresh
                if len(S) > Mcap:
                    pos = {v: j for j, v in enumerate(R)}
                    S.sort(key=lambda v: pos.get(v, TOP + 1))
                    S = S[:Mcap]
            policy[f"T{T}_M{Mcap}"] = passes / max(len(entering), 1)
    res = {
        "window": which, "pivots": total, "nucleus_size_start": k0, "nucleus_size_end": s.stats()["nucleus_size"],
        "eligible_columns_last": n_elig, "bound_flips": flips,
        "hist_rank_of_next_entering_among_runners_up": {str(r): int(hist[r]) for r in range(1, TOP)},
        "beyond_top31": int(hist[0]),
        "P_next_entering_within_first_t_runners_up": cum,
        "expected_passes_per_pivot_with_t_candidates": {k: 1.0 / (1.0 + v) for k, v in cum.items()},
        "passes_per_pivot_persistent_set_T_rhs_per_pass_M_held": policy,
        "median_score_ratio_second_over_first": float(np.median(gaps)) if gaps else None,
    }
    print(json.dumps(res, indent=1))
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
