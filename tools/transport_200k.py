#!/usr/bin/env python3
"""Evidence for SURVEY §8 row f3: a LARGE model whose nucleus is large and stays sparse.

Family: minilp_amd.lpgen.gen_transport_lp — a network with gains (S supply rows, D demand rows, `deg` arcs per demand node,
continuous random data): every column has two entries, so every basis is a forest (its LU has no fill, an iterated
column-singleton peel leaves no bump) while the STRUCTURAL nucleus — what the explicit inverse of the default representation
holds densely, 8 k^2 bytes — grows to ~0.65 m columns.  At S = D = 100 000 (m = 200 000 rows, n = 400 000 columns, nnz = 800 000)
that is k = 130 000: 135 GB of inverse and 16 k^2 bytes of traffic per pivot.

    python tools/transport_200k.py [S D deg tight] [--paths oracle,factor,dense] [--dense-pivots N] [--json out.json] [--family mixed]

  oracle : the single-threaded restatement of the reference (LU + eta file), whole solve, us per pivot per 20 000-pivot chunk
  factor : minilp_amd on the compact factor (auto-selected when the nucleus passes MLP_FACTOR_FROM slots)
  dense  : minilp_amd with MLP_FACTOR=0 (explicit inverse only), for at most --dense-pivots pivots past the point where the
           other run switched (it would need hours, or end in MLP_ENOMEM)
  factor_dense_bump : the compact factor with MLP_FACTOR_SB=0 (the bump through its dense inverse, at most 1 024 columns: the tree
           before round 5), for at most --dense-pivots pivots
Objectives of finished runs must agree to 1e-9 relative."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CODE = r'''
import json, sys, time
sys.path.insert(0, %r)
import numpy as np
from minilp_amd import lpgen
S, D, deg, tight, path, limit = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5], int(sys.argv[6])
family = sys.argv[7] if len(sys.argv) > 7 else "transport"
# (--family mixed: the config-3 generator at scale — S rows, D columns, deg entries per row, E/L/G operators)
lp = lpgen.gen_mixed_lp(S, D, deg, 3) if family == "mixed" else lpgen.gen_transport_lp(S, D, deg, tight=tight)
if path == "oracle":
    from oracle import minilp_oracle as B
else:
    import minilp_amd as B
prob = lpgen.build_problem(B.Problem, lp)
t0 = time.perf_counter()
rec = dict(path=path, m=lp["m"], n=lp["n"], nnz=int(len(lp["data"])), chunks=[])
try:
    s = prob.solve(budget=0)
    chunk = int(sys.argv[8]) if len(sys.argv) > 8 else 20000
    done = 0
    while True:
        t1 = time.perf_counter()
        s.continue_solve(chunk)
        dt = time.perf_counter() - t1
        st = s.stats()
        piv = int(st["iterations"]) if path != "oracle" else int(st["pivots"] + st["bound_flips"])
        e = dict(pivots=piv, us_per_pivot=round(dt * 1e6 / max(1, piv - done), 1), objective=s.objective())
        if path != "oracle":
            e.update(nucleus=int(st["nucleus_size"]), capacity=int(st["nucleus_capacity"]), factor=int(st["factor_active"]), levels=int(st["factor_levels"]),
                     refactorisations=int(st["factor_refactors"]), bump=int(st["factor_bump"]), bump_max=int(st["factor_bump_max"]), switches=int(st["factor_switches"]),
                     sparse_bump=dict(zip(("in_use", "factorisations", "fallbacks", "rounds"), s.state("factor_sb").astype(int).tolist())))
        rec["chunks"].append(e)
        print(json.dumps(e), file=sys.stderr, flush=True)
        done = piv
        if not s.budget_exhausted or (limit > 0 and piv >= limit):
            break
    rec.update(finished=not s.budget_exhausted, pivots=done, objective=s.objective(), wall_s=round(time.perf_counter() - t0, 1))
    if path != "oracle" and not s.budget_exhausted:
        x = s.values()
        lhs = np.add.reduceat(lp["data"] * x[lp["indices"]], lp["indptr"][:-1])
        viol = np.where(lp["ops"] == lpgen.LE, lhs - lp["rhs"], lp["rhs"] - lhs)
        rec.update(max_row_violation=float(viol.max()), min_x=float(x.min()), max_pivot_err=float(st["max_pivot_err"]))
except Exception as ex:
    rec.update(finished=False, error=str(ex)[:300], wall_s=round(time.perf_counter() - t0, 1))
print(json.dumps(rec))
''' % ROOT


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    S, D, deg, tight = (int(pos[0]), int(pos[1]), int(pos[2]), float(pos[3])) if len(pos) >= 4 else (100000, 100000, 4, 0.4)

    def opt(name, default):
        return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
    paths = opt("--paths", "oracle,factor,dense").split(",")
    dense_pivots = int(opt("--dense-pivots", "60000"))
    family = opt("--family", "transport")
    out = dict(family="transport (network with gains)" if family == "transport" else "mixed (config-3 generator at scale: S rows, D columns, deg per row)",
               S=S, D=D, deg=deg, tight=tight, runs={})
    for path in paths:
        env = dict(os.environ)
        if path == "dense":
            env["MLP_FACTOR"] = "0"
        if path == "factor_dense_bump":  # the compact factor as it was before round 5's sparse LU of the bump: dense inverse, 1 024 columns at most
            env["MLP_FACTOR_SB"] = "0"
        limit = dense_pivots if path in ("dense", "factor_dense_bump") else 0
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", CODE, str(S), str(D), str(deg), str(tight), path, str(limit), family, opt("--chunk", "20000")], env=env, capture_output=True, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        rec = json.loads(lines[-1]) if lines else dict(error=(r.stderr or "")[-400:])
        out["runs"][path] = rec
        print("%-7s %s" % (path, json.dumps({k: v for k, v in rec.items() if k != "chunks"})), flush=True)
        for e in rec.get("chunks", []):
            print("        ", e, flush=True)
    fin = {k: v for k, v in out["runs"].items() if v.get("finished")}
    objs = [v["objective"] for v in fin.values()]
    if len(objs) >= 2:
        out["objectives_agree"] = bool(max(objs) - min(objs) <= 1e-9 * max(1.0, abs(objs[0])))
        print("objectives agree to 1e-9:", out["objectives_agree"])
    if "--json" in sys.argv:
        with open(opt("--json", "out.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
