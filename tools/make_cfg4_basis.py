#!/usr/bin/env python3
"""Generate the mid-solve basis fixtures of config 4 (run on the GPU box; output under gpurun_out/).

  python tools/make_cfg4_basis.py --at 45000 240000 --out gpurun_out/basis

Solves config 4 (lpgen.gen_sparse_lp(100000, 100000, 100, 4)) from the slack basis, stops at the given pivot
counts and writes a mode-1 checkpoint (sets, flags, x_N, f32 steepest-edge weights) per stop, gzip-compressed:
cfg4_basis_p<pivots>.bin.gz + a JSON line with the nucleus size, objective and wall time at each stop.  The
committed copies live under tests/golden/ and are what bench.py's mid/late windows start from."""
import argparse
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--at", type=int, nargs="+", default=[45000, 240000])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "basis"))
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--cols", type=int, default=100000)
    ap.add_argument("--nnz-per-row", type=int, default=100)
    ap.add_argument("--seed", type=int, default=4)
    a = ap.parse_args()
    import minilp_amd as M
    from minilp_amd import lpgen
    os.makedirs(a.out, exist_ok=True)
    lp = lpgen.gen_sparse_lp(a.rows, a.cols, a.nnz_per_row, a.seed)
    s = lpgen.build_problem(M.Problem, lp).solve(budget=0)
    done, t0 = 0, time.perf_counter()
    for stop in sorted(a.at):
        s.continue_solve(stop - done)
        done = stop
        st = s.stats()
        blob = s.save_basis(1)
        path = os.path.join(a.out, f"cfg4_basis_p{stop}.bin.gz")
        with gzip.open(path, "wb", compresslevel=9) as f:
            f.write(blob)
        rec = dict(pivots=int(st["iterations"]), nucleus_size=int(st["nucleus_size"]), objective=s.objective(),
                   wall_s=time.perf_counter() - t0, raw_bytes=len(blob), gz_bytes=os.path.getsize(path), file=os.path.basename(path),
                   workload=dict(rows=a.rows, cols=a.cols, nnz_per_row=a.nnz_per_row, seed=a.seed))
        print(json.dumps(rec), flush=True)
        with open(os.path.join(a.out, "cfg4_basis_index.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
