#!/usr/bin/env python3
"""solve_mps — counterpart of the reference's examples/solve_mps.rs (19-43): read a free-format MPS file,
minimise, print the objective and the non-zero variables.

    python examples/solve_mps.py model.mps [--max] [--backend hip|oracle] [--budget N]

The default back end is the MI355X engine (libminilp_hip.so, no CPU fallback); `--backend oracle` runs
the CPU restatement of minilp 0.2.2 for comparison (test infrastructure, not the product path)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--max", action="store_true", help="maximise instead of minimise (solve_mps.rs:32 minimises)")
    ap.add_argument("--backend", default="hip", choices=["hip", "oracle"])
    ap.add_argument("--all", action="store_true", help="print zero-valued variables too")
    a = ap.parse_args()
    if a.backend == "hip":
        import minilp_amd as B
    else:
        from oracle import minilp_oracle as B
    text = open(a.file).read()
    t0 = time.time()
    f = B.MpsFile(text, B.MAXIMIZE if a.max else B.MINIMIZE)  # MpsFile::parse (mps.rs:39)
    t1 = time.time()
    try:
        sol = f.problem.solve()
    except B.Infeasible:
        print("problem %s: infeasible" % f.problem_name)
        return 1
    except B.Unbounded:
        print("problem %s: unbounded" % f.problem_name)
        return 1
    t2 = time.time()
    print("problem %s: %d variables, parsed in %.3fs, solved in %.3fs" % (f.problem_name, len(f.variables), t1 - t0, t2 - t1))
    print("objective: %.12g" % sol.objective())
    x = sol.values()
    for name, var in sorted(f.variables.items(), key=lambda kv: kv[1]):
        if a.all or x[var] != 0.0:
            print("%s = %.12g" % (name, x[var]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
