#!/usr/bin/env python3
"""solve_mps — counterpart of the reference's examples/solve_mps.rs (19-43): read a free-format MPS file,
minimise, print the objective and the non-zero variables.

    python examples/solve_mps.py model.mps [--max] [--all]

Runs on the MI355X engine (libminilp_hip.so); there is no CPU back end.  `run(B, ...)` takes the module that
provides the reference's API so that the tests can drive the same code with their checker."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(B, path, maximize=False, show_all=False):
    text = open(path).read()
    t0 = time.time()
    f = B.MpsFile(text, B.MAXIMIZE if maximize else B.MINIMIZE)  # MpsFile::parse (mps.rs:39)
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
        if show_all or x[var] != 0.0:
            print("%s = %.12g" % (name, x[var]))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--max", action="store_true", help="maximise instead of minimise (solve_mps.rs:32 minimises)")
    ap.add_argument("--all", action="store_true", help="print zero-valued variables too")
    a = ap.parse_args()
    import minilp_amd as B
    return run(B, a.file, a.max, a.all)


if __name__ == "__main__":
    sys.exit(main())
