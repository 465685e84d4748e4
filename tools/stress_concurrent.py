#!/usr/bin/env python3
"""Stress of the cross-workgroup hand-offs under oversubscription: P processes solve the SAME LP concurrently on ONE GPU
(their kernels interleave on the CUs, so ticketed reductions, the in-kernel publish / wait of the fused primal ratio
test and the hipGraph replays all run while other queues compete for the same CUs) and every process must take the
pivot sequence of a solo run.

    python tools/stress_concurrent.py P M N K PIVOTS [family]
"""
import os
import sys
import time

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(idx, m, n, k, pivots, family, start, out):
    import minilp_amd as M
    from minilp_amd import lpgen
    lp = lpgen.gen_cover_lp(m, n, k, 4) if family == "cover" else lpgen.gen_sparse_lp(m, n, k, 4)
    p = lpgen.build_problem(M.Problem, lp)
    s = p.solve(budget=0, trace=True)
    start.wait()                       # all processes enter the pivot loop together
    t0 = time.time()
    s.continue_solve(pivots)
    out.put((idx, [t[:5] for t in s.trace()], s.objective(), time.time() - t0))


def main():
    procs_n, m, n, k, pivots = (int(x) for x in sys.argv[1:6])
    family = sys.argv[6] if len(sys.argv) > 6 else "sparse"
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    start = ctx.Barrier(procs_n)
    procs = [ctx.Process(target=worker, args=(i, m, n, k, pivots, family, start, out)) for i in range(procs_n)]
    for p in procs:
        p.start()
    res = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    # solo reference in this process, after the others are gone
    import minilp_amd as M
    from minilp_amd import lpgen
    lp = lpgen.gen_cover_lp(m, n, k, 4) if family == "cover" else lpgen.gen_sparse_lp(m, n, k, 4)
    ref = lpgen.build_problem(M.Problem, lp).solve(budget=pivots, trace=True)
    rtr = [t[:5] for t in ref.trace()]
    ok = all(tr == rtr for _, tr, _, _ in res) and len(rtr) > 0
    print("concurrent processes: %d, pivots each: %s, wall: %s | solo pivots: %d | identical to solo: %s" % (
        procs_n, [len(tr) for _, tr, _, _ in res], ["%.2f" % dt for *_, dt in res], len(rtr), ok), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
