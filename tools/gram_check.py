"""Gram mode on small instances with the large-nucleus machinery forced on: pivot-for-pivot against
MLP_GRAM=0 (the streaming pass), the drift monitor, the number of pivots that took the path.
usage: gram_check.py [family m n k seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MLP_LOWRANK", "16")
os.environ["MLP_BIGTILE"] = "1"; os.environ["MLP_LDPAD"] = "16"; os.environ["MLP_BANDED"] = "1"
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen

cases = [("sparse", 200, 200, 10, 4), ("sparse", 700, 600, 12, 6), ("sparse", 2000, 2000, 20, 1), ("mixed", 300, 400, 8, 4)]
if len(sys.argv) >= 6:
    cases = [(sys.argv[1],) + tuple(int(a) for a in sys.argv[2:6])]
for fam, m, n, k, seed in cases:
    lp = lpgen.gen_sparse_lp(m, n, k, seed) if fam == "sparse" else lpgen.gen_mixed_lp(m, n, k, seed)
    res = {}
    for gram in ("1", "0"):
        os.environ["MLP_GRAM"] = gram
        s = lpgen.build_problem(M.Problem, lp).solve(trace=True)
        st = s.stats()
        res[gram] = ([t[:5] for t in s.trace()], s.objective())
        print(f"{fam} {m}x{n} gram={gram}: pivots {st['iterations']} obj {s.objective():.12g} gram_pivots {st['gram_pivots']} "
              f"rebuilds {st['gram_rebuilds']} gram_err {st['gram_err']:.2e} kase {list(st['kase'])} reinvert {s.reinvert():.2e}", flush=True)
    same = res["1"][0] == res["0"][0]
    first = next((i for i, (a, b) in enumerate(zip(res["1"][0], res["0"][0])) if a != b), None)
    print(f"  identical pivot sequences: {same} (first difference at {first}), objective diff {abs(res['1'][1] - res['0'][1]):.2e}", flush=True)
