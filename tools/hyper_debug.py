"""Debug aid: first pivot at which the hypersparse path leaves the oracle's trace on a cover (dual-only) instance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MLP_HYPER", "1")
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
m, n, k, seed = (int(a) for a in sys.argv[1:5])
lp = lpgen.gen_cover_lp(m, n, k, seed)
so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
to = [t[:5] for t in so.trace()]
print("oracle pivots", len(to))
sg = lpgen.build_problem(M.Problem, lp).solve(budget=0, trace=True)
step = int(sys.argv[5]) if len(sys.argv) > 5 else 64
done = 0
while True:
    try:
        sg.continue_solve(step)
    except Exception as e:
        print("exception after", done, ":", e)
        break
    tg = [t[:5] for t in sg.trace()]
    st = sg.stats()
    bad = next((i for i, (a, b) in enumerate(zip(tg, to)) if a != b), None)
    if bad is not None:
        print("first difference at pivot", bad, "engine", tg[bad], "oracle", to[bad], "| hyper", st["hyper_iters"], "bails", st["hyper_bails"],
              "k", st["nucleus_size"], "max_pivot_err", st["max_pivot_err"], "kases", st["kase"])
        print("engine around:", tg[max(0, bad - 2):bad + 2])
        print("oracle around:", to[max(0, bad - 2):bad + 2])
        break
    done = len(tg)
    if not sg.budget_exhausted:
        print("finished identical:", done, "pivots; hyper", st["hyper_iters"], "bails", st["hyper_bails"], "max_pivot_err", st["max_pivot_err"])
        break
