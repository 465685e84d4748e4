"""MLP_KPROF=1: in-kernel marks of the last FTRAN / BTRAN of the compact factor (k_fac_solve: entry, coefficients, the first five segments,
join, epilogue end) late in a config-3-family solve, with the plan of the walk."""
import os, sys
os.environ["MLP_KPROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
S, D, piv = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fam = sys.argv[4] if len(sys.argv) > 4 else "mixed"
lp = lpgen.gen_mixed_lp(S, D, 4, 3) if fam == "mixed" else lpgen.gen_transport_lp(S, D, 4, tight=0.4)
s = lpgen.build_problem(M.Problem, lp).solve(budget=piv)
for rep in range(4):
    s.continue_solve(5)
    tl = s.state("kernel_timeline")
    plan = s.state("factor_plan").astype(int)
    meta, segs = plan[:8], plan[8:].reshape(-1, 4)
    st = s.stats()
    print(f"pivots {int(st['iterations'])} levels {meta[0]} bump {meta[2]} col steps {meta[4]} row steps {meta[5]} small-level positions {meta[6]} segments (kind first last positions): {segs.tolist()}")
    for d, name in ((0, "FTRAN"), (1, "BTRAN")):
        m = tl[d * 10: d * 10 + 9]
        if m[0] < 0: continue
        print(f"   {name}: " + " ".join(f"{x - m[0]:.1f}" if x >= 0 else "-" for x in m), " [entry, coef, seg1..seg5, join, end] us;  sb", s.state("factor_sb").astype(int).tolist())
    print("   bump marks of the last solve (rhs built, phase 1 done, phase 2 done), relative to the FTRAN entry (the last solve of a dual iteration):", [round(x - tl[0], 1) for x in tl[20:23]])
