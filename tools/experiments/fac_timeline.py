"""Where the time of a compact-factor solve goes (MLP_KPROF=1 marks of factor.inc): run to `skip` pivots, then `n` single pivots, print the marks."""
import os, sys
os.environ["MLP_KPROF"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
S, D, deg, skip, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
lp = lpgen.gen_mixed_lp(S, D, deg, 3) if os.environ.get("FAMILY") == "mixed" else lpgen.gen_transport_lp(S, D, deg, tight=0.4)
s = lpgen.build_problem(M.Problem, lp).solve(budget=skip)
names = ["entry", "coef", "seg1", "seg2", "bump", "seg4", "seg5", "join", "epilogue"]
for it in range(n):
    s.continue_solve(1)
    tl = np.array(s.state("kernel_timeline"))
    st = s.stats()
    for d, tag in ((0, "FTRAN"), (10, "BTRAN")):
        m = tl[d:d + 9]
        if (m < 0).all():
            continue
        base = m[0]
        print("%s levels %d bump %d: " % (tag, st["factor_levels"], st["factor_bump"]) + "  ".join("%s %.1f" % (names[i], m[i] - base) for i in range(1, 9) if m[i] >= 0))
pl = np.array(s.state("factor_plan")).astype(int)
print("meta", pl[:8].tolist())
print("segments (kind, first, last, positions):", pl[8:].reshape(-1, 4).tolist())
