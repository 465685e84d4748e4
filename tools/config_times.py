"""Wall time of configs 2 and 3 (BASELINE.json) on the GPU engine and on the CPU restatement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O

def run(name, lp, via_mps=False):
    row = [name]
    for B, tag in ((M, "gpu"), (O, "cpu")):
        if via_mps:
            text = lpgen.to_mps(lp)
            p = B.MpsFile(text, lp["direction"]).problem
        else:
            p = lpgen.build_problem(B.Problem, lp)
        t = time.time(); s = p.solve(); dt = time.time() - t
        st = s.stats()
        its = st.get("iterations", st.get("primal_iters", 0) + st.get("dual_iters", 0))
        row.append(f"{tag}: {dt:.3f}s obj={s.objective():.9f} pivots={its} ({its/max(dt,1e-9):.0f}/s)")
    print(" | ".join(row), flush=True)

run("warm-up", lpgen.gen_dense_lp(60, 60, 2))
run("config 2 dense 1000x1000", lpgen.gen_dense_lp(1000, 1000, 2))
run("config 3 mixed 6000x10000 via MPS", lpgen.gen_mixed_lp(6000, 10000, 4, 3), via_mps=True)
