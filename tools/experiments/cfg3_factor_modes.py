"""Config 3 on the compact factor: pivot trace against the oracle for the carriers of the bump (dense inverse / sparse LU) and the default path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
lp = lpgen.gen_mixed_lp(6000, 10000, 4, 3)
so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
to = [t[:5] for t in so.trace()]
for name, env in (("default", {}), ("factor dense", {"MLP_FACTOR": "1", "MLP_FACTOR_SB": "0"}), ("factor sparse from 2", {"MLP_FACTOR": "1", "MLP_FACTOR_SB_FROM": "2"}),
                  ("factor sparse from 48", {"MLP_FACTOR": "1"}), ("factor dense J=1", {"MLP_FACTOR": "1", "MLP_FACTOR_SB": "0", "MLP_FACTOR_J": "1"}),
                  ("factor sparse J=1", {"MLP_FACTOR": "1", "MLP_FACTOR_SB_FROM": "2", "MLP_FACTOR_J": "1"})):
    for k in ("MLP_FACTOR", "MLP_FACTOR_SB", "MLP_FACTOR_SB_FROM", "MLP_FACTOR_J"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
    tg = [t[:5] for t in sg.trace()]
    div = next((i for i, (a, b) in enumerate(zip(tg, to)) if a != b), None)
    st = sg.stats()
    print(f"{name:24s} pivots {len(tg)} (oracle {len(to)}) first divergence {div} obj {sg.objective():.9f} (oracle {so.objective():.9f}) bump max {st['factor_bump_max']} "
          f"sb {sg.state('factor_sb').astype(int).tolist()} max_pivot_err {st['max_pivot_err']:.2e}", flush=True)
