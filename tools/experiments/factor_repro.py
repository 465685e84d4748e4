"""Run-to-run reproducibility of a solve on the compact factor (sparse LU of the bump, dense tail, fused launches): two solves of the
config-3 family at 60 000 rows in two fresh processes — pivot count, objective bits, SHA-1 of x."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_mixed_lp(60000, 100000, 4, 3)
s = lpgen.build_problem(M.Problem, lp).solve()
st = s.stats()
print("RESULT", int(st["iterations"]), np.float64(s.objective()).tobytes().hex(), hashlib.sha1(np.asarray(s.values()).tobytes()).hexdigest(),
      int(st["factor_bump_max"]), s.state("factor_sb").astype(int).tolist())
''' % ROOT
out = []
for _ in range(2):
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True)
    out.append([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1])
    print(out[-1], flush=True)
print("identical:", out[0] == out[1])
