import sys, time, os
sys.path.insert(0, "/root/repo")
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_mixed_lp(300, 8000, 60, 5)
s = lpgen.build_problem(M.Problem, lp).solve()
os.environ["X"]="1"
for i in range(3):
    t=time.time(); c = s.clone(); t1=time.time(); del c; t2=time.time()
    print("clone %.2f ms  free %.2f ms" % ((t1-t)*1e3, (t2-t1)*1e3), flush=True)
