# positions per thread of the update kernel (MLP_UPDATE_PT) on the mid / late windows of config 4 and on the first 20 000 pivots of the transport instance
for pt in 1 2 4; do
  echo "== MLP_UPDATE_PT=$pt"
  MLP_UPDATE_PT=$pt python tools/window_profile.py mid 512 64 2>&1 | grep "pivots in"
  MLP_UPDATE_PT=$pt python tools/window_profile.py late 256 32 2>&1 | grep "pivots in"
  MLP_UPDATE_PT=$pt python - <<'PY'
import time, sys, os
sys.path.insert(0, os.getcwd())
import minilp_amd as M
from minilp_amd import lpgen
lp = lpgen.gen_transport_lp(100000, 100000, 4, tight=0.4)
s = lpgen.build_problem(M.Problem, lp).solve(budget=0)
t0 = time.perf_counter(); s.continue_solve(20000); dt = time.perf_counter() - t0
print("transport first 20000: %.1f us/pivot" % (dt * 1e6 / 20000))
PY
done
