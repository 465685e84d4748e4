"""Where the TSP driver's wall time goes (config 5): time inside each Solution / Problem API call, the native min cut,
and the Python around them.  usage: tsp_profile.py tests/golden/bn130.tsp"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import minilp_amd as M
import tsp

acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[cls.__name__ + "." + name][0] += 1
            acc[cls.__name__ + "." + name][1] += time.perf_counter() - t
    setattr(cls, name, g)
for n in ("add_constraint", "fix_var", "unfix_var", "clone", "values", "objective", "__del__"):
    wrap(M.Solution, n)
wrap(M.Problem, "solve")
mc = M.min_cut
def mc2(w):
    t = time.perf_counter()
    try:
        return mc(w)
    finally:
        acc["min_cut"][0] += 1; acc["min_cut"][1] += time.perf_counter() - t
M.min_cut = mc2
name, pts = tsp.read_tsplib(sys.argv[1])
t0 = time.perf_counter()
s = tsp.TspSolver(M, pts)
cost, tour = s.solve()
wall = time.perf_counter() - t0
print(f"tour cost {cost:.10f}, wall {wall:.2f} s, {s.stats}")
tot = 0.0
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {n:6d} calls {t:7.3f} s  ({t / max(n, 1) * 1e3:.3f} ms each)")
    tot += t
print(f"  python around the calls        {wall - tot:7.3f} s")
