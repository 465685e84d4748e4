"""Config 3 (and other sparse models) on the hypersparse path against the multi-kernel path and the CPU restatement:
wall time of solve(), time inside the pivot loops, iterations taken by the persistent workgroup, bail-outs.
usage: hyper_profile.py [m n k seed]   (default: config 3 = mixed 6000 x 10000, 4 per row, seed 3, through the MPS reader)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O

m, n, k, seed = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (6000, 10000, 4, 3)))
fam = sys.argv[5] if len(sys.argv) > 5 else "mixed"
lp = {"mixed": lpgen.gen_mixed_lp, "cover": lpgen.gen_cover_lp, "sparse": lpgen.gen_sparse_lp}[fam](m, n, k, seed)
text = lpgen.to_mps(lp)
pg = M.MpsFile(text, lp["direction"]).problem
po = O.MpsFile(text, lp["direction"]).problem
pg.solve()
for mode in ("1", "1p", "0"):
    os.environ["MLP_HYPER"] = mode[0]
    os.environ.pop("MLP_HYPER_PROF", None)
    if mode == "1p":
        os.environ["MLP_HYPER_PROF"] = "1"
    best = (1e9, None)
    for _ in range(3):
        t = time.perf_counter(); s = pg.solve(); dt = time.perf_counter() - t
        if dt < best[0]:
            best = (dt, s.stats())
    dt, st = best
    print(f"MLP_HYPER={mode}: solve {dt * 1e3:.1f} ms, in pivot loops {st['solve_wall_s'] * 1e3:.1f} ms, pivots {st['iterations']} "
          f"({st['solve_wall_s'] * 1e6 / max(1, st['iterations']):.1f} us/pivot in loops), hyper {st['hyper_iters']}, bails {st['hyper_bails']}, "
          f"nucleus {st['nucleus_size']}, obj {s.objective():.9f}, bail reasons {s.state('hyper_bail_reasons').astype(int).tolist()}", flush=True)
    if mode == "1p" and st["hyper_iters"]:
        names = ["pricing", "btran", "touched cols", "row pull", "ratio", "ftran head", "alpha_K", "alpha_S", "plan", "tau+xB/beta", "eta update", "partition+apply"]
        prof = s.state("hyper_profile")
        print("   hypersparse kernel, us per iteration by stage:", ", ".join(f"{nm} {x / st['hyper_iters']:.2f}" for nm, x in zip(names, prof)),
              f"| total {prof[:12].sum() / st['hyper_iters']:.2f} | prologues {prof[14] / 1e3:.2f} ms | sub-marks {[round(x / st['hyper_iters'], 2) for x in prof[16:24]]} | launches {prof[12] / 1e3:.1f} ms at {prof[13] / max(prof[12], 1e-9):.0f} MHz shader clock", flush=True)
bo = 1e9
for _ in range(3):
    t = time.perf_counter(); so = po.solve(); bo = min(bo, time.perf_counter() - t)
print(f"CPU restatement: {bo * 1e3:.1f} ms, obj {so.objective():.9f}")
