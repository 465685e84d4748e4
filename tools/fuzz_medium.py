"""Differential fuzz on medium instances of the generator families (every back-end feature on by env
rotation): objective parity with the oracle and feasibility; pivot-for-pivot on the non-degenerate families."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import check_feasible, obj_close

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(12345)
ENVS = [{}, {"MLP_LOWRANK": "3", "MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_BANDED": "1"}, {"MLP_BANDED": "1"},
        {"MLP_LOWRANK": "16", "MLP_BIGTILE": "1"}, {"MLP_GRAPH_ITERS": "1"}, {"MLP_NO_GRAPH": "1"},
        {"MLP_LOWRANK": "16", "MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_STREAM_BALANCED": "5"},
        {"MLP_LOWRANK": "8", "MLP_BIGTILE": "1", "MLP_STREAM_BALANCED": "0"},
        {"MLP_LOWRANK": "16", "MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_BANDED": "1"},
        {"MLP_BANDED": "1", "MLP_ORDER_FROM": "0", "MLP_ORDER_EVERY": "5"},
        {"MLP_BANDED": "1", "MLP_ORDER_FROM": "0", "MLP_ORDER_EVERY": "64", "MLP_LOWRANK": "8", "MLP_BIGTILE": "1"},
        {"MLP_BANDED": "1", "MLP_ORDER_FROM": "3", "MLP_ORDER_EVERY": "11", "MLP_SWEEP_PACKED": "0"},
        # round 5: pushed F products forced on small models, so that the small-nucleus primal head (and its three forms) takes pivots
        {"MLP_DETERMINISTIC": "0", "MLP_HYPER": "0"},
        {"MLP_DETERMINISTIC": "0", "MLP_HYPER": "0", "MLP_HEAD_APPLY": "0"},
        {"MLP_DETERMINISTIC": "0", "MLP_HYPER": "0", "MLP_PULL_INSIDE": "0"},
        {"MLP_DETERMINISTIC": "0", "MLP_PRIMAL_HEAD_K": "5"},
        {"MLP_DETERMINISTIC": "0", "MLP_HYPER": "0", "MLP_GRAPH_ITERS": "3", "MLP_RATIO_ONE": "0"},
        # round 6: the pulled F product (fpull.inc) forced on small instances (large-nucleus machinery + the grid forms of the ratio test), with
        # and without the banded sweep, eager and in long graphs; and the pushed form it replaces under the same machinery
        {"MLP_LOWRANK": "3", "MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_BANDED": "1", "MLP_STR_K": "0", "MLP_RATIO_ONE": "0", "MLP_HYPER": "0"},
        {"MLP_LOWRANK": "16", "MLP_BIGTILE": "1", "MLP_STR_K": "0", "MLP_RATIO_ONE": "0", "MLP_HYPER": "0", "MLP_NO_GRAPH": "1"},
        {"MLP_LOWRANK": "8", "MLP_BIGTILE": "1", "MLP_STR_K": "0", "MLP_RATIO_ONE": "0", "MLP_HYPER": "0", "MLP_GRAPH_ITERS": "3"},
        {"MLP_LOWRANK": "3", "MLP_BIGTILE": "1", "MLP_LDPAD": "16", "MLP_BANDED": "1", "MLP_STR_K": "0", "MLP_RATIO_ONE": "0", "MLP_HYPER": "0", "MLP_FPULL": "0"}]
if os.environ.get("FUZZ_ENVS") == "head":   # only the round-5 configurations
    ENVS = ENVS[-9:-4]
if os.environ.get("FUZZ_ENVS") == "fpull":  # only the round-6 configurations
    ENVS = ENVS[-4:]
if os.environ.get("FUZZ_ENVS") == "factor":   # round 5, second session: the compact factor forced, the carriers of its bump
    ENVS = [{"MLP_FACTOR": "1"}, {"MLP_FACTOR": "1", "MLP_FACTOR_SB_FROM": "2"}, {"MLP_FACTOR": "1", "MLP_FACTOR_SB_FROM": "2", "MLP_FACTOR_J": "5"},
            {"MLP_FACTOR": "1", "MLP_FACTOR_BUMP": "16"},
            {"MLP_FACTOR": "1", "MLP_FACTOR_SB": "0"}, {"MLP_FACTOR": "1", "MLP_FACTOR_SB_FROM": "2", "MLP_FACTOR_J": "64", "MLP_NO_GRAPH": "1"}]
bad = 0
t0 = time.time()
for case in range(n_cases):
    fam = ["sparse", "cover", "mixed", "dense", "twophase"][case % 5]
    m = int(rng.integers(40, 500)); n = int(rng.integers(40, 500)); k = int(rng.integers(3, min(n, 25)))
    seed = int(rng.integers(1, 10**6))
    if fam == "sparse":
        if n > m: m, n = n, m
        lp = lpgen.gen_sparse_lp(m, n, k, seed)
    elif fam == "cover":
        lp = lpgen.gen_cover_lp(m, n, k, seed)
    elif fam == "twophase":
        lp = lpgen.gen_twophase_lp(m, n, k, seed)
    elif fam == "mixed":
        lp = lpgen.gen_mixed_lp(m, n, min(k, 8), seed)
    else:
        lp = lpgen.gen_dense_lp(min(m, 150), min(n, 150), seed)
    env = ENVS[case % len(ENVS)]
    for kk in ("MLP_LOWRANK", "MLP_BIGTILE", "MLP_LDPAD", "MLP_BANDED", "MLP_GRAPH_ITERS", "MLP_NO_GRAPH", "MLP_DETERMINISTIC", "MLP_HYPER",
               "MLP_HEAD_APPLY", "MLP_PULL_INSIDE", "MLP_PRIMAL_HEAD_K", "MLP_RATIO_ONE", "MLP_STREAM_BALANCED", "MLP_ORDER_FROM", "MLP_ORDER_EVERY",
               "MLP_SWEEP_PACKED", "MLP_FACTOR", "MLP_FACTOR_SB_FROM", "MLP_FACTOR_J", "MLP_FACTOR_BUMP", "MLP_FACTOR_SB", "MLP_STR_K", "MLP_FPULL"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    try:
        so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
        ostat = "ok"
    except (O.Infeasible, O.Unbounded) as e:
        ostat = type(e).__name__
    try:
        sg = lpgen.build_problem(M.Problem, lp).solve(trace=True)
        gstat = "ok"
    except (M.Infeasible, M.Unbounded) as e:
        gstat = type(e).__name__
    ok = ostat == gstat
    why = "status"
    if ok and ostat == "ok" and not (np.isfinite(sg.objective()) or np.isfinite(so.objective())):
        pass  # the reference algorithm itself ends in a non-finite objective (an unbounded direction): agreement
    elif ok and ostat == "ok":
        ok = obj_close(sg.objective(), so.objective())
        why = "objective %r vs %r" % (sg.objective(), so.objective())
        if ok:
            try:
                check_feasible(lp, sg.values())
            except AssertionError:
                ok = False
                why = "feasibility"
        if ok and fam != "mixed":
            a, b = [t[:5] for t in sg.trace()], [t[:5] for t in so.trace()]
            ok = a == b
            if not ok:
                i = next((j for j, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                why = "trace: first difference at pivot %d of %d / %d" % (i, len(a), len(b))
    if not ok:
        bad += 1
        print("MISMATCH", case, fam, lp["name"], env, ostat, gstat, why, flush=True)
print("cases %d mismatches %d in %.1fs" % (n_cases, bad, time.time() - t0))
