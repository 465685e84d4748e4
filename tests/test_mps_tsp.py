"""Config 3 (MPS path) and config 5 (TSP cutting-plane / branch & bound on the warm-started basis).
CPU part: both MPS readers (oracle = restatement of mps.rs, product = mps.cpp) produce the same
Problem from the same bytes; the min-cut KAT of the reference's TSP example.  GPU part: solves."""
import importlib.util
import math
import os

import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import HIGHS_RTOL, ROOT, check_feasible, highs_cases, obj_close

_spec = importlib.util.spec_from_file_location("tsp_example", os.path.join(ROOT, "examples", "tsp.py"))
tsp = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(tsp)
BN130 = os.path.join(ROOT, "tests", "golden", "bn130.tsp")  # data file of the reference's example (examples/bn130.tsp)
BN130_OPT = 1084.5552622511539  # cost of the tour drawn in examples/bn130.tsp.svg


def same_problem(a, b):
    assert a.variables() == b.variables()
    ca, cb = a.constraints(), b.constraints()
    assert len(ca) == len(cb)
    for x, y in zip(ca, cb):
        assert (x[0] == y[0]).all() and (x[1] == y[1]).all() and x[2] == y[2] and x[3] == y[3]


RANGES_TEXT = """NAME r
ROWS
 N obj
 N free1
 L r1
 G r2
 E r3
 E r4
COLUMNS
    x obj 1 r1 1
    x r2 1 r3 1
    x r4 1 free1 5
    y obj 1 r1 1
    z obj 1 r1 1
RHS
    rhs r1 10 r2 2
    rhs r3 5 r4 5
    other r1 99
RANGES
    rng r1 4 r2 3
    rng r3 2 r4 -2
    other r1 50
BOUNDS
 UP b x 7
 UP b y -3
 FR b z
 UP other x 1
ENDATA
"""


def test_mps_readers_agree_on_ranges_and_bounds_rules():
    fo, fg = O.MpsFile(RANGES_TEXT, O.MINIMIZE), M.MpsFile(RANGES_TEXT, M.MINIMIZE)
    same_problem(fo.problem, fg.problem)
    assert fo.variables == fg.variables == {"x": 0, "y": 1, "z": 2}
    v = fg.problem.variables()
    assert v[0] == (1.0, 0.0, 7.0)            # UP only, positive  -> [0, ub]      (mps.rs:300)
    assert v[1] == (1.0, -math.inf, -3.0)     # UP only, negative  -> (-inf, ub]   (mps.rs:299)
    assert v[2] == (1.0, -math.inf, math.inf)  # FR
    rows = [(op, rhs) for _, _, op, rhs in fg.problem.constraints()]
    # RANGES (mps.rs:306-321): L r1 rhs 10 R 4 -> [6,10]; G r2 rhs 2 R 3 -> [2,5]; E r3 R>0 -> [5,7]; E r4 R<0 -> [3,5]
    assert rows == [(M.GE, 6.0), (M.LE, 10.0), (M.GE, 2.0), (M.LE, 5.0), (M.GE, 5.0), (M.LE, 7.0), (M.GE, 3.0), (M.LE, 5.0)]


@pytest.mark.parametrize("kw", [dict(m=60, n=80, k=4, seed=7), dict(m=400, n=600, k=5, seed=8)], ids=str)
def test_mps_roundtrip_readers_agree(kw):
    lp = lpgen.gen_mixed_lp(**kw)
    text = lpgen.to_mps(lp, ranges={3: 2.5, 5: -1.0})
    fo, fg = O.MpsFile(text, O.MINIMIZE), M.MpsFile(text, M.MINIMIZE)
    same_problem(fo.problem, fg.problem)
    direct = lpgen.build_problem(O.Problem, lp)
    assert fo.problem.variables() == direct.variables()
    plain = O.MpsFile(lpgen.to_mps(lp), O.MINIMIZE)
    assert plain.problem.solve().objective() == direct.solve().objective()  # same bytes in, same pivots out


def test_min_cut_kat():  # examples/tsp.rs:541-564
    w = np.array([[0, 2, 3, 0, 2, 2, 0, 0], [2, 0, 0, 0, 3, 0, 0, 0], [3, 0, 0, 4, 0, 0, 2, 0], [0, 0, 4, 0, 0, 0, 2, 2],
                  [2, 3, 0, 0, 0, 3, 0, 0], [2, 0, 0, 0, 3, 0, 1, 0], [0, 0, 2, 2, 0, 1, 0, 3], [0, 0, 0, 2, 0, 0, 3, 0]], dtype=float)
    weight, mask = tsp.stoer_wagner(w)
    assert weight == 4.0
    side = set(np.nonzero(mask)[0].tolist())
    assert side in ({2, 3, 6, 7}, {0, 1, 4, 5})
    weight_py, mask_py = tsp.stoer_wagner_py(w)
    assert weight_py == 4.0 and (mask_py == mask).all()


def test_min_cut_native_equals_numpy_specification():
    rng = np.random.default_rng(7)
    for n in (2, 3, 9, 40):
        for _ in range(5):
            a = rng.random((n, n)) * (rng.random((n, n)) < 0.4)
            w = np.triu(a, 1)
            w = w + w.T
            wn, mn = tsp.stoer_wagner(w)
            wp, mp = tsp.stoer_wagner_py(w)
            assert wn == wp and (mn == mp).all()


def test_tsp_oracle_small():
    name, pts = tsp.read_tsplib(BN130, 25)
    s = tsp.TspSolver(O, pts)
    cost, tour = s.solve()
    assert sorted(tour) == list(range(25))
    assert abs(s.tour_cost(tour) - cost) < 1e-6


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_config3_mps_solve_matches_oracle_and_highs():
    """Config 3 stand-in: ~10k-variable sparse LP with E/L/G rows and mixed bounds, fed as MPS text."""
    case = [c for c in highs_cases() if c["name"] == "mixed_6000x10000_k4_s3"][0]
    lp = lpgen.gen_mixed_lp(**case["args"])
    text = lpgen.to_mps(lp)
    fg, fo = M.MpsFile.parse(text, M.MINIMIZE), O.MpsFile(text, O.MINIMIZE)
    sg, so = fg.problem.solve(), fo.problem.solve()
    assert obj_close(sg.objective(), so.objective())
    assert obj_close(sg.objective(), case["objective"], HIGHS_RTOL)
    check_feasible(lp, sg.values())
    assert sg.stats()["dual_iters"] > 1000


@pytest.mark.gpu
def test_mps_with_ranges_solve_matches_oracle():
    lp = lpgen.gen_mixed_lp(m=400, n=600, k=5, seed=8)
    text = lpgen.to_mps(lp, ranges={3: 2.5, 5: -1.0, 17: 4.0})
    sg = M.MpsFile.parse(text, M.MINIMIZE).problem.solve()
    so = O.MpsFile(text, O.MINIMIZE).problem.solve()
    assert obj_close(sg.objective(), so.objective())


@pytest.mark.gpu
def test_tsp_relaxation_and_optimum_match_oracle_40_cities():
    """Cutting-plane loop + B&B on the device-resident basis (add_constraint / fix_var / clone).
    The LP vertices differ between backends on this degenerate LP, so the cut sequences differ; the
    subtour-elimination bound and the optimal tour cost are unique and must agree."""
    name, pts = tsp.read_tsplib(BN130, 40)
    sg, so = tsp.TspSolver(M, pts), tsp.TspSolver(O, pts)
    rg, ro = sg.relaxation(), so.relaxation()
    assert abs(rg.objective() - ro.objective()) <= 1e-7 * ro.objective()
    cg, tg = tsp.TspSolver(M, pts).solve()
    co, to = tsp.TspSolver(O, pts).solve()
    assert abs(cg - co) <= 1e-7 * co
    assert sorted(tg) == list(range(40))
    assert cg >= rg.objective() - 1e-7


@pytest.mark.gpu
def test_tsp_bn130_reaches_reference_optimum():
    """BASELINE config 5: examples/bn130.tsp to optimality on the GPU (about a minute)."""
    name, pts = tsp.read_tsplib(BN130)
    s = tsp.TspSolver(M, pts)
    cost, tour = s.solve()
    assert abs(cost - BN130_OPT) < 1e-6
    assert sorted(tour) == list(range(130))
    assert abs(s.tour_cost(tour) - BN130_OPT) < 1e-6
    assert s.stats["cuts"] > 100 and s.stats["clones"] > 100


def test_solve_mps_example_driver_with_the_checker(tmp_path, capsys):  # examples/solve_mps.rs:19-43 counterpart
    """The example's driver code run on the oracle module (the CLI itself only knows the GPU engine)."""
    import sys
    from tests.test_oracle_kat import MPS_TESTPROB
    p = tmp_path / "testprob.mps"
    p.write_text(MPS_TESTPROB)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    import solve_mps
    assert solve_mps.run(O, str(p)) == 0
    out = capsys.readouterr().out
    assert "objective: 54" in out and "XONE = 4" in out and "YTWO = -1" in out and "ZTHREE = 6" in out
    assert "--backend" not in open(os.path.join(root, "examples", "solve_mps.py")).read()
    assert "--backend" not in open(os.path.join(root, "examples", "tsp.py")).read()


@pytest.mark.gpu
def test_solve_mps_example_cli_hip_backend(tmp_path):
    import subprocess
    import sys
    from tests.test_oracle_kat import MPS_TESTPROB
    p = tmp_path / "testprob.mps"
    p.write_text(MPS_TESTPROB)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "solve_mps.py"), str(p)],
                         capture_output=True, text=True, check=True).stdout
    assert "objective: 54" in out and "XONE = 4" in out and "YTWO = -1" in out and "ZTHREE = 6" in out


def test_tour_svg_writer_matches_the_reference_format():
    """`Tour::to_svg` (tsp.rs:169-208): 600-pixel canvas, 50-pixel margin, whole-pixel coordinates, one closed path
    starting at city 0.  Hand-computed for a 4-city rectangle (scale = 500 / 10 = 50, height = round(4 * 50) + 100)."""
    pts = [(0.0, 0.0), (10.0, 0.0), (10.0, 4.0), (0.0, 4.0)]
    svg = tsp.tour_to_svg(pts, [2, 3, 0, 1])
    assert svg == ('<?xml version="1.0" encoding="UTF-8" standalone="no"?>\n'
                   '<!DOCTYPE svg PUBLIC "-//W3C//DTD SVG 1.1//EN"\n'
                   '  "http://www.w3.org/Graphics/SVG/1.1/DTD/svg11.dtd">\n'
                   '<svg width="600px" height="300px" version="1.1"     xmlns="http://www.w3.org/2000/svg">\n'
                   '    <path fill="none" stroke="black" stroke-width="4px" d="\n'
                   '        M 50 50\n        L 550 50\n        L 550 250\n        L 50 250\n        Z\n'
                   '    "/>\n</svg>\n')
