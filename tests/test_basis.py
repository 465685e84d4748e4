"""Basis checkpoint (include/minilp_hip.h: mlp_solution_save_basis / mlp_problem_solve_from_basis).

SURVEY.md §8(d) asks for "pivots from a saved mid-solve basis"; the reference has no basis I/O, so the
contract is stated against the uninterrupted run: a mode-2 checkpoint continues pivot for pivot, modes 0/1
(sets only / + f32 weights) recompute x_B and d from the basis and reach the same optimum."""
import struct

import numpy as np
import pytest

import minilp_amd as M
from minilp_amd import lpgen
from oracle import minilp_oracle as O
from tests.common import GEN, OBJ_RTOL, X_ATOL, obj_close

pytestmark = pytest.mark.gpu

CASES = [("sparse", (300, 240, 10, 31), 120), ("sparse", (500, 500, 20, 32), 260), ("dense", (60, 50, 33), 25),
         ("cover", (300, 300, 12, 34), 90), ("twophase", (400, 400, 14, 35), 60), ("twophase", (400, 400, 14, 35), 400),
         ("mixed", (250, 300, 5, 36), 80)]


def _key(t):
    return t[:5]


@pytest.mark.parametrize("fam,args,cut", CASES)
def test_full_checkpoint_continues_pivot_for_pivot(fam, args, cut):
    lp = GEN[fam](*args)
    prob = lpgen.build_problem(M.Problem, lp)
    ref = prob.solve(trace=True)                      # the uninterrupted run
    full = ref.trace()
    if len(full) <= cut:
        pytest.skip("instance solves before the cut")
    s = prob.solve(budget=cut, trace=True)
    cut = len(s.trace())   # a phase switch inside the budget spends one unit on its (pivot-free) decision record
    blob = s.save_basis(2)
    del s
    t = prob.solve_from_basis(blob, trace=True)       # new Solver, basis installed, nucleus re-inverted from A
    tail = t.trace()
    assert [_key(x) for x in tail] == [_key(x) for x in full[cut:]]
    assert obj_close(t.objective(), ref.objective(), OBJ_RTOL)
    assert np.abs(t.values() - ref.values()).max() <= X_ATOL
    # the oracle agrees on the whole sequence (the same check the uninterrupted run passes)
    so = lpgen.build_problem(O.Problem, lp).solve(trace=True)
    assert [_key(x) for x in so.trace()[cut:]] == [_key(x) for x in tail]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("fam,args,cut", [CASES[1], CASES[3], CASES[5]])
def test_compact_checkpoint_reaches_the_same_optimum(fam, args, cut, mode):
    lp = GEN[fam](*args)
    prob = lpgen.build_problem(M.Problem, lp)
    ref = prob.solve()
    s = prob.solve(budget=cut)
    blob = s.save_basis(mode)
    x_mid = s.values()
    del s
    t0 = prob.solve_from_basis(blob, budget=0)        # no pivot yet: x_B recomputed from the basis
    assert np.abs(t0.values() - x_mid).max() <= 1e-9 * max(1.0, np.abs(x_mid).max())
    t0.continue_solve(-1)
    assert obj_close(t0.objective(), ref.objective(), OBJ_RTOL)
    assert np.abs(t0.values() - ref.values()).max() <= X_ATOL


def test_compact_checkpoint_with_f32_weights_follows_the_same_path_for_a_while():
    # mode 1 keeps the pricing weights to f32 accuracy: on a non-degenerate instance the pricing decisions
    # right after the cut are those of the uninterrupted run
    lp = GEN["sparse"](500, 500, 20, 32)
    prob = lpgen.build_problem(M.Problem, lp)
    full = prob.solve(trace=True).trace()
    cut = 260
    s = prob.solve(budget=cut)
    t = prob.solve_from_basis(s.save_basis(1), budget=20, trace=True)
    assert [_key(x) for x in t.trace()] == [_key(x) for x in full[cut:cut + 20]]


def test_checkpoint_of_another_model_is_refused():
    lp = GEN["sparse"](300, 240, 10, 31)
    prob = lpgen.build_problem(M.Problem, lp)
    blob = prob.solve(budget=50).save_basis(0)
    other = lpgen.build_problem(M.Problem, GEN["sparse"](280, 240, 10, 31))
    with pytest.raises(M.InternalError) as e:
        other.solve_from_basis(blob)
    assert e.value.code == -1 and "kept rows" in str(e.value)
    with pytest.raises(M.InternalError):
        prob.solve_from_basis(blob[:100])
    bad = bytearray(blob)
    bad[0:8] = b"NOTBASIS"
    with pytest.raises(M.InternalError):
        prob.solve_from_basis(bytes(bad))
    # a corrupted set (variable 0 listed twice) is not a basis
    hdr = 56
    bad = bytearray(blob)
    first, second = struct.unpack_from("<ii", blob, hdr)
    struct.pack_into("<ii", bad, hdr, first, first)
    with pytest.raises(M.InternalError):
        prob.solve_from_basis(bytes(bad))


def test_checkpoint_after_optimality_reloads_as_optimal():
    lp = GEN["sparse"](300, 240, 10, 31)
    prob = lpgen.build_problem(M.Problem, lp)
    ref = prob.solve()
    for mode in (0, 1, 2):
        t = prob.solve_from_basis(ref.save_basis(mode), trace=True)
        assert len(t.trace()) == 0
        assert obj_close(t.objective(), ref.objective(), OBJ_RTOL)
        # the reloaded solution is a live Solution: warm-start mutators work on it
        t2 = t.add_constraint([(0, 1.0), (1, 1.0)], M.LE, 0.5 * (ref[0] + ref[1]) - 1e-3)
        r2 = ref.clone().add_constraint([(0, 1.0), (1, 1.0)], M.LE, 0.5 * (ref[0] + ref[1]) - 1e-3)
        assert obj_close(t2.objective(), r2.objective(), 1e-8)


def test_blob_contents_are_validated_not_trusted():
    """a blob's flags, values and weights are checked or re-derived on load (layout: 56-byte header, int32
    basic_vars[m], nb_vars[n], uint8 flags[n] padded to 8, f64 x_N[n], then f32 gamma[n], beta[m] in mode 1)."""
    lp = lpgen.gen_sparse_lp(300, 240, 10, 31)
    m, n = lp["m"], lp["n"]
    prob = lpgen.build_problem(M.Problem, lp)
    ref = prob.solve()
    s = prob.solve(budget=100)
    blob = bytearray(s.save_basis(1))
    pad8 = lambda x: (x + 7) & ~7
    off_flags = 56 + pad8(4 * m) + pad8(4 * n)
    off_xn = off_flags + pad8(n)
    off_gamma = off_xn + 8 * n
    # (1) garbage at-min / at-max flags: re-derived from x_N against the bounds, the solve reaches the same optimum
    bad = bytearray(blob)
    for c in range(n):
        bad[off_flags + c] = (bad[off_flags + c] & 4) | (3 - (bad[off_flags + c] & 3))
    t = prob.solve_from_basis(bytes(bad))
    assert obj_close(t.objective(), ref.objective())
    # (2) the "dual feasible" bit set on a basis that is not: re-derived from the recomputed reduced costs
    bad = bytearray(blob)
    flags_word = struct.unpack_from("<I", bad, 32)[0]
    struct.pack_into("<I", bad, 32, flags_word | 2)
    t = prob.solve_from_basis(bytes(bad))
    assert obj_close(t.objective(), ref.objective())
    # (3) a non-basic value outside its bounds, a NaN weight, a negative weight: refused
    for patch in (lambda b: struct.pack_into("<d", b, off_xn, -1.0),
                  lambda b: struct.pack_into("<f", b, off_gamma, float("nan")),
                  lambda b: struct.pack_into("<f", b, off_gamma + 4, -2.0)):
        bad = bytearray(blob)
        patch(bad)
        with pytest.raises(M.InternalError):
            prob.solve_from_basis(bytes(bad))
    # (4) an unaligned copy of the blob (the f32 weights are read with memcpy)
    raw = bytearray(b"\0" * 3) + blob
    t = prob.solve_from_basis(bytes(memoryview(raw)[3:]))
    assert obj_close(t.objective(), ref.objective())
