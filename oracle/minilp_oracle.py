"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It mirrors the reference's public API (lib.rs:61-464: Problem / Solution /
ComparisonOp / OptimizationDirection / Error) on top of oracle/libminilp_oracle.so,
the single-threaded C++ restatement of minilp 0.2.2.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libminilp_oracle.so")

MINIMIZE, MAXIMIZE = 0, 1
EQ, LE, GE = 0, 1, 2


class Infeasible(Exception):
    pass


class Unbounded(Exception):
    pass


class OraclePanic(Exception):
    pass


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "minilp_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-B"] if force else ["make", "-C", _HERE])
    return _SO


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "pivots", "bound_flips", "refactors", "primal_iters", "dual_iters", "ftran", "btran",
        "ftran_lu_nnz", "btran_lu_nnz", "eta_nnz_applied", "row_sweep_nnz",
        "lu_nnz", "eta_nnz", "eta_count", "num_constraints", "num_total_vars")] + [("t_refactor", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, u64, i64, dbl, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_double, C.c_int
    pu32, pdbl, pu64 = C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint64)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_last_error", C.c_char_p)
    sig("orc_problem_new", vp, i32)
    sig("orc_problem_clone", vp, vp)
    sig("orc_problem_free", None, vp)
    sig("orc_problem_add_var", u64, vp, dbl, dbl, dbl)
    sig("orc_problem_num_vars", u64, vp)
    sig("orc_problem_add_constraint", i32, vp, pu32, pdbl, u64, i32, dbl)
    sig("orc_problem_solve", i32, vp, C.POINTER(vp))
    sig("orc_problem_solve_ex", i32, vp, C.POINTER(vp), i64, i32)
    sig("orc_problem_try_new", i32, vp, C.POINTER(vp))
    sig("orc_solution_continue", i32, vp, i64)
    sig("orc_solution_budget_exhausted", i32, vp)
    sig("orc_solution_clone", vp, vp)
    sig("orc_solution_free", None, vp)
    sig("orc_solution_objective", dbl, vp)
    sig("orc_solution_num_vars", u64, vp)
    sig("orc_solution_var_value", i32, vp, u64, pdbl)
    sig("orc_solution_add_constraint", i32, C.POINTER(vp), pu32, pdbl, u64, i32, dbl)
    sig("orc_solution_fix_var", i32, C.POINTER(vp), u64, dbl)
    sig("orc_solution_unfix_var", i32, C.POINTER(vp), u64, C.POINTER(i32))
    sig("orc_solution_add_gomory_cut", i32, C.POINTER(vp), u64)
    sig("orc_solution_stats", None, vp, C.POINTER(OrcStats))
    sig("orc_solution_trace_len", u64, vp)
    sig("orc_solution_trace_get", None, vp, u64, C.POINTER(C.c_int32), C.POINTER(i64), C.POINTER(i64),
        C.POINTER(i64), C.POINTER(i64), pdbl, pdbl)
    sig("orc_solution_state", u64, vp, C.c_char_p, pdbl, u64)
    sig("orc_solution_set_capture", None, vp, C.c_int)
    sig("orc_lu_factorize", i32, u64, pu64, pu64, pdbl, dbl, C.POINTER(vp))
    sig("orc_lu_free", None, vp)
    sig("orc_lu_nnz", u64, vp)
    sig("orc_lu_get_factor", u64, vp, i32, i32, pu64, pu64, pdbl, pdbl)
    sig("orc_lu_has_diag", i32, vp, i32, i32)
    sig("orc_lu_get_perms", None, vp, i32, pu64, pu64, pu64, pu64)
    sig("orc_lu_solve_dense", None, vp, i32, pdbl)
    sig("orc_lu_solve_sparse", u64, vp, i32, pu64, pdbl, u64, pu64, pdbl)
    sig("orc_sparse_transpose", None, u64, u64, pu64, pu64, pdbl, pu64, pu64, pdbl)
    sig("orc_mps_parse", i32, C.c_char_p, u64, i32, C.POINTER(vp))
    sig("orc_mps_free", None, vp)
    sig("orc_mps_name", C.c_char_p, vp)
    sig("orc_mps_num_vars", u64, vp)
    sig("orc_mps_var_name", C.c_char_p, vp, u64)
    sig("orc_mps_var_index", i64, vp, C.c_char_p)
    sig("orc_mps_problem", vp, vp)
    sig("orc_problem_num_constraints", u64, vp)
    sig("orc_problem_var", None, vp, u64, pdbl, pdbl, pdbl)
    sig("orc_problem_constraint", u64, vp, u64, pu32, pdbl, u64, C.POINTER(i32), pdbl)
    _lib = L
    return L


def _raise(st):
    if st == 0:
        return
    if st == 1:
        raise Infeasible("problem is infeasible")
    if st == 2:
        raise Unbounded("problem is unbounded")
    raise OraclePanic(lib().orc_last_error().decode())


def _terms(expr):
    """expr: iterable of (var, coeff) pairs (lib.rs:131-158 LinearExpr conversions)."""
    pairs = list(expr)
    idx = np.ascontiguousarray([int(p[0]) for p in pairs], dtype=np.uint32)
    val = np.ascontiguousarray([float(p[1]) for p in pairs], dtype=np.float64)
    return idx, val, len(pairs)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Problem:
    """lib.rs:193-305"""

    def __init__(self, direction, _h=None):
        self._h = _h if _h is not None else lib().orc_problem_new(int(direction))
        self.direction = direction

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_problem_free(self._h)
            self._h = None

    def clone(self):
        return Problem(self.direction, lib().orc_problem_clone(self._h))

    @property
    def num_vars(self):
        return lib().orc_problem_num_vars(self._h)

    def add_var(self, obj_coeff, bounds):
        return int(lib().orc_problem_add_var(self._h, obj_coeff, bounds[0], bounds[1]))

    def add_constraint(self, expr, op, rhs):
        idx, val, k = _terms(expr)
        _raise(lib().orc_problem_add_constraint(self._h, _p(idx, C.c_uint32), _p(val, C.c_double), k, op, rhs))

    def add_constraint_arrays(self, idx, val, op, rhs):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        _raise(lib().orc_problem_add_constraint(self._h, _p(idx, C.c_uint32), _p(val, C.c_double), len(idx), op, rhs))

    def solve(self, budget=-1, trace=False):
        out = C.c_void_p()
        _raise(lib().orc_problem_solve_ex(self._h, C.byref(out), budget, int(trace)))
        return Solution(out)

    def try_new(self):
        out = C.c_void_p()
        _raise(lib().orc_problem_try_new(self._h, C.byref(out)))
        return Solution(out)

    # raw access so tests can feed the identical problem to the product
    def variables(self):
        o, a, b = C.c_double(), C.c_double(), C.c_double()
        res = []
        for v in range(self.num_vars):
            lib().orc_problem_var(self._h, v, C.byref(o), C.byref(a), C.byref(b))
            res.append((o.value, a.value, b.value))
        return res

    def constraints(self):
        res = []
        n = lib().orc_problem_num_constraints(self._h)
        op, rhs = C.c_int(), C.c_double()
        for c in range(n):
            k = lib().orc_problem_constraint(self._h, c, None, None, 0, C.byref(op), C.byref(rhs))
            idx = np.zeros(k, dtype=np.uint32)
            val = np.zeros(k, dtype=np.float64)
            lib().orc_problem_constraint(self._h, c, _p(idx, C.c_uint32), _p(val, C.c_double), k, C.byref(op), C.byref(rhs))
            res.append((idx, val, op.value, rhs.value))
        return res


class Solution:
    """lib.rs:313-424.  Mutators consume self (like the Rust `self` receivers) and return the new
    Solution; on error the underlying solver is freed, as in the reference."""

    def __init__(self, h):
        self._h = C.c_void_p(h.value if isinstance(h, C.c_void_p) else h)

    def __del__(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().orc_solution_free(self._h)
            self._h = C.c_void_p()

    def _take(self):
        h = C.c_void_p(self._h.value)
        self._h = C.c_void_p()
        return h

    def clone(self):
        return Solution(C.c_void_p(lib().orc_solution_clone(self._h)))

    def objective(self):
        return lib().orc_solution_objective(self._h)

    @property
    def num_vars(self):
        return lib().orc_solution_num_vars(self._h)

    def var_value(self, var):
        out = C.c_double()
        _raise(lib().orc_solution_var_value(self._h, int(var), C.byref(out)))
        return out.value

    __getitem__ = var_value

    def __iter__(self):
        for v in range(self.num_vars):
            yield v, self.var_value(v)

    def values(self):
        return np.array([self.var_value(v) for v in range(self.num_vars)])

    def add_constraint(self, expr, op, rhs):
        idx, val, k = _terms(expr)
        h = self._take()
        _raise(lib().orc_solution_add_constraint(C.byref(h), _p(idx, C.c_uint32), _p(val, C.c_double), k, op, rhs))
        return Solution(h)

    def fix_var(self, var, val):
        h = self._take()
        _raise(lib().orc_solution_fix_var(C.byref(h), int(var), val))
        return Solution(h)

    def unfix_var(self, var):
        h = self._take()
        was = C.c_int()
        _raise(lib().orc_solution_unfix_var(C.byref(h), int(var), C.byref(was)))
        return Solution(h), bool(was.value)

    def add_gomory_cut(self, var):
        h = self._take()
        _raise(lib().orc_solution_add_gomory_cut(C.byref(h), int(var)))
        return Solution(h)

    # --- instrumentation
    def set_capture(self, on=True):
        """Keep the two extra solves of every pivot (v, tau) and rho for the per-stage differential tests."""
        lib().orc_solution_set_capture(self._h, 1 if on else 0)

    def continue_solve(self, budget):
        _raise(lib().orc_solution_continue(self._h, budget))

    @property
    def budget_exhausted(self):
        return bool(lib().orc_solution_budget_exhausted(self._h))

    def stats(self):
        s = OrcStats()
        lib().orc_solution_stats(self._h, C.byref(s))
        return {n: getattr(s, n) for n, _ in OrcStats._fields_}

    def trace(self):
        n = lib().orc_solution_trace_len(self._h)
        out = []
        ph, col, row, ev, lv = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        pc, ob = C.c_double(), C.c_double()
        for i in range(n):
            lib().orc_solution_trace_get(self._h, i, C.byref(ph), C.byref(col), C.byref(row), C.byref(ev), C.byref(lv),
                                         C.byref(pc), C.byref(ob))
            out.append((ph.value, col.value, row.value, ev.value, lv.value, pc.value, ob.value))
        return out

    def state(self, what):
        n = lib().orc_solution_state(self._h, what.encode(), None, 0)
        if n == 2 ** 64 - 1:
            raise KeyError(what)
        a = np.zeros(n, dtype=np.float64)
        lib().orc_solution_state(self._h, what.encode(), _p(a, C.c_double), n)
        return a


class LU:
    """lu.rs: lu_factorize + LUFactors for the white-box KATs."""

    def __init__(self, size, indptr, rows, vals, stability):
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self.size = size
        h = C.c_void_p()
        st = lib().orc_lu_factorize(size, _p(indptr, C.c_uint64), _p(rows, C.c_uint64), _p(vals, C.c_double), stability, C.byref(h))
        if st == -2:
            raise ArithmeticError("SingularMatrix")
        _raise(st)
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_lu_free(self._h)

    def nnz(self):
        return lib().orc_lu_nnz(self._h)

    def factor(self, which, transp=False):
        """Dense (size x size) off-diagonal matrix + diag vector of L (which=0) or U (which=1)."""
        n = self.size
        nnz = lib().orc_lu_get_factor(self._h, int(transp), which, None, None, None, None)
        indptr = np.zeros(n + 1, dtype=np.uint64)
        rows = np.zeros(max(nnz, 1), dtype=np.uint64)
        vals = np.zeros(max(nnz, 1), dtype=np.float64)
        diag = np.zeros(n, dtype=np.float64)
        lib().orc_lu_get_factor(self._h, int(transp), which, _p(indptr, C.c_uint64), _p(rows, C.c_uint64), _p(vals, C.c_double), _p(diag, C.c_double))
        dense = np.zeros((n, n))
        for c in range(n):
            for p in range(int(indptr[c]), int(indptr[c + 1])):
                dense[int(rows[p]), c] = vals[p]
        has_diag = bool(lib().orc_lu_has_diag(self._h, int(transp), which))
        return dense, (diag if has_diag else None)

    def perms(self, transp=False):
        n = self.size
        a = [np.zeros(n, dtype=np.uint64) for _ in range(4)]
        lib().orc_lu_get_perms(self._h, int(transp), *[_p(x, C.c_uint64) for x in a])
        return dict(row_new2orig=a[0], col_new2orig=a[1], row_orig2new=a[2], col_orig2new=a[3])

    def solve_dense(self, rhs, transp=False):
        x = np.ascontiguousarray(rhs, dtype=np.float64).copy()
        lib().orc_lu_solve_dense(self._h, int(transp), _p(x, C.c_double))
        return x

    def solve_sparse(self, idx, val, transp=False):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        oi = np.zeros(self.size, dtype=np.uint64)
        ov = np.zeros(self.size, dtype=np.float64)
        k = lib().orc_lu_solve_sparse(self._h, int(transp), _p(idx, C.c_uint64), _p(val, C.c_double), len(idx), _p(oi, C.c_uint64), _p(ov, C.c_double))
        return oi[:k].astype(np.int64), ov[:k]


def sparse_transpose(n_rows, indptr, rows, vals):
    indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    n_cols = len(indptr) - 1
    oi = np.zeros(n_rows + 1, dtype=np.uint64)
    ox = np.zeros(len(rows), dtype=np.uint64)
    od = np.zeros(len(rows), dtype=np.float64)
    lib().orc_sparse_transpose(n_rows, n_cols, _p(indptr, C.c_uint64), _p(rows, C.c_uint64), _p(vals, C.c_double),
                               _p(oi, C.c_uint64), _p(ox, C.c_uint64), _p(od, C.c_double))
    return oi, ox, od


class MpsFile:
    """mps.rs:7-16, MpsFile::parse (mps.rs:39)."""

    def __init__(self, text, direction):
        if isinstance(text, str):
            text = text.encode()
        h = C.c_void_p()
        st = lib().orc_mps_parse(text, len(text), int(direction), C.byref(h))
        if st != 0:
            raise ValueError(lib().orc_last_error().decode())
        self._h = h
        self.problem_name = lib().orc_mps_name(h).decode()
        n = lib().orc_mps_num_vars(h)
        self.variables = {lib().orc_mps_var_name(h, i).decode(): i for i in range(n)}
        self.problem = Problem(direction, lib().orc_mps_problem(h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_mps_free(self._h)
