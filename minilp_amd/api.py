"""ctypes binding of libminilp_hip.so with the reference's names and argument meaning
(lib.rs:61-464: OptimizationDirection, ComparisonOp, Problem, Solution, Error; mps.rs: MpsFile)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libminilp_hip.so")
# the sharded solve maps peer mailboxes through HIP IPC: this driver stack only supports the dmabuf flavour
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MINIMIZE, MAXIMIZE = 0, 1          # lib.rs:61-68 OptimizationDirection
EQ, LE, GE = 0, 1, 2               # lib.rs:160-169 ComparisonOp


class Infeasible(Exception):       # lib.rs:175 Error::Infeasible
    pass


class Unbounded(Exception):        # lib.rs:177 Error::Unbounded
    pass


class InternalError(Exception):    # where the reference panics (or a HIP error / missing GPU)
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class MlpStats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("iterations", "basis_changes", "bound_flips", "primal_iters", "dual_iters",
                                            "reinversions", "num_constraints", "num_total_vars", "nucleus_size",
                                            "nucleus_capacity", "nnz")]
                + [(n, C.c_double) for n in ("fused_bytes", "fused_ms", "sweep_bytes", "sweep_ms")]
                + [(n, C.c_uint64) for n in ("fused_launches", "sweep_launches")]
                + [("solve_wall_s", C.c_double), ("kase", C.c_uint64 * 5), ("update_ms", C.c_double),
                   ("update_launches", C.c_uint64), ("banded_sweep", C.c_uint64), ("final_refreshes", C.c_uint64), ("max_pivot_err", C.c_double),
                   ("ftran_bytes", C.c_double), ("ftran_ms", C.c_double), ("ftran_launches", C.c_uint64),
                   ("iter_ms", C.c_double), ("iter_samples", C.c_uint64), ("beta_rebuilds", C.c_uint64),
                   ("fold_bytes", C.c_double), ("fold_ms", C.c_double), ("fold_launches", C.c_uint64),
                   ("dense_ftran_bytes", C.c_double), ("dense_ftran_ms", C.c_double), ("dense_ftran_launches", C.c_uint64),
                   ("str_ms", C.c_double), ("str_launches", C.c_uint64),
                   ("hyper_iters", C.c_uint64), ("hyper_bails", C.c_uint64), ("ratio_stalls", C.c_uint64),
                   ("reinversion_fallbacks", C.c_uint64),
                   ("factor_active", C.c_uint64), ("factor_refactors", C.c_uint64), ("factor_levels", C.c_uint64),
                   ("factor_switches", C.c_uint64), ("factor_bump", C.c_uint64), ("factor_bump_max", C.c_uint64)])  # appended in ABI version 4 (the struct only grows at its end from here on)


ABI_VERSION = 4  # include/minilp_hip.h: MLP_ABI_VERSION


class MlpIterInfo(C.Structure):  # include/minilp_hip.h: mlp_iter_info
    _fields_ = [("status", C.c_int32), ("phase", C.c_int32), ("next_stage", C.c_int32), ("reserved", C.c_int32),
                ("col", C.c_int64), ("row", C.c_int64), ("entering_var", C.c_int64), ("leaving_var", C.c_int64),
                ("pivot_coeff", C.c_double), ("step", C.c_double), ("objective", C.c_double), ("nucleus_size", C.c_uint64)]


STAGE_FTRAN, STAGE_RATIO, STAGE_BTRAN, STAGE_BASIS, STAGE_ROW, STAGE_APPLY = range(6)
ITER_PIVOT, ITER_FLIP, ITER_OPTIMAL, ITER_UNBOUNDED, ITER_FEASIBLE, ITER_INFEASIBLE, ITER_SINGULAR = range(7)

_lib = None


def lib_path():
    return _SO


def lib():
    """Load the HIP extension.  Fails loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(f"{_SO} is missing: build it with `python -m minilp_amd.build` (hipcc, gfx950). "
                          "minilp_amd has no CPU fallback.")
    L = C.CDLL(_SO)
    vp, u64, i64, u32, dbl, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_double, C.c_int
    pu32, pdbl = C.POINTER(C.c_uint32), C.POINTER(C.c_double)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("mlp_abi_version", u32)
    sig("mlp_stats_size", u64)
    if L.mlp_abi_version() != ABI_VERSION or L.mlp_stats_size() != C.sizeof(MlpStats):
        raise ImportError(f"{_SO} was built from another version of include/minilp_hip.h (ABI {L.mlp_abi_version()}, mlp_stats "
                          f"{L.mlp_stats_size()} bytes; this binding: ABI {ABI_VERSION}, {C.sizeof(MlpStats)} bytes): rebuild it")
    sig("mlp_last_error", C.c_char_p)
    sig("mlp_device_count", i32)
    sig("mlp_set_device", i32, i32)
    sig("mlp_problem_new", vp, i32)
    sig("mlp_problem_clone", vp, vp)
    sig("mlp_problem_free", None, vp)
    sig("mlp_problem_add_var", u32, vp, dbl, dbl, dbl)
    sig("mlp_problem_num_vars", u32, vp)
    sig("mlp_problem_add_constraint", i32, vp, pu32, pdbl, u64, i32, dbl)
    sig("mlp_problem_add_vars", i32, vp, u64, pdbl, pdbl, pdbl)
    sig("mlp_problem_add_constraints_csr", i32, vp, u64, C.POINTER(C.c_uint64), pu32, pdbl, C.POINTER(C.c_int32), pdbl)
    sig("mlp_problem_solve", i32, vp, C.POINTER(vp))
    sig("mlp_problem_num_constraints", u64, vp)
    sig("mlp_problem_var", i32, vp, u32, pdbl, pdbl, pdbl)
    sig("mlp_problem_constraint", u64, vp, u64, pu32, pdbl, u64, C.POINTER(i32), pdbl)
    sig("mlp_problem_solve_ex", i32, vp, C.POINTER(vp), i64, u32)
    sig("mlp_solution_continue", i32, vp, i64)
    sig("mlp_solution_save_basis", u64, vp, i32, vp, u64)
    sig("mlp_problem_solve_from_basis", i32, vp, vp, u64, C.POINTER(vp), i64, u32)
    sig("mlp_solution_set_sampling", i32, vp, i32)
    sig("mlp_solution_budget_exhausted", i32, vp)
    sig("mlp_solution_reinvert", i32, vp, pdbl)
    sig("mlp_solution_recompute_basic_values", i32, vp)
    sig("mlp_solution_enable_sharding", i32, vp, i32, i32, C.c_char_p)
    sig("mlp_solution_enable_sharding_ex", i32, vp, i32, i32, C.c_char_p, C.c_char_p, vp)
    sig("mlp_rccl_unique_id", i32, vp)
    sig("mlp_solution_transport", C.c_char_p, vp)
    sig("mlp_solution_clone", vp, vp)
    sig("mlp_solution_free", None, vp)
    sig("mlp_solution_objective", dbl, vp)
    sig("mlp_solution_num_vars", u32, vp)
    sig("mlp_solution_var_value", i32, vp, u32, pdbl)
    sig("mlp_solution_values", i32, vp, pdbl, u32)
    sig("mlp_solution_add_constraint", i32, C.POINTER(vp), pu32, pdbl, u64, i32, dbl)
    sig("mlp_solution_fix_var", i32, C.POINTER(vp), u32, dbl)
    sig("mlp_solution_unfix_var", i32, C.POINTER(vp), u32, C.POINTER(i32))
    sig("mlp_solution_add_gomory_cut", i32, C.POINTER(vp), u32)
    sig("mlp_solution_stats", None, vp, C.POINTER(MlpStats))
    sig("mlp_solution_reset_stats", None, vp)
    sig("mlp_solution_trace_len", u64, vp)
    sig("mlp_solution_trace_get", None, vp, u64, C.POINTER(C.c_int32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
        C.POINTER(i64), pdbl, pdbl)
    sig("mlp_solution_state", u64, vp, C.c_char_p, pdbl, u64)
    sig("mlp_mps_parse", i32, C.c_char_p, u64, i32, C.POINTER(vp))
    sig("mlp_mps_free", None, vp)
    sig("mlp_mps_name", C.c_char_p, vp)
    sig("mlp_mps_num_vars", u32, vp)
    sig("mlp_mps_var_name", C.c_char_p, vp, u32)
    sig("mlp_mps_var_index", i64, vp, C.c_char_p)
    sig("mlp_mps_problem", vp, vp)
    sig("mlp_util_min_cut", C.c_double, u32, pdbl, C.POINTER(C.c_uint8))
    sig("mlp_engine_open", i32, vp, C.POINTER(MlpIterInfo))
    sig("mlp_engine_stage", i32, vp, i32, C.POINTER(MlpIterInfo))
    _lib = L
    return L


def device_count():
    return lib().mlp_device_count()


def rccl_unique_id():
    """128-byte ncclUniqueId (rank 0 makes it, the launcher distributes it) for Solution.enable_sharding_ex(..., "rccl", id)."""
    buf = (C.c_char * 128)()
    _raise(lib().mlp_rccl_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


def set_device(device):
    _raise(lib().mlp_set_device(int(device)))


def _raise(st):
    if st == 0:
        return
    if st == 1:
        raise Infeasible("problem is infeasible")
    if st == 2:
        raise Unbounded("problem is unbounded")
    raise InternalError(st, lib().mlp_last_error().decode())


class LinearExpr:
    """lib.rs:84-158: a linear expression, built term by term (`LinearExpr.empty().add(x, 1.0)`) or from any
    iterable of (variable, coefficient) pairs.  Every method that takes an expression accepts either form."""

    def __init__(self, terms=()):
        self.vars, self.coeffs = [], []
        for var, coeff in terms:
            self.add(var, coeff)

    @classmethod
    def empty(cls):  # lib.rs:92
        return cls()

    def add(self, var, coeff):  # lib.rs:98
        self.vars.append(int(var))
        self.coeffs.append(float(coeff))
        return self

    def __iter__(self):
        return iter(zip(self.vars, self.coeffs))

    def __len__(self):
        return len(self.vars)


def _terms(expr):
    pairs = list(expr)
    idx = np.ascontiguousarray([int(p[0]) for p in pairs], dtype=np.uint32)
    val = np.ascontiguousarray([float(p[1]) for p in pairs], dtype=np.float64)
    return idx, val


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def min_cut(weights):
    """Stoer-Wagner global minimum cut of a dense symmetric weight matrix (host helper of the TSP
    driver, include/minilp_hip.h).  Returns (cut weight, boolean side mask)."""
    w = np.ascontiguousarray(weights, dtype=np.float64)
    n = w.shape[0]
    side = np.zeros(n, dtype=np.uint8)
    val = lib().mlp_util_min_cut(n, _p(w, C.c_double), _p(side, C.c_uint8))
    return float(val), side.astype(bool)


class Problem:
    """lib.rs:193-305.  `Problem(direction)`, `add_var(obj_coeff, (min, max)) -> Variable index`,
    `add_constraint(expr, cmp_op, rhs)` with expr an iterable of (variable, coeff), `solve()`."""

    def __init__(self, direction, _h=None):
        self._h = _h if _h is not None else lib().mlp_problem_new(int(direction))
        self.direction = direction

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mlp_problem_free(self._h)
            self._h = None

    def clone(self):
        return Problem(self.direction, lib().mlp_problem_clone(self._h))

    @property
    def num_vars(self):
        return lib().mlp_problem_num_vars(self._h)

    def add_var(self, obj_coeff, bounds):
        return int(lib().mlp_problem_add_var(self._h, obj_coeff, bounds[0], bounds[1]))

    def add_constraint(self, expr, cmp_op, rhs):
        idx, val = _terms(expr)
        _raise(lib().mlp_problem_add_constraint(self._h, _p(idx, C.c_uint32), _p(val, C.c_double), len(idx), cmp_op, rhs))

    def add_constraint_arrays(self, idx, val, cmp_op, rhs):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        _raise(lib().mlp_problem_add_constraint(self._h, _p(idx, C.c_uint32), _p(val, C.c_double), len(idx), cmp_op, rhs))

    def add_vars_bulk(self, obj_coeffs, mins, maxs):
        """n x add_var in one call; returns the index of the first new variable."""
        first = self.num_vars
        o = np.ascontiguousarray(obj_coeffs, dtype=np.float64)
        a = np.ascontiguousarray(mins, dtype=np.float64)
        b = np.ascontiguousarray(maxs, dtype=np.float64)
        _raise(lib().mlp_problem_add_vars(self._h, len(o), _p(o, C.c_double), _p(a, C.c_double), _p(b, C.c_double)))
        return first

    def add_constraints_csr(self, indptr, indices, data, cmp_ops, rhs):
        """m x add_constraint in one call (rows in CSR form)."""
        ip = np.ascontiguousarray(indptr, dtype=np.uint64)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        dv = np.ascontiguousarray(data, dtype=np.float64)
        ops = np.ascontiguousarray(cmp_ops, dtype=np.int32)
        rh = np.ascontiguousarray(rhs, dtype=np.float64)
        _raise(lib().mlp_problem_add_constraints_csr(self._h, len(rh), _p(ip, C.c_uint64), _p(ix, C.c_uint32), _p(dv, C.c_double),
                                                     _p(ops, C.c_int32), _p(rh, C.c_double)))

    def variables(self):
        """[(obj_coeff, min, max)] as given to add_var."""
        o, a, b = C.c_double(), C.c_double(), C.c_double()
        res = []
        for v in range(self.num_vars):
            _raise(lib().mlp_problem_var(self._h, v, C.byref(o), C.byref(a), C.byref(b)))
            res.append((o.value, a.value, b.value))
        return res

    def constraints(self):
        """[(vars, coeffs, cmp_op, rhs)] with terms sorted by variable (sprs CsVec order, lib.rs:279)."""
        res = []
        op, rhs = C.c_int(), C.c_double()
        for c in range(lib().mlp_problem_num_constraints(self._h)):
            k = lib().mlp_problem_constraint(self._h, c, None, None, 0, C.byref(op), C.byref(rhs))
            idx = np.zeros(k, dtype=np.uint32)
            val = np.zeros(k, dtype=np.float64)
            lib().mlp_problem_constraint(self._h, c, _p(idx, C.c_uint32), _p(val, C.c_double), k, C.byref(op), C.byref(rhs))
            res.append((idx, val, op.value, rhs.value))
        return res

    def solve(self, budget=-1, trace=False, profile=False):
        out = C.c_void_p()
        _raise(lib().mlp_problem_solve_ex(self._h, C.byref(out), budget, (1 if trace else 0) | (2 if profile else 0)))
        return Solution(out)

    def solve_from_basis(self, blob, budget=-1, trace=False, profile=False):
        """Build the solver for this problem, install a basis saved by `Solution.save_basis` and continue."""
        blob = bytes(blob)
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        out = C.c_void_p()
        _raise(lib().mlp_problem_solve_from_basis(self._h, C.cast(buf, C.c_void_p), len(blob), C.byref(out), budget,
                                                  (1 if trace else 0) | (2 if profile else 0)))
        return Solution(out)


class Solution:
    """lib.rs:313-424.  Mutators consume self like the Rust receivers and return the new Solution;
    on error the device-resident solver is freed (lib.rs:359, 385)."""

    def __init__(self, h):
        self._h = C.c_void_p(h.value if isinstance(h, C.c_void_p) else h)

    def __del__(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().mlp_solution_free(self._h)
            self._h = C.c_void_p()

    def _take(self):
        h = C.c_void_p(self._h.value)
        self._h = C.c_void_p()
        return h

    def clone(self):
        h = lib().mlp_solution_clone(self._h)
        if not h:
            raise InternalError(-3, lib().mlp_last_error().decode())
        return Solution(C.c_void_p(h))

    def objective(self):
        return lib().mlp_solution_objective(self._h)

    @property
    def num_vars(self):
        return lib().mlp_solution_num_vars(self._h)

    def var_value(self, var):
        out = C.c_double()
        _raise(lib().mlp_solution_var_value(self._h, int(var), C.byref(out)))
        return out.value

    __getitem__ = var_value

    def values(self):
        a = np.zeros(self.num_vars, dtype=np.float64)
        _raise(lib().mlp_solution_values(self._h, _p(a, C.c_double), len(a)))
        return a

    def __iter__(self):  # lib.rs:350 iter()
        return iter(enumerate(self.values()))

    def add_constraint(self, expr, cmp_op, rhs):
        idx, val = _terms(expr)
        h = self._take()
        _raise(lib().mlp_solution_add_constraint(C.byref(h), _p(idx, C.c_uint32), _p(val, C.c_double), len(idx), cmp_op, rhs))
        return Solution(h)

    def fix_var(self, var, val):
        h = self._take()
        _raise(lib().mlp_solution_fix_var(C.byref(h), int(var), val))
        return Solution(h)

    def unfix_var(self, var):
        h = self._take()
        was = C.c_int()
        _raise(lib().mlp_solution_unfix_var(C.byref(h), int(var), C.byref(was)))
        return Solution(h), bool(was.value)

    def add_gomory_cut(self, var):
        h = self._take()
        _raise(lib().mlp_solution_add_gomory_cut(C.byref(h), int(var)))
        return Solution(h)

    def save_basis(self, mode=2):
        """Basis checkpoint as bytes (include/minilp_hip.h): 0 sets + flags + x_N, 1 + f32 weights, 2 full f64 state."""
        n = lib().mlp_solution_save_basis(self._h, int(mode), None, 0)
        if n == 0:
            raise InternalError(-1, lib().mlp_last_error().decode())
        buf = (C.c_char * n)()
        if lib().mlp_solution_save_basis(self._h, int(mode), C.cast(buf, C.c_void_p), n) != n:
            raise InternalError(-1, lib().mlp_last_error().decode())
        return bytes(buf)

    def set_sampling(self, every_iteration):
        """True: every iteration is an eager, event-bracketed one; False: default cadence; None: sampling off for good."""
        _raise(lib().mlp_solution_set_sampling(self._h, -1 if every_iteration is None else (1 if every_iteration else 0)))

    # --- engine-level controls (fixed pivot budget, stats, trace)
    def continue_solve(self, budget):
        _raise(lib().mlp_solution_continue(self._h, budget))

    @property
    def budget_exhausted(self):
        return bool(lib().mlp_solution_budget_exhausted(self._h))

    def enable_sharding(self, rank, world, shm_name):
        """Column-block sharding of the pricing path (include/minilp_hip.h); see minilp_amd.dist."""
        _raise(lib().mlp_solution_enable_sharding(self._h, int(rank), int(world), shm_name.encode()))

    def enable_sharding_ex(self, rank, world, shm_name, transport, rccl_id=None):
        """enable_sharding with the transport named: "peer", "host", "rccl" (rccl_id: rank 0's 128-byte id) or "pump"."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(rccl_id)) if rccl_id else None
        _raise(lib().mlp_solution_enable_sharding_ex(self._h, int(rank), int(world), shm_name.encode(), transport.encode(),
                                                      C.cast(buf, C.c_void_p) if buf is not None else None))

    def transport(self):
        return lib().mlp_solution_transport(self._h).decode()

    # ---- engine-level stepping (include/minilp_hip.h: mlp_engine_open / mlp_engine_stage)
    def engine_open(self):
        """Pricing decision of the phase initial_solve would run next; returns (status, info dict)."""
        info = MlpIterInfo()
        st = lib().mlp_engine_open(self._h, C.byref(info))
        if st < 0:
            _raise(st)
        return st, {n: getattr(info, n) for n, _ in MlpIterInfo._fields_}

    def engine_stage(self, stage):
        """One stage (STAGE_*) of the open iteration; returns (status, info dict)."""
        info = MlpIterInfo()
        st = lib().mlp_engine_stage(self._h, int(stage), C.byref(info))
        if st < 0:
            _raise(st)
        return st, {n: getattr(info, n) for n, _ in MlpIterInfo._fields_}

    def reinvert(self):
        d = C.c_double()
        _raise(lib().mlp_solution_reinvert(self._h, C.byref(d)))
        return d.value

    def recompute_basic_values(self):
        """x_B = B^-1 (b - N x_N) from the basis, two refinement steps (solver.rs:1177-1197)."""
        _raise(lib().mlp_solution_recompute_basic_values(self._h))

    def stats(self):
        s = MlpStats()
        lib().mlp_solution_stats(self._h, C.byref(s))
        d = {n: getattr(s, n) for n, _ in MlpStats._fields_}
        d["kase"] = list(d["kase"])
        return d

    def reset_stats(self):
        lib().mlp_solution_reset_stats(self._h)

    def trace(self):
        n = lib().mlp_solution_trace_len(self._h)
        out = []
        ph, col, row, ev, lv = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        pc, ob = C.c_double(), C.c_double()
        for i in range(n):
            lib().mlp_solution_trace_get(self._h, i, C.byref(ph), C.byref(col), C.byref(row), C.byref(ev), C.byref(lv),
                                         C.byref(pc), C.byref(ob))
            out.append((ph.value, col.value, row.value, ev.value, lv.value, pc.value, ob.value))
        return out

    def state(self, what):
        n = lib().mlp_solution_state(self._h, what.encode(), None, 0)
        if n == 2 ** 64 - 1:
            raise KeyError(what)
        a = np.zeros(n, dtype=np.float64)
        lib().mlp_solution_state(self._h, what.encode(), _p(a, C.c_double), n)
        return a


class MpsFile:
    """mps.rs:7-16; MpsFile.parse(text, direction) mirrors MpsFile::parse (mps.rs:39)."""

    def __init__(self, text, direction):
        if isinstance(text, str):
            text = text.encode()
        h = C.c_void_p()
        st = lib().mlp_mps_parse(text, len(text), int(direction), C.byref(h))
        if st != 0:
            raise ValueError(lib().mlp_last_error().decode())  # io::ErrorKind::InvalidData
        self._h = h
        self.problem_name = lib().mlp_mps_name(h).decode()
        n = lib().mlp_mps_num_vars(h)
        self.variables = {lib().mlp_mps_var_name(h, i).decode(): i for i in range(n)}
        self.problem = Problem(direction, lib().mlp_mps_problem(h))

    parse = classmethod(lambda cls, text, direction: cls(text, direction))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mlp_mps_free(self._h)
