"""Deterministic synthetic LP generators for the BASELINE.json configs.

RNG = SplitMix64 (u -> (u >> 11) * 2^-53), vectorised with numpy so the same bytes feed
every solver (oracle, HIP engine, HiGHS fixture generator).  Instances are returned as a
plain dict in CSR form; `build_problem` replays them through any object exposing the
reference's Problem API (lib.rs:217-283: add_var / add_constraint).
"""
import numpy as np

MINIMIZE, MAXIMIZE = 0, 1
EQ, LE, GE = 0, 1, 2

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, offset=0):
    """n outputs of SplitMix64 seeded with `seed`, starting at stream position `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n, offset=0):
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _stream(seed, tag):
    """Derive an independent sub-seed for a named stream."""
    return int(splitmix64(seed * 1000003 + tag, 1)[0])


def gen_dense_lp(m, n, seed=2):
    """Config 2 family: A_ij ~ U[0,1), b_i = rowsum * U[0.5,1.5), c_j ~ U[0.1,1); Max c'x, Ax<=b, x>=0."""
    A = uniform01(_stream(seed, 1), m * n).reshape(m, n)
    b = A.sum(axis=1) * (0.5 + uniform01(_stream(seed, 2), m))
    c = 0.1 + 0.9 * uniform01(_stream(seed, 3), n)
    indptr = np.arange(0, m * n + 1, n, dtype=np.int64)
    indices = np.tile(np.arange(n, dtype=np.int64), m)
    return dict(name=f"dense_{m}x{n}_s{seed}", direction=MAXIMIZE, m=m, n=n, obj=c,
                lo=np.zeros(n), hi=np.full(n, np.inf), indptr=indptr, indices=indices,
                data=A.reshape(-1).copy(), ops=np.full(m, LE, dtype=np.int32), rhs=b)


def gen_sparse_lp(m, n, k, seed=4):
    """Config 4 family: exactly k distinct columns per row (first k distinct of a 2k-draw stream,
    sorted), values U[0,1), b = rowsum * U[0.5,1.5), c ~ U[0.1,1); Max c'x, Ax<=b, x>=0."""
    assert k <= n
    if k == n:
        return gen_dense_lp(m, n, seed)
    draws = 2 * k + 16
    cand = (splitmix64(_stream(seed, 11), m * draws) % np.uint64(n)).astype(np.int64).reshape(m, draws)
    cols = np.empty((m, k), dtype=np.int64)
    # fast path: rows whose first k candidates are already distinct
    first = np.sort(cand[:, :k], axis=1)
    ok = (first[:, 1:] != first[:, :-1]).all(axis=1)
    cols[ok] = first[ok]
    for r in np.nonzero(~ok)[0]:
        _, idx = np.unique(cand[r], return_index=True)
        keep = np.sort(idx)[:k]
        assert len(keep) == k, "not enough distinct candidates"
        cols[r] = np.sort(cand[r][keep])
    vals = uniform01(_stream(seed, 12), m * k).reshape(m, k)
    b = vals.sum(axis=1) * (0.5 + uniform01(_stream(seed, 13), m))
    c = 0.1 + 0.9 * uniform01(_stream(seed, 14), n)
    return dict(name=f"sparse_{m}x{n}_k{k}_s{seed}", direction=MAXIMIZE, m=m, n=n, obj=c,
                lo=np.zeros(n), hi=np.full(n, np.inf), indptr=np.arange(0, m * k + 1, k, dtype=np.int64),
                indices=cols.reshape(-1), data=vals.reshape(-1), ops=np.full(m, LE, dtype=np.int32), rhs=b)


def gen_cover_lp(m, n, k, seed=5):
    """Covering LP on the pattern of the config-4 family: Min c'x, Ax >= b, x >= 0 with positive data.
    x = 0 is dual feasible (c > 0) and primal infeasible, so the whole solve is the DUAL simplex
    (restore_feasibility with the real objective, solver.rs:513-547); continuous random data, so the
    pivot sequence is non-degenerate."""
    base = gen_sparse_lp(m, n, k, seed)
    out = dict(base)
    out.update(name=f"cover_{m}x{n}_k{k}_s{seed}", direction=MINIMIZE, ops=np.full(m, GE, dtype=np.int32))
    return out


def gen_twophase_lp(m, n, k, seed=6, ge_every=7):
    """Two-phase instance on the pattern of the config-4 family: Max c'x with every `ge_every`-th row turned
    into a >= row, both kinds of right-hand side built around the interior point x0 = 0.3 so that the LP is
    feasible and bounded.  x = 0 is neither primal feasible (the >= rows) nor dual feasible (Max, c > 0):
    the solve is the dual loop on the artificial objective (solver.rs:261, 513-547), recalc_obj_coeffs, then
    the primal loop with steepest edge (solver.rs:470-511); continuous random data, non-degenerate."""
    base = gen_sparse_lp(m, n, k, seed)
    x0 = np.full(n, 0.3)
    lhs = np.add.reduceat(base["data"] * x0[base["indices"]], base["indptr"][:-1])
    u = uniform01(_stream(seed, 31), m)
    ge = (np.arange(m) % ge_every) == 0
    ops = np.where(ge, GE, LE).astype(np.int32)
    rhs = np.where(ge, lhs * (0.5 + 0.4 * u), lhs * (1.2 + 0.8 * u))
    out = dict(base)
    out.update(name=f"twophase_{m}x{n}_k{k}_s{seed}", ops=ops, rhs=rhs)
    return out


def gen_mixed_lp(m, n, k, seed=3):
    """Config 3 stand-in (no NETLIB file is available offline): sparse rows with E/L/G operators,
    finite/infinite/fixed/free bounds and mixed-sign costs, built around a known feasible point so
    the instance is feasible; Minimize.  Starts neither primal- nor dual-feasible, which exercises
    the dual simplex (a7/a8) and the artificial objective (solver.rs:261)."""
    base = gen_sparse_lp(m, n, k, seed)
    u = uniform01(_stream(seed, 21), n)
    x0 = np.floor(4.0 * uniform01(_stream(seed, 22), n))  # feasible point, small integers
    kind = (splitmix64(_stream(seed, 23), n) % np.uint64(10)).astype(np.int64)
    lo = np.where(kind < 6, 0.0, np.where(kind < 8, x0 - 1.0, -np.inf))
    hi = np.where(kind < 4, np.inf, np.where(kind < 8, x0 + 2.0, np.inf))
    fixed = kind == 8
    lo = np.where(fixed, x0, lo)
    hi = np.where(fixed, x0, hi)
    x0 = np.where((kind < 6) | (kind == 9), np.maximum(x0, 0.0), x0)
    # free vars (kind 9) get zero-sum pressure from rows only; costs: mixed sign, free vars cost 0
    c = np.where(kind == 9, 0.0, np.round(4.0 * u - 1.0, 3))
    c = np.where((hi == np.inf) & (c < 0), -c, c)  # keep the LP bounded
    data = np.round(base["data"] * 8.0 - 3.0)      # small integer coefficients, some negative
    data = np.where(data == 0.0, 1.0, data)
    lhs = np.add.reduceat(data * x0[base["indices"]], base["indptr"][:-1])
    opk = (splitmix64(_stream(seed, 24), m) % np.uint64(4)).astype(np.int64)
    ops = np.where(opk == 0, EQ, np.where(opk < 3, LE, GE)).astype(np.int32)
    slack = np.floor(3.0 * uniform01(_stream(seed, 25), m))
    rhs = np.where(ops == EQ, lhs, np.where(ops == LE, lhs + slack, lhs - slack))
    out = dict(base)
    out.update(name=f"mixed_{m}x{n}_k{k}_s{seed}", direction=MINIMIZE, obj=c, lo=lo, hi=hi, data=data, ops=ops, rhs=rhs)
    return out


def build_problem(problem_cls, lp):
    """Replay an instance through the reference's Problem API (any backend)."""
    p = problem_cls(lp["direction"])
    if hasattr(p, "add_vars_bulk"):  # same calls, one FFI crossing
        p.add_vars_bulk(lp["obj"], lp["lo"], lp["hi"])
        p.add_constraints_csr(lp["indptr"], lp["indices"], lp["data"], lp["ops"], lp["rhs"])
        return p
    for j in range(lp["n"]):
        p.add_var(float(lp["obj"][j]), (float(lp["lo"][j]), float(lp["hi"][j])))
    ip, ix, dv = lp["indptr"], lp["indices"], lp["data"]
    for i in range(lp["m"]):
        b, e = int(ip[i]), int(ip[i + 1])
        p.add_constraint_arrays(ix[b:e], dv[b:e], int(lp["ops"][i]), float(lp["rhs"][i]))
    return p


def to_mps(lp, ranges=None):
    """Free-format MPS text of an instance (mps.rs reader).  `ranges`: optional {row: R} entries
    written to a RANGES section (mps.rs:306-321 turns each into a >= row and a <= row).
    Bounds that free-format MPS as read by the reference cannot express (-inf lower with a
    non-negative finite upper) are rejected."""
    m, n = lp["m"], lp["n"]
    out = [f"NAME {lp['name']}", "ROWS", " N COST"]
    tag = {EQ: "E", LE: "L", GE: "G"}
    for i in range(m):
        out.append(f" {tag[int(lp['ops'][i])]} R{i}")
    out.append("COLUMNS")
    ip, ix, dv = lp["indptr"], lp["indices"], lp["data"]
    rows_of = [[] for _ in range(n)]
    for i in range(m):
        for q in range(int(ip[i]), int(ip[i + 1])):
            rows_of[int(ix[q])].append((i, float(dv[q])))
    for j in range(n):
        if lp["obj"][j] != 0.0 or not rows_of[j]:
            out.append(f"    X{j} COST {float(lp['obj'][j])!r}")
        pairs = rows_of[j]
        for t in range(0, len(pairs), 2):  # one or two (row, value) pairs per line
            line = f"    X{j} R{pairs[t][0]} {pairs[t][1]!r}"
            if t + 1 < len(pairs):
                line += f" R{pairs[t + 1][0]} {pairs[t + 1][1]!r}"
            out.append(line)
    out.append("RHS")
    for i in range(m):
        if lp["rhs"][i] != 0.0:
            out.append(f"    RHS R{i} {float(lp['rhs'][i])!r}")
    if ranges:
        out.append("RANGES")
        for i, r in sorted(ranges.items()):
            out.append(f"    RNG R{i} {float(r)!r}")
    out.append("BOUNDS")
    for j in range(n):
        lo, hi = float(lp["lo"][j]), float(lp["hi"][j])
        if lo == -np.inf and hi == np.inf:
            out.append(f" FR BND X{j}")
        elif lo == hi:
            out.append(f" FX BND X{j} {lo!r}")
        elif lo == -np.inf:
            if hi >= 0.0:
                raise ValueError("(-inf, ub>=0] is not expressible for the reference's MPS reader")
            out.append(f" UP BND X{j} {hi!r}")
        else:
            if lo != 0.0:
                out.append(f" LO BND X{j} {lo!r}")
            if hi != np.inf:
                out.append(f" UP BND X{j} {hi!r}")
    out.append("ENDATA")
    return "\n".join(out) + "\n"


def gen_transport_lp(S, D, deg, seed=7, tight=1.0):
    """Generalised transportation (network-with-gains) family — the LARGE-AND-SPARSE-NUCLEUS evidence family of SURVEY §8 f3.
    S supply nodes, D demand nodes, every demand node is linked to `deg` distinct supply nodes: n = D * deg arcs, every column has
    exactly two entries (its supply row and its demand row).  Min c'x,  sum_j g_ij x_ij <= s_i,  sum_i h_ij x_ij >= d_j,  x >= 0
    with continuous random gains, costs, supplies and demands (non-degenerate: pivot sequences are comparable).  x = 0 is dual
    feasible (c > 0) and primal infeasible (the demand rows), so the solve is the dual simplex.  Every basis of such a model is
    a forest of trees / one-cycle components: permutable to triangular form up to a small bump, i.e. its LU has no fill, while
    the explicit inverse of the structural part is dense along every root path — the case a compact factor exists for."""
    n = D * deg
    draws = 2 * deg + 8
    cand = (splitmix64(_stream(seed, 41), D * draws) % np.uint64(S)).astype(np.int64).reshape(D, draws)
    sup = np.empty((D, deg), dtype=np.int64)
    first = np.sort(cand[:, :deg], axis=1)
    ok = (first[:, 1:] != first[:, :-1]).all(axis=1) if deg > 1 else np.ones(D, dtype=bool)
    sup[ok] = first[ok]
    for r in np.nonzero(~ok)[0]:
        _, idx = np.unique(cand[r], return_index=True)
        keep = np.sort(idx)[:deg]
        assert len(keep) == deg, "not enough distinct candidates"
        sup[r] = np.sort(cand[r][keep])
    arc_sup = sup.reshape(-1)                       # arc a = j * deg + t  ->  supply node
    used = np.unique(arc_sup)                       # supply nodes without an arc would be empty rows (dropped by the solver): renumber
    arc_sup = np.searchsorted(used, arc_sup)
    S = len(used)
    arc_dem = np.repeat(np.arange(D, dtype=np.int64), deg)
    g = 0.5 + uniform01(_stream(seed, 42), n)       # gain on the supply row
    h = 0.5 + uniform01(_stream(seed, 43), n)       # gain on the demand row
    c = 1.0 + uniform01(_stream(seed, 44), n)
    d = 1.0 + uniform01(_stream(seed, 45), D)
    load = np.bincount(arc_sup, weights=np.ones(n), minlength=S)
    # supplies: `tight` = 1 leaves about three times the supply the demands need (few supply rows bind at the optimum: shallow
    # basis trees); smaller values make more supply rows bind (deeper trees, more pivots); below ~0.35 instances turn infeasible
    s = (1.0 + uniform01(_stream(seed, 46), S)) * (0.6 + load) * (2.0 * D / max(1.0, float(n))) * 1.5 * tight
    # CSR: S supply rows (arcs sorted by arc index), then D demand rows
    order = np.argsort(arc_sup, kind="stable")
    sup_ptr = np.concatenate(([0], np.cumsum(np.bincount(arc_sup, minlength=S)))).astype(np.int64)
    indptr = np.concatenate((sup_ptr, sup_ptr[-1] + deg * np.arange(1, D + 1, dtype=np.int64)))
    indices = np.concatenate((order, np.arange(n, dtype=np.int64)))
    data = np.concatenate((g[order], h))
    ops = np.concatenate((np.full(S, LE, dtype=np.int32), np.full(D, GE, dtype=np.int32)))
    rhs = np.concatenate((s, d))
    return dict(name=f"transport_{S}x{D}_deg{deg}_s{seed}_t{tight}", direction=MINIMIZE, m=S + D, n=n, obj=c, lo=np.zeros(n),
                hi=np.full(n, np.inf), indptr=indptr, indices=indices, data=data, ops=ops, rhs=rhs)


def gen_staircase_lp(T, R, P, caps, k=3, seed=5):
    """Multi-period production / inventory model (a STAIRCASE: the classical structure whose bases triangularise around small bumps):
    period t has R product-balance rows  sum_j a_ij x_tj + s_(t-1)i - s_ti = d_ti  (P activities, k products each; inventories s link
    period t to t + 1: columns with two entries) and `caps` capacity rows  sum_j b_cj x_tj <= cap_tc  (every activity uses two
    resources).  Minimise production + holding cost; built around a known feasible point; continuous random data (comparable pivot
    sequences).  m = T (R + caps) rows, n = T (P + R) columns."""
    m, n = T * (R + caps), T * (P + R)
    u = lambda tag, cnt: uniform01(_stream(seed, tag), cnt)
    pick = lambda tag, cnt, mod: (splitmix64(_stream(seed, tag), cnt) % np.uint64(mod)).astype(np.int64)
    rows, cols, vals = [], [], []
    prod_rows = pick(31, T * P * k, R).reshape(T, P, k)
    prod_vals = np.round(0.5 + 2.0 * u(32, T * P * k), 3).reshape(T, P, k)
    cap_rows = pick(33, T * P * 2, caps).reshape(T, P, 2)
    cap_vals = np.round(0.2 + 1.0 * u(34, T * P * 2), 3).reshape(T, P, 2)
    for t in range(T):
        r0, c0 = t * (R + caps), t * (P + R)
        for e in range(k):   # activities on the balance rows (duplicates of a row within an activity are merged below)
            rows.append(r0 + prod_rows[t, :, e]); cols.append(c0 + np.arange(P)); vals.append(prod_vals[t, :, e])
        for e in range(2):
            rows.append(r0 + R + cap_rows[t, :, e]); cols.append(c0 + np.arange(P)); vals.append(cap_vals[t, :, e])
        rows.append(r0 + np.arange(R)); cols.append(c0 + P + np.arange(R)); vals.append(-np.ones(R))        # - s_t
        if t + 1 < T:
            rows.append(r0 + (R + caps) + np.arange(R)); cols.append(c0 + P + np.arange(R)); vals.append(np.ones(R))  # + s_t in period t + 1
    import scipy.sparse as sp
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    x0 = np.zeros(n)
    act = u(35, T * P) < 0.35
    lvl = np.round(3.0 * u(36, T * P), 3) * act
    inv = np.round(2.0 * u(37, T * R), 3) * (u(38, T * R) < 0.3)
    for t in range(T):
        c0 = t * (P + R)
        x0[c0:c0 + P] = lvl[t * P:(t + 1) * P]
        x0[c0 + P:c0 + P + R] = inv[t * R:(t + 1) * R]
    lhs = A @ x0
    ops = np.empty(m, np.int32)
    rhs = np.empty(m)
    slack = np.round(1.0 + 4.0 * u(39, m), 3)
    for t in range(T):
        r0 = t * (R + caps)
        ops[r0:r0 + R] = EQ
        rhs[r0:r0 + R] = lhs[r0:r0 + R]
        ops[r0 + R:r0 + R + caps] = LE
        rhs[r0 + R:r0 + R + caps] = lhs[r0 + R:r0 + R + caps] + slack[r0 + R:r0 + R + caps]
    cost = np.empty(n)
    for t in range(T):
        c0 = t * (P + R)
        cost[c0:c0 + P] = np.round(1.0 + 3.0 * u(40 + (t % 7), P), 3) * (1.0 + 0.3 * np.sin(0.37 * t))
        cost[c0 + P:c0 + P + R] = np.round(0.05 + 0.2 * u(50 + (t % 5), R), 3)
    return dict(name=f"staircase_T{T}_R{R}_P{P}_c{caps}_s{seed}", m=m, n=n, direction=MINIMIZE, obj=cost, lo=np.zeros(n), hi=np.full(n, np.inf),
                indptr=A.indptr.astype(np.int64), indices=A.indices.astype(np.int64), data=A.data.astype(np.float64), ops=ops, rhs=rhs)
