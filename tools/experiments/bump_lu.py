"""Evidence for DESIGN 2.8 (CPU, scipy; the oracle only supplies bases): the bump the two-sided peel leaves of the bases a solve walks
through — size, non-zeros, SuperLU fill (COLAMD, threshold 0.1), and a sequential-greedy model of the elimination in rounds of
independent pivots (rounds, L / U non-zeros, longest row).   python tools/experiments/bump_lu.py mixed 100000 160000 4 4500 55000"""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from minilp_amd import lpgen
from oracle import minilp_oracle as O
import basis_structure as BS

def two_sided(Bc, m):
    Br = Bc.tocsr()
    act_r = np.ones(m, bool); act_c = np.ones(m, bool)
    ent_col = np.repeat(np.arange(m), np.diff(Bc.indptr)); ent_row = np.repeat(np.arange(m), np.diff(Br.indptr))
    while True:
        progressed = False
        while True:
            cnt = np.bincount(ent_col, weights=act_r[Bc.indices].astype(float), minlength=m).astype(np.int64)
            cand = np.nonzero((cnt == 1) & act_c)[0]
            if len(cand) == 0: break
            sel = act_r[Bc.indices] & np.isin(ent_col, cand)
            rows, cols = Bc.indices[sel], ent_col[sel]
            u, idx = np.unique(rows, return_index=True)
            act_c[cols[idx]] = False; act_r[u] = False; progressed = True
        while True:
            cnt = np.bincount(ent_row, weights=act_c[Br.indices].astype(float), minlength=m).astype(np.int64)
            cand = np.nonzero((cnt == 1) & act_r)[0]
            if len(cand) == 0: break
            sel = act_c[Br.indices] & np.isin(ent_row, cand)
            cols, rows = Br.indices[sel], ent_row[sel]
            u, idx = np.unique(cols, return_index=True)
            act_r[rows[idx]] = False; act_c[u] = False; progressed = True
        if not progressed: break
    return act_r, act_c

def markowitz_rounds(K):
    """right-looking elimination in rounds of independent pivots chosen by Markowitz count with threshold 0.1; returns rounds, fill"""
    b = K.shape[0]
    rows = [dict() for _ in range(b)]
    K = K.tocoo()
    for i, j, v in zip(K.row, K.col, K.data): rows[i][j] = v
    cols = [set() for _ in range(b)]
    for i in range(b):
        for j in rows[i]: cols[j].add(i)
    act_r = set(range(b)); act_c = set(range(b))
    rounds = 0; nnzL = 0; nnzU = 0; maxrow = 0; sizes = []
    while act_r:
        # candidates: for each active column the entry with min row count among those passing threshold
        cand = []
        for j in act_c:
            if not cols[j]: raise RuntimeError("singular")
            cmax = max(abs(rows[i][j]) for i in cols[j])
            best = None
            for i in cols[j]:
                if abs(rows[i][j]) >= 0.1 * cmax:
                    cost = (len(rows[i]) - 1) * (len(cols[j]) - 1)
                    if best is None or cost < best[0]: best = (cost, i, j)
            cand.append(best)
        cand.sort()
        mincost = cand[0][0]
        used_r = set(); used_c = set(); touched_r = set(); chosen = []
        for cost, i, j in cand:
            if cost > max(4 * mincost, mincost + 4): break
            if i in used_r or j in used_c: continue
            # independence: pivot row i must not contain chosen pivot cols; column j must not contain chosen pivot rows; target rows disjoint
            if any(jj in used_c for jj in rows[i]) or any(ii in used_r for ii in cols[j]): continue
            tgt = cols[j] - {i}
            if tgt & touched_r or i in touched_r: continue
            if any(t in used_r for t in tgt): continue
            chosen.append((i, j)); used_r.add(i); used_c.add(j); touched_r |= tgt
        for i, j in chosen:
            piv = rows[i][j]
            nnzU += len(rows[i])
            for t in list(cols[j]):
                if t == i: continue
                f = rows[t][j] / piv; nnzL += 1
                del rows[t][j]
                for jj, v in rows[i].items():
                    if jj == j: continue
                    if jj in rows[t]: rows[t][jj] -= f * v
                    else: rows[t][jj] = -f * v; cols[jj].add(t)
                maxrow = max(maxrow, len(rows[t]))
            for jj in rows[i]: cols[jj].discard(i)
            cols[j] = set()
            act_r.discard(i); act_c.discard(j)
        rounds += 1; sizes.append(len(chosen))
    return rounds, nnzL, nnzU, maxrow, sizes

fam, a, b, k, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
skip = int(sys.argv[6]) if len(sys.argv) > 6 else 0
# (staircase: T periods, b = R products, k = P activities per period, R // 4 capacity rows)
lp = lpgen.gen_staircase_lp(a, b, k, max(1, b // 4)) if fam == "staircase" else lpgen.gen_mixed_lp(a, b, k, 3)
m, n = lp["m"], lp["n"]
A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n)).tocsc()
Aext = sp.hstack([A, sp.identity(m, format="csc")], format="csc")
s = lpgen.build_problem(O.Problem, lp).solve(budget=0)
if skip: s.continue_solve(skip)
while True:
    s.continue_solve(chunk)
    st = s.stats()
    bv = s.state("basic_vars").astype(np.int64)
    Bc = Aext[:, bv].tocsc()
    ar, ac = two_sided(Bc, m)
    K = Bc[ar][:, ac].tocsc()
    bb = K.shape[0]
    msg = f"pivots {st['pivots']+st['bound_flips']} bump {bb} nnz(K) {K.nnz}"
    if bb > 0:
        lu = spl.splu(K, permc_spec="COLAMD", diag_pivot_thresh=0.1)
        msg += f" splu nnz L {lu.L.nnz} U {lu.U.nnz}"
        t = time.time(); r = markowitz_rounds(K); msg += f" | rounds {r[0]} nnzL {r[1]} nnzU {r[2]} maxrow {r[3]} first sizes {r[4][:8]} last {r[4][-5:]} ({time.time()-t:.1f}s)"
        rc = np.diff(K.tocsr().indptr); cc = np.diff(K.indptr)
        msg += f" | row cnt max {rc.max()} col cnt max {cc.max()}"
    print(msg, flush=True)
    if not s.budget_exhausted: break
