#!/usr/bin/env python3
"""Structure of the bases a solve walks through (CPU, scipy; evidence for SURVEY §8 f3): for the basis after every CHUNK pivots of
the oracle's solve, the size of the STRUCTURAL nucleus (basic columns that are not structural singletons: what the explicit
inverse of minilp_amd holds densely, 8 k^2 bytes), and what an iterated column-singleton peel leaves of it (levels, bump).

    python tools/basis_structure.py transport 20000 20000 4 [chunk]
    python tools/basis_structure.py sparse 20000 20000 12 [chunk]
    python tools/basis_structure.py mixed 30000 50000 4 [chunk]      (the config-3 generator at scale)
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minilp_amd import lpgen  # noqa: E402
from oracle import minilp_oracle as O  # noqa: E402  (analysis tool: the oracle only supplies bases to look at)


def peel(Bc, m):
    """Iterated column-singleton peel of the m x m basis matrix Bc (CSC): returns (levels, peeled, level sizes)."""
    active_row = np.ones(m, bool)
    done = np.zeros(m, bool)
    ent_col = np.repeat(np.arange(m), np.diff(Bc.indptr))
    sizes = []
    while True:
        cnt = np.bincount(ent_col, weights=active_row[Bc.indices].astype(np.float64), minlength=m).astype(np.int64)
        cand = np.nonzero((cnt == 1) & ~done)[0]
        if len(cand) == 0:
            break
        # the single active row of every candidate: the active entries of the candidate columns
        sel = active_row[Bc.indices] & np.isin(ent_col, cand)
        rows, cols = Bc.indices[sel], ent_col[sel]
        u, idx = np.unique(rows, return_index=True)   # two candidates on one row: one of them waits (or the basis is singular)
        done[cols[idx]] = True
        active_row[u] = False
        sizes.append(len(u))
    return len(sizes), int(done.sum()), sizes


def peel_rows(Bc, m):
    """Column-singleton peel to its fixed point, then a ROW-singleton peel of what is left: returns (column levels, row levels, bump)."""
    Br = Bc.tocsr()
    act_r = np.ones(m, bool)
    act_c = np.ones(m, bool)
    ent_col = np.repeat(np.arange(m), np.diff(Bc.indptr))
    ent_row = np.repeat(np.arange(m), np.diff(Br.indptr))
    ncl = nrl = 0
    while True:  # alternate the two peels until neither removes anything (ALT=0: one pass of each)
        progressed = False
        while True:
            cnt = np.bincount(ent_col, weights=act_r[Bc.indices].astype(np.float64), minlength=m).astype(np.int64)
            cand = np.nonzero((cnt == 1) & act_c)[0]
            if len(cand) == 0:
                break
            sel = act_r[Bc.indices] & np.isin(ent_col, cand)
            rows, cols = Bc.indices[sel], ent_col[sel]
            u, idx = np.unique(rows, return_index=True)
            act_c[cols[idx]] = False
            act_r[u] = False
            ncl += 1
            progressed = True
        while True:
            cnt = np.bincount(ent_row, weights=act_c[Br.indices].astype(np.float64), minlength=m).astype(np.int64)
            cand = np.nonzero((cnt == 1) & act_r)[0]
            if len(cand) == 0:
                break
            sel = act_c[Br.indices] & np.isin(ent_row, cand)
            cols, rows = Br.indices[sel], ent_row[sel]
            u, idx = np.unique(cols, return_index=True)
            act_r[rows[idx]] = False
            act_c[u] = False
            nrl += 1
            progressed = True
        if not progressed or os.environ.get("ALT") == "0":
            break
    return ncl, nrl, int(act_c.sum())


def main():
    fam, a, b, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 5000
    lp = lpgen.gen_transport_lp(a, b, k, tight=float(os.environ.get("TIGHT", "1.0"))) if fam == "transport" else lpgen.gen_cover_lp(a, b, k) if fam == "cover" else lpgen.gen_mixed_lp(a, b, k, 3) if fam == "mixed" else lpgen.gen_sparse_lp(a, b, k)
    m, n = lp["m"], lp["n"]
    A = sp.csr_matrix((lp["data"], lp["indices"], lp["indptr"]), shape=(m, n)).tocsc()
    Aext = sp.hstack([A, sp.identity(m, format="csc")], format="csc")
    colnnz = np.diff(Aext.indptr)
    s = lpgen.build_problem(O.Problem, lp).solve(budget=0)
    t_all = time.time()
    while True:
        t0 = time.time()
        s.continue_solve(chunk)
        dt = time.time() - t0
        st = s.stats()
        bv = s.state("basic_vars").astype(np.int64)
        knuc = int((colnnz[bv] != 1).sum())
        lev, peeled, sizes = peel(Aext[:, bv].tocsc(), m)
        if os.environ.get("ROWPEEL"):
            print("    column + row peel: %d column levels, %d row levels, bump %d" % peel_rows(Aext[:, bv].tocsc(), m), flush=True)
        print("pivots %7d  oracle %7.1f us/pivot  structural nucleus k = %6d (dense inverse %8.1f MB)  peel: %4d levels, bump %5d  largest levels %s  lu_nnz %d" % (
            st["pivots"] + st["bound_flips"], dt * 1e6 / chunk, knuc, 8e-6 * knuc * knuc, lev, m - peeled, sorted(sizes, reverse=True)[:4], st["lu_nnz"]), flush=True)
        if not s.budget_exhausted or time.time() - t_all > float(os.environ.get("MAX_S", "600")):
            break
    print("objective %.9f, finished: %s, %.1f s" % (s.objective(), not s.budget_exhausted, time.time() - t_all))


if __name__ == "__main__":
    main()
