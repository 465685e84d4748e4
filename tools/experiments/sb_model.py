"""Model of the sparse bump factor (rounds of independent pivots, Luby bidding) + the four pull-form solves."""
import sys, numpy as np, scipy.sparse as sp
def factor(K, substeps=1, tau=0.1):
    b = K.shape[0]; Kr = K.tocsr(); Kc = K.tocsc()
    rows = [list(zip(Kr.indices[Kr.indptr[u]:Kr.indptr[u+1]].tolist(), Kr.data[Kr.indptr[u]:Kr.indptr[u+1]].tolist())) for u in range(b)]
    rows = [[list(e) for e in r] for r in rows]
    colrows = [Kc.indices[Kc.indptr[s]:Kc.indptr[s+1]].tolist() for s in range(b)]
    rowstate = [0]*b; colstate = [0]*b; pivcol = [-1]*b; pivrow = [-1]*b; piv = [0.0]*b
    L = [[] for _ in range(b)]
    nleft = b; rnd = 0; sizes = []
    def find(u, s):
        for k, e in enumerate(rows[u]):
            if e[0] == s: return k
        return -1
    while nleft > 0:
        rnd += 1
        cand = {}
        for s in range(b):
            if colstate[s]: continue
            act = [u for u in colrows[s] if rowstate[u] == 0]
            vals = [abs(rows[u][find(u, s)][1]) for u in act]
            cmax = max(vals)
            best = None
            for u, a in zip(act, vals):
                if a >= tau * cmax:
                    key = ((len(rows[u]) - 1) * (len(act) - 1), len(rows[u]), u)
                    if best is None or key < best: best = key
            cand[s] = (best[0], best[2], act)
        mincost = min(c[0] for c in cand.values())
        limit = max(4 * mincost, mincost + 4)
        taken = set(); winners = []
        for sub in range(substeps):
            claim = {}
            bidders = [s for s in cand if cand[s][0] <= limit and s not in [w for w in winners] and not (set(cand[s][2]) & taken)]
            for s in bidders:
                key = (cand[s][0], s)
                for t in cand[s][2]:
                    if t not in claim or key < claim[t]: claim[t] = key
            neww = [s for s in bidders if all(claim[t] == (cand[s][0], s) for t in cand[s][2])]
            for s in neww: taken |= set(cand[s][2])
            winners += neww
        for s in winners:
            u = cand[s][1]; p = rows[u][find(u, s)][1]
            for t in cand[s][2]:
                if t == u: continue
                k = find(t, s); f = rows[t][k][1] / p
                L[t].append((u, f))
                rows[t][k] = rows[t][-1]; rows[t].pop()
                for c, v in rows[u]:
                    if c == s: continue
                    kk = find(t, c)
                    if kk >= 0: rows[t][kk][1] -= f * v
                    else: rows[t].append([c, -f * v]); colrows[c].append(t)
        for s in winners:
            u = cand[s][1]; rowstate[u] = rnd; colstate[s] = rnd; pivcol[u] = s; pivrow[s] = u; piv[u] = rows[u][find(u, s)][1]
        nleft -= len(winners); sizes.append(len(winners))
    # records
    U = {}; UT = {}; LT = {}
    for u in range(b):
        s = pivcol[u]
        U[u] = [(c, v) for c, v in rows[u] if c != s]
        UT[u] = sorted((u2, rows[u2][find(u2, s)][1]) for u2 in colrows[s] if u2 != u and rowstate[u2] < rowstate[u] and find(u2, s) >= 0)
        LT[u] = sorted((t, dict(L[t])[u]) for t in colrows[s] if rowstate[t] > rowstate[u] and u in dict(L[t]))
    return dict(b=b, rounds=rnd, sizes=sizes, rowstate=rowstate, pivcol=pivcol, piv=piv, L=L, U=U, UT=UT, LT=LT, maxrow=max(len(r) for r in rows), maxcol=max(len(c) for c in colrows), maxL=max(len(l) for l in L))
def ftran(F, t):
    b = F["b"]; Y = t.copy(); X = np.zeros(b)
    order = sorted(range(b), key=lambda u: F["rowstate"][u])
    for u in order: Y[u] = Y[u] - sum(f * Y[u2] for u2, f in F["L"][u])
    for u in reversed(order): X[F["pivcol"][u]] = (Y[u] - sum(v * X[c] for c, v in F["U"][u])) / F["piv"][u]
    return X
def btran(F, c):
    b = F["b"]; W = np.zeros(b)
    order = sorted(range(b), key=lambda u: F["rowstate"][u])
    for u in order: W[u] = (c[F["pivcol"][u]] - sum(v * W[u2] for u2, v in F["UT"][u])) / F["piv"][u]
    for u in reversed(order): W[u] = W[u] - sum(f * W[t] for t, f in F["LT"][u])
    return W
if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    # cycles + a few chords
    perm = rng.permutation(b)
    K = sp.lil_matrix((b, b))
    for i in range(b):
        K[i, i] = rng.uniform(0.5, 2) * rng.choice([-1, 1]); K[i, perm[i]] += rng.uniform(0.5, 2)
    for _ in range(b // 10): K[rng.integers(b), rng.integers(b)] = rng.uniform(-1, 1)
    K = sp.csr_matrix(K)
    for ss in (1, 2, 3):
        F = factor(K, ss)
        t = rng.normal(size=b)
        x = ftran(F, t); y = btran(F, t)
        print(f"substeps {ss}: rounds {F['rounds']} sizes {F['sizes'][:10]} maxrow {F['maxrow']} maxcol {F['maxcol']} maxL {F['maxL']} ftran err {np.abs(K @ x - t).max():.2e} btran err {np.abs(K.T @ y - t).max():.2e}")
