#!/usr/bin/env python3
"""TSP by LP relaxation + subtour cuts + depth-first branch & bound — the driver of BASELINE config 5
(warm-started `add_constraint` / `fix_var` / `clone` on the device-resident basis).

Same method as the reference's example (examples/tsp.rs:211-434): edge variables in [0,1], degree
rows = 2, Stoer–Wagner min cut to find violated subtour rows (cut weight < 2 - 1e-8), most-fractional
branching with `fix_var` on cloned solutions.  Written against the Problem/Solution API only, so it
runs on any backend module that mirrors it (`minilp_amd` on the GPU, or the CPU oracle in tests).

usage: python examples/tsp.py tests/golden/bn130.tsp [--nodes N] [--max-bb-nodes K]
"""
import argparse
import math
import sys
import time

import numpy as np


def read_tsplib(path, limit=None):
    """EUC_2D TSPLIB reader (name, list of (x, y))."""
    name, pts, in_coords = "", [], False
    for line in open(path):
        line = line.strip()
        if not line or line == "EOF":
            continue
        if in_coords:
            parts = line.split()
            pts.append((float(parts[1]), float(parts[2])))
        elif line.startswith("NAME"):
            name = line.split(":")[1].strip()
        elif line.startswith("NODE_COORD_SECTION"):
            in_coords = True
    if limit:
        pts = pts[:limit]
    return name, np.array(pts)


def stoer_wagner(w):
    """Global min cut of a dense symmetric weight matrix.  Returns (weight, boolean side mask).
    Native implementation (libminilp_hip.so: mlp_util_min_cut); `stoer_wagner_py` below is the
    same algorithm in numpy, kept as its executable specification (tests compare the two)."""
    import minilp_amd
    return minilp_amd.min_cut(w)


def stoer_wagner_py(w):
    n = w.shape[0]
    w = w.copy()
    groups = [[i] for i in range(n)]
    active = list(range(n))
    best_w, best_set = math.inf, None
    while len(active) > 1:
        a = active[0]
        added = [a]
        weights = w[a, active].copy()
        in_a = np.zeros(len(active), dtype=bool)
        in_a[0] = True
        prev = a
        last = a
        for _ in range(len(active) - 1):
            weights_masked = np.where(in_a, -np.inf, weights)
            j = int(np.argmax(weights_masked))
            prev, last = last, active[j]
            cut_of_phase = weights[j]
            in_a[j] = True
            added.append(last)
            weights = weights + w[last, active]
        if cut_of_phase < best_w:
            best_w, best_set = cut_of_phase, list(groups[last])
        # merge last into prev
        groups[prev] = groups[prev] + groups[last]
        w[prev, :] += w[last, :]
        w[:, prev] += w[:, last]
        w[prev, prev] = 0.0
        active.remove(last)
    mask = np.zeros(n, dtype=bool)
    mask[best_set] = True
    return float(best_w), mask


class TspSolver:
    def __init__(self, backend, pts, log=None):
        self.B = backend
        self.pts = pts
        self.n = len(pts)
        self.log = log or (lambda *a: None)
        d = pts[:, None, :] - pts[None, :, :]
        self.dist = np.sqrt((d ** 2).sum(axis=2))  # un-rounded Euclidean distance (tsp.rs:19-25)
        self.stats = dict(lp_solves=1, cuts=0, fix_calls=0, clones=0, bb_nodes=0)
        n = self.n
        self.edge_var = -np.ones((n, n), dtype=np.int64)
        p = backend.Problem(backend.MINIMIZE)
        for i in range(n):
            for j in range(i + 1, n):
                v = p.add_var(float(self.dist[i, j]), (0.0, 1.0))
                self.edge_var[i, j] = self.edge_var[j, i] = v
        for i in range(n):
            p.add_constraint([(int(self.edge_var[i, j]), 1.0) for j in range(n) if j != i], backend.EQ, 2.0)
        self.problem = p
        self.iu = np.triu_indices(n, 1)

    def weights(self, sol):
        x = np.asarray(sol.values())
        w = np.zeros((self.n, self.n))
        w[self.iu] = x[self.edge_var[self.iu]]
        return w + w.T

    def add_subtour_constraints(self, sol):  # tsp.rs:398-434
        n = self.n
        while True:
            cut_w, mask = stoer_wagner(self.weights(sol))
            if cut_w > 2.0 - 1e-8:
                return sol
            terms = [(int(self.edge_var[i, j]), 1.0) for i in range(n) for j in range(i) if mask[i] != mask[j]]
            sol = sol.add_constraint(terms, self.B.GE, 2.0)
            self.stats["cuts"] += 1
            self.stats["lp_solves"] += 1

    def relaxation(self):
        """Subtour-elimination LP bound (the cutting-plane loop only)."""
        sol = self.problem.solve()
        return self.add_subtour_constraints(sol)

    @staticmethod
    def choose_branch_var(sol):  # tsp.rs:291-303
        x = np.asarray(sol.values())
        div = np.abs(x - np.round(x))
        v = int(np.argmax(div))
        return v if div[v] > 1e-5 else None

    def solve(self, max_nodes=None):  # tsp.rs:211-392
        sol = self.relaxation()
        self.log("relaxation bound %.6f after %d cuts" % (sol.objective(), self.stats["cuts"]))
        var = self.choose_branch_var(sol)
        if var is None:
            return sol.objective(), self.tour(sol)
        best_cost, best_tour = math.inf, None
        stack = [dict(start=sol, var=var, start_val=0 if sol[var] < 0.5 else 1, cur=None)]
        while stack:
            step = stack[-1]
            if step["cur"] is None:
                step["cur"] = step["start_val"]
            elif step["cur"] == step["start_val"]:
                step["cur"] = 1 - step["cur"]
            else:
                stack.pop()
                continue
            self.stats["bb_nodes"] += 1
            if max_nodes and self.stats["bb_nodes"] > max_nodes:
                break
            cur = step["start"].clone()
            self.stats["clones"] += 1
            try:
                cur = cur.fix_var(step["var"], float(step["cur"]))
                self.stats["fix_calls"] += 1
                self.stats["lp_solves"] += 1
            except self.B.Infeasible:
                continue
            try:
                cur = self.add_subtour_constraints(cur)
            except self.B.Infeasible:
                continue
            obj = cur.objective()
            if obj > best_cost:
                continue
            var = self.choose_branch_var(cur)
            if var is not None:
                stack.append(dict(start=cur, var=var, start_val=0 if cur[var] < 0.5 else 1, cur=None))
            elif obj < best_cost:
                best_cost, best_tour = obj, self.tour(cur)
                self.log("node %d depth %d: new best tour %.6f" % (self.stats["bb_nodes"], len(stack), obj))
        return best_cost, best_tour

    def tour(self, sol):  # tsp.rs:569-587
        x = np.asarray(sol.values())
        n = self.n
        adj = [[] for _ in range(n)]
        for i in range(n):
            for j in range(i + 1, n):
                if x[self.edge_var[i, j]] > 0.5:
                    adj[i].append(j)
                    adj[j].append(i)
        tour, prev, cur = [0], -1, 0
        while True:
            nxt = [a for a in adj[cur] if a != prev]
            if not nxt or nxt[0] == 0 and len(tour) > 1:
                break
            prev, cur = cur, nxt[0]
            if cur == 0:
                break
            tour.append(cur)
        return tour

    def tour_cost(self, tour):
        return float(sum(self.dist[tour[i], tour[(i + 1) % len(tour)]] for i in range(len(tour))))


def tour_to_svg(pts, tour, width=600, margin=50):
    """SVG drawing of a tour, the counterpart of `Tour::to_svg` (tsp.rs:169-208): the points are scaled to a
    `width`-pixel-wide canvas with a `margin`, rounded to whole pixels, and the closed tour is one black 4-pixel path
    that starts (M) at city 0 and visits the others with L commands."""
    pts = np.asarray(pts, dtype=np.float64)
    min_x, max_x = pts[:, 0].min(), pts[:, 0].max()
    min_y, max_y = pts[:, 1].min(), pts[:, 1].max()
    scale = (width - 2 * margin) / (max_x - min_x)

    def rnd(x):  # f64::round: half away from zero (Python's round is half to even)
        return int(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))
    height = rnd((max_y - min_y) * scale) + 2 * margin
    tour = [int(t) for t in tour]
    if 0 in tour:  # the reference's tours start at city 0 (tsp.rs:398-434)
        at = tour.index(0)
        tour = tour[at:] + tour[:at]
    out = ['<?xml version="1.0" encoding="UTF-8" standalone="no"?>\n',
           '<!DOCTYPE svg PUBLIC "-//W3C//DTD SVG 1.1//EN"\n',
           '  "http://www.w3.org/Graphics/SVG/1.1/DTD/svg11.dtd">\n',
           '<svg width="%dpx" height="%dpx" version="1.1"' % (width, height),
           '     xmlns="http://www.w3.org/2000/svg">\n',
           '    <path fill="none" stroke="black" stroke-width="4px" d="\n']
    for i in tour:
        px = rnd((pts[i, 0] - min_x) * scale) + margin
        py = rnd((pts[i, 1] - min_y) * scale) + margin
        out.append("        %s %d %d\n" % ("M" if i == 0 else "L", px, py))
    out += ["        Z\n", '    "/>\n', "</svg>\n"]
    return "".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--nodes", type=int, default=None, help="use only the first N cities")
    ap.add_argument("--max-bb-nodes", type=int, default=None)
    ap.add_argument("--svg", default=None, help="write the tour as an SVG drawing (tsp.rs:169-208)")
    a = ap.parse_args()
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import minilp_amd as backend  # the MI355X engine; there is no CPU back end (tests pass their checker to TspSolver)
    name, pts = read_tsplib(a.file, a.nodes)
    t0 = time.time()
    s = TspSolver(backend, pts, log=lambda m: print("[%.1fs] %s" % (time.time() - t0, m), flush=True))
    cost, tour = s.solve(a.max_bb_nodes)
    print("problem %s (%d cities): tour cost %.10f, %s, %.1fs" % (name, len(pts), cost, s.stats, time.time() - t0))
    print("tour:", " ".join(str(int(t) + 1) for t in tour))  # Tour::to_string (tsp.rs:161-167): 1-based city numbers
    if a.svg:
        with open(a.svg, "w") as f:
            f.write(tour_to_svg(pts, tour))


if __name__ == "__main__":
    main()
