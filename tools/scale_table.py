#!/usr/bin/env python3
"""One table from the bench lines of a 1/2/4/8-GPU run (tools/first_8gpu_run.sh step 4).

  python tools/scale_table.py DIR 1 2 4 8      reads DIR/bench_n<N>.json (the ONE line bench.py printed)

`value` of an N > 1 line is a replica window by construction (deferred sharding: the driver's 20 pivots from the slack basis are
latency-bound and run unsharded on every rank), so the scaling evidence is the late window: us per pivot with the pricing path
sharded over column blocks and the streaming pass over row strips, and the pricing path (tableau-row sweep + update + pricing scan)
against the unsharded run of the same invocation — north_star's ">= 3x at 8 GPUs" is about that last column."""
import json
import os
import sys


def load(d, n):
    p = os.path.join(d, f"bench_n{n}.json")
    try:
        lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def f(x, nd=1):
    return "—" if x is None else (f"{x:.{nd}f}" if isinstance(x, float) else str(x))


def main():
    d = sys.argv[1]
    ns = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
    rows = []
    for n in ns:
        r = load(d, n)
        if r is None:
            rows.append((n, "no line", *["—"] * 8))
            continue
        w = r.get("windows") or {}
        late1 = (w.get("late") or {}).get("us_per_pivot")
        ls = r.get("late_sharded") or {}
        un = r.get("unsharded_same_run") or {}
        rk = r.get("ranks") or {}
        rows.append((n, r.get("scaling"), f(r.get("value")), f(r.get("value_vs_1gpu"), 3),
                     f(ls.get("us_per_pivot") if n > 1 else late1), f(un.get("late_us_per_pivot") if n > 1 else late1),
                     f(ls.get("pricing_path_us") if n > 1 else None), f(un.get("late_pricing_path_us") if n > 1 else None),
                     f(r.get("pricing_speedup_vs_1gpu"), 2),
                     f"{rk.get('distinct_devices', 1)} dev" + (", oversubscribed" if rk.get("oversubscribed") else "")))
    head = ("N", "scaling", "value (pivots/s, timed window)", "value vs 1 GPU", "late us/pivot (sharded)", "late us/pivot (unsharded, same run)",
            "late pricing path us (sharded)", "(unsharded)", "pricing speed-up vs 1 GPU", "devices")
    print("| " + " | ".join(head) + " |")
    print("|" + "---|" * len(head))
    for row in rows:
        print("| " + " | ".join(str(x) for x in row) + " |")


if __name__ == "__main__":
    main()
