#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box (separate PMC passes, kernel-trace only).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_calib"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
"$ROOT/tools/pmc_calib" > "$OUT/bytes.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- "$ROOT/tools/pmc_calib" > "$OUT/$c.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(c, k, "per-dispatch avg (KB):", sum(v) / len(v), "n=", len(v))
PY
cat "$OUT/bytes.txt"
