#!/bin/bash
# HBM traffic of the two dominant kernels from the PMC counters, as MI355X_MICROARCH.md §HBM prescribes:
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only; FETCH_SIZE doubled (gfx950 tallies
# 128-B read requests at 64 B — confirmed for 4/8/16-B/lane streams and for the sweep's stream+gather
# pattern by tools/pmc_calib.sh), WRITE_SIZE as reported (calibrated exact).  Config 4, pivots 0..1200.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_traffic"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/$c"
  MLP_IMPORT_TORCH=1 MLP_NO_REINV=1 timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o p -- python $ROOT/tools/gpu_perf.py 100000 100000 100 300 4 > "$OUT/$c.log" 2>&1
done
python - "$OUT" "$ROOT/gpurun_out/pmc_traffic.json" <<'PY'
import csv, glob, sys, json, collections
out, dst = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # the tableau-row pass is k_sweep, or k_sweep_band + k_band_combine (summed per launch)
        key = "sweep" if ("k_sweep" in k or "k_band_combine" in k) else ("fused" if "k_fused_w" in k else None)
        if key is None:
            continue
        d = res[key].setdefault(c, dict(avg_kb=0.0, launches=len(v), kernel=""))
        d["avg_kb"] += sum(v) / len(v)
        d["kernel"] = (d["kernel"] + " + " if d["kernel"] else "") + k
doc = dict(workload=dict(rows=100000, cols=100000, nnz_per_row=100, seed=4, pivots="0..1200"),
           method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes); traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); see tools/pmc_calib.sh for the calibration",
           kernels={})
for name, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        doc["kernels"][name] = dict(kernel=d["FETCH_SIZE"]["kernel"], launches=d["FETCH_SIZE"]["launches"],
                                    fetch_size_kb=d["FETCH_SIZE"]["avg_kb"], write_size_kb=d["WRITE_SIZE"]["avg_kb"],
                                    hbm_bytes_per_launch=(2.0 * d["FETCH_SIZE"]["avg_kb"] + d["WRITE_SIZE"]["avg_kb"]) * 1024.0)
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
