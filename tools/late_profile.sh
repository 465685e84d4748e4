#!/bin/bash
# Per-kernel average durations in the LATE regime (last dispatches of a long run): rocprofv3 kernel trace,
# summarised on the box (the raw trace is too large to bring back).  usage: late_profile.sh <chunk> <nchunks> [ENV=VAL ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CH=${1:-4000}; NC=${2:-12}; shift 2
TAG=${TAG:-late}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_late
env "$@" timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_late -o t -- python $ROOT/tools/gpu_perf.py 100000 100000 100 $CH $NC > $ROOT/gpurun_out/${TAG}_run.log 2>&1
python - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/prof_late/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tail = rows[-int(len(rows) * 0.08):]           # the last 8 % of the dispatches
span = (tail[-1][1] - tail[0][0]) * 1e-3
acc = collections.defaultdict(list)
for s, e, k in tail:
    acc[k.split("(")[0][-60:]].append((e - s) * 1e-3)
n_piv = max(len(v) for k, v in acc.items() if "update_pivot" in k)
with open(sys.argv[1], "w") as out:
    out.write("kernel,calls,avg_us,us_per_pivot,share_of_span,min_us,median_us,max_us\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        vs = sorted(v); out.write(f"\"{k}\",{len(v)},{sum(v)/len(v):.2f},{sum(v)/n_piv:.2f},{sum(v)/span:.4f},{vs[0]:.1f},{vs[len(vs)//2]:.1f},{vs[-1]:.1f}\n")
    out.write(f"\"(wall span of the window, us per pivot)\",{n_piv},,{span/n_piv:.2f},1\n")
print(open(sys.argv[1]).read())
PY
tail -3 $ROOT/gpurun_out/${TAG}_run.log | cut -c1-220
