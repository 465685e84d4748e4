#!/bin/bash
# Round 6 — the measurement set of the final tree, one call on one box: the GPU suite, rocprofv3 --kernel-trace --stats of the three
# windows of config 4 and of the compact-factor transport family, the full bench line (timed window, mid / late windows, CPU baseline, full
# solve to the certified optimum, live transport leg).  PMC traffic is a separate call (tools/pmc_traffic_r06.sh).
export TAG=${TAG:-r06f}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd $ROOT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2000 python -m pytest tests -x -q -m "gpu and not slow" > gpurun_out/${TAG}_tests.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_tests.log | tail -3
fi
export TMPDIR=/tmp
for spec in "early 32 5" "mid 256 64" "late 256 32"; do
  set -- $spec
  rm -rf /tmp/prof_$1
  ( cd /tmp && MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o t -- python $ROOT/tools/window_profile.py $1 $2 $3 > $ROOT/gpurun_out/${TAG}_prof_$1.log 2>&1 )
  python tools/prof_summary.py /tmp/prof_$1 gpurun_out/${TAG}_$1_kernel_stats.csv 16 | grep -v "k_inv\|rocclr" | head -12
done
rm -rf /tmp/prof_tr
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o t -- python $ROOT/tools/experiments/factor_once.py transport 100000 100000 4 60000 20000 > $ROOT/gpurun_out/${TAG}_tr.log 2>&1 )
python tools/prof_summary.py /tmp/prof_tr gpurun_out/${TAG}_transport_first60k_kernel_stats.csv 16 | head -14
timeout 200 python tools/experiments/factor_once.py mixed 100000 160000 4 0 20000 2>&1 | tail -1   # config-3 family at 100 000 rows on the compact factor (sparse LU of the bump)
timeout 300 python tools/config_times.py 2>&1 | grep -v Warn | cut -c1-300
timeout 200 python tools/hyper_profile.py 2>&1 | grep -v Warn | grep "MLP_HYPER=1:\|CPU restatement" | cut -c1-260
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json
cp gpurun_out/bench_detail_n1.json gpurun_out/${TAG}_bench_detail.json 2>/dev/null
python - <<'PY'
import json, os
d = json.load(open("gpurun_out/" + os.environ.get("TAG", "r06f") + "_bench_line.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["pivot_level"], {w: (d["windows"][w]["us_per_pivot"]) for w in d["windows"]}, d.get("full_solve"), d["cpu_baseline"].get("gpu_over_cpu_same_window"))
print(d["roofline"].get("ftran"))
print(d.get("factor_transport"))
PY
