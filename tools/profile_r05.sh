#!/bin/bash
# Round 5: the GPU suite, rocprofv3 --kernel-trace --stats of the three windows of config 4 (early: pivots 5..37 from the
# slack basis = the driver-timed window and the batch after it; mid / late: from the committed bases), configs 2 / 3 / 5 wall times,
# and the full bench line (timed window, mid / late windows, CPU baseline, full solve to the certified optimum, transport solve).
export TAG=${TAG:-r05b}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd $ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_tests.log | tail -3
export TMPDIR=/tmp
for spec in "early 32 5" "mid 256 64" "late 256 32"; do
  set -- $spec
  rm -rf /tmp/prof_$1
  ( cd /tmp && MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o t -- python $ROOT/tools/window_profile.py $1 $2 $3 > $ROOT/gpurun_out/${TAG}_prof_$1.log 2>&1 )
  python tools/prof_summary.py /tmp/prof_$1 gpurun_out/${TAG}_$1_kernel_stats.csv 16 | grep -v "k_inv\|rocclr" | head -14
done
timeout 300 python tools/config_times.py 2>&1 | grep -v Warn | cut -c1-300
timeout 300 python tools/tsp_profile.py tests/golden/bn130.tsp 2>&1 | grep -v Warn | tail -3 | cut -c1-300
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json
cp gpurun_out/bench_detail_n1.json gpurun_out/${TAG}_bench_detail.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/" + __import__("os").environ.get("TAG", "r05a") + "_bench_line.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["pivot_level"], {w: (d["windows"][w]["us_per_pivot"]) for w in d["windows"]}, d.get("full_solve"), d["cpu_baseline"].get("gpu_over_cpu_same_window"))
PY
