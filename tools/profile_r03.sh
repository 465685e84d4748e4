#!/bin/bash
# Round-3 profiles: rocprofv3 --kernel-trace --stats of the three windows of bench.py (early: pivots 32..160 from the slack
# basis; mid / late: from the committed bases), summarised into gpurun_out/r03_<window>_kernel_stats.csv, then the PMC traffic passes.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for spec in "early 128 32" "mid 256 64" "late 256 32"; do
  set -- $spec
  rm -rf /tmp/prof_$1
  MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o t -- python $ROOT/tools/window_profile.py $1 $2 $3 > $ROOT/gpurun_out/r03_prof_$1.log 2>&1
  python $ROOT/tools/prof_summary.py /tmp/prof_$1 $ROOT/gpurun_out/r03_$1_kernel_stats.csv 16 > /dev/null
done
bash $ROOT/tools/pmc_traffic_r03.sh > $ROOT/gpurun_out/r03_pmc.log 2>&1
