#!/bin/bash
# Round-4 profiles: rocprofv3 --kernel-trace --stats of the three windows of config 4 (early: pivots 32..160 from the slack basis;
# mid / late: from the committed bases) and of the 200 000-row transport solve on the compact factor (pivots 110 000..118 000: 14-20
# levels), summarised into gpurun_out/r04_<window>_kernel_stats.csv; then the PMC traffic passes (tools/pmc_traffic_r04.sh).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for spec in "early 128 32" "mid 256 64" "late 256 32"; do
  set -- $spec
  rm -rf /tmp/prof_$1
  MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o t -- python $ROOT/tools/window_profile.py $1 $2 $3 > $ROOT/gpurun_out/r04_prof_$1.log 2>&1
  python $ROOT/tools/prof_summary.py /tmp/prof_$1 $ROOT/gpurun_out/r04_$1_kernel_stats.csv 16 > /dev/null
done
rm -rf /tmp/prof_fac
MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fac -o t -- python $ROOT/tools/experiments/fac_profile.py 100000 100000 4 0.4 110000 8000 > $ROOT/gpurun_out/r04_prof_factor.log 2>&1
python $ROOT/tools/prof_summary.py /tmp/prof_fac $ROOT/gpurun_out/r04_factor_transport_kernel_stats.csv 16 > /dev/null
bash $ROOT/tools/pmc_traffic_r04.sh > $ROOT/gpurun_out/r04_pmc.log 2>&1
tail -3 $ROOT/gpurun_out/r04_prof_*.log | grep -v "^W\|^E\|Warn"
