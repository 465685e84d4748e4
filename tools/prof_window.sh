#!/bin/bash
# rocprofv3 kernel statistics of one window of config 4 (tools/window_profile.py):  tools/prof_window.sh TAG {early|mid|late} PIVOTS WARM [ENV=VAL ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; W=$2; P=$3; WARM=$4; shift 4
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
( cd /tmp && env "$@" MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- python $ROOT/tools/window_profile.py $W $P $WARM > $ROOT/gpurun_out/${TAG}_prof.log 2>&1 )
tail -1 $ROOT/gpurun_out/${TAG}_prof.log
python $ROOT/tools/prof_summary.py /tmp/prof_$TAG $ROOT/gpurun_out/${TAG}_kernel_stats.csv 18 | grep -v "k_inv\|rocclr" | head -16
