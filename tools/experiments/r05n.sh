#!/bin/bash
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
export TMPDIR=/tmp
rm -rf /tmp/prof_mx
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mx -o t -- python $ROOT/tools/experiments/factor_once.py mixed 100000 160000 4 0 10000 > $ROOT/gpurun_out/r05n_mx.log 2>&1 )
python tools/prof_summary.py /tmp/prof_mx gpurun_out/r05n_mixed100k_kernel_stats.csv 16 | head -14
grep "^pivots\|^total" gpurun_out/r05n_mx.log | tail -3 | cut -c1-220
