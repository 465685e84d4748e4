#!/bin/bash
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -v "^W2026" | tail -5
rm -rf /tmp/prof_mx
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mx -o t -- python $ROOT/tools/experiments/factor_once.py mixed 100000 160000 4 0 10000 > $ROOT/gpurun_out/r05f_mx.log 2>&1 )
python tools/prof_summary.py /tmp/prof_mx gpurun_out/r05f_mixed100k_kernel_stats.csv 16 | head -12
grep -v "^W2026" gpurun_out/r05f_mx.log | tail -5 | cut -c1-300
timeout 300 python tools/experiments/factor_once.py transport 100000 100000 4 0 20000 2>&1 | grep -v Warn | tail -3 | cut -c1-200
