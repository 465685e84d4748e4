#!/bin/bash
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -v "^W2026" | tail -2
rm -rf /tmp/prof_tr
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o t -- python $ROOT/tools/experiments/factor_once.py transport 100000 100000 4 60000 20000 > $ROOT/gpurun_out/r05m_tr.log 2>&1 )
python tools/prof_summary.py /tmp/prof_tr gpurun_out/r05m_transport_first60k_kernel_stats.csv 16 | head -6
timeout 300 python tools/experiments/factor_once.py transport 100000 100000 4 0 50000 2>&1 | grep -v Warn | tail -1
timeout 300 python tools/experiments/factor_once.py mixed 100000 160000 4 0 20000 2>&1 | grep -v Warn | tail -1
