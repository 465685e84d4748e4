#!/bin/bash
# round 5, second session: where the factor pivot and config 3's dense tail spend their time today
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
export TMPDIR=/tmp
# config 3: hyper path with kernel stats (the multi-kernel launches that remain = the dense tail)
rm -rf /tmp/prof_c3
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o t -- python $ROOT/tools/experiments/cfg3_once.py > $ROOT/gpurun_out/r05c_cfg3.log 2>&1 )
python tools/prof_summary.py /tmp/prof_c3 gpurun_out/r05c_cfg3_kernel_stats.csv 16 | head -24
tail -3 gpurun_out/r05c_cfg3.log
# transport 200k on the factor: first 40 000 pivots, kernel stats
rm -rf /tmp/prof_tr
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o t -- python $ROOT/tools/experiments/factor_once.py transport 100000 100000 4 60000 > $ROOT/gpurun_out/r05c_tr.log 2>&1 )
python tools/prof_summary.py /tmp/prof_tr gpurun_out/r05c_transport_kernel_stats.csv 16 | head -24
tail -5 gpurun_out/r05c_tr.log | cut -c1-400
# config-3 family at 60 000 rows on the factor (bump <= 1024), kernel stats
rm -rf /tmp/prof_mx
( cd /tmp && MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mx -o t -- python $ROOT/tools/experiments/factor_once.py mixed 60000 100000 4 0 10000 > $ROOT/gpurun_out/r05c_mx.log 2>&1 )
python tools/prof_summary.py /tmp/prof_mx gpurun_out/r05c_mixed60k_kernel_stats.csv 16 | head -24
grep -v "^W2026" gpurun_out/r05c_mx.log | tail -8 | cut -c1-400
