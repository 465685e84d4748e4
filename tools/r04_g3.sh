set -x
mkdir -p gpurun_out/r04
( MLP_PB_DET=1 timeout 300 python tools/experiments/nan_probe.py ) > gpurun_out/r04/g3_nan1.log 2>&1
( MLP_PB_DET=0 timeout 300 python tools/experiments/nan_probe.py ) > gpurun_out/r04/g3_nan0.log 2>&1
( timeout 600 python tools/experiments/shard_xb_diff.py ) > gpurun_out/r04/g3_xb.log 2>&1
grep -v Warn gpurun_out/r04/g3_nan1.log | tail -12; grep -v Warn gpurun_out/r04/g3_nan0.log | tail -12; grep pivot gpurun_out/r04/g3_xb.log | head -60
