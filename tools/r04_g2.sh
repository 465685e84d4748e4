set -x
mkdir -p gpurun_out/r04
( MLP_PB_DET=1 timeout 900 python -m pytest tests/test_late_regime.py -x -q -m gpu -k "stepped" 2>&1 | tail -40 ) > gpurun_out/r04/g2_tests_det.log 2>&1
( timeout 900 python tools/shard_bitwise.py 2 3000 1000 default ) > gpurun_out/r04/g2_bitwise_default.json 2> gpurun_out/r04/g2_bitwise_default.err
( timeout 600 python tools/ab_env.py late MLP_PB_DET=0 MLP_PB_DET=1,MLP_PB_CHUNKS=16 MLP_PB_DET=1,MLP_PB_CHUNKS=24 MLP_PB_DET=1,MLP_PB_CHUNKS=8 --reps 2 ) > gpurun_out/r04/g2_ab_pbdet_late.log 2>&1
cat gpurun_out/r04/g2_tests_det.log gpurun_out/r04/g2_ab_pbdet_late.log
