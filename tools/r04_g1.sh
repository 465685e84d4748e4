set -x
mkdir -p gpurun_out/r04
( MLP_PB_DET=1 timeout 900 python -m pytest tests/test_lowrank.py tests/test_late_regime.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r04/g1_tests_det.log 2>&1
( timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r04/g1_tests_dist.log 2>&1
( timeout 900 python tools/shard_bitwise.py 2 8000 1000 default ) > gpurun_out/r04/g1_bitwise_default.json 2> gpurun_out/r04/g1_bitwise_default.err
( timeout 900 python tools/shard_bitwise.py 2 8000 1000 replicated ) > gpurun_out/r04/g1_bitwise_replicated.json 2> gpurun_out/r04/g1_bitwise_replicated.err
( timeout 600 python tools/ab_env.py late MLP_PB_DET=0 MLP_PB_DET=1 ) > gpurun_out/r04/g1_ab_pbdet_late.log 2>&1
( timeout 600 python tools/ab_env.py mid MLP_PB_DET=0 MLP_PB_DET=1 ) > gpurun_out/r04/g1_ab_pbdet_mid.log 2>&1
tail -3 gpurun_out/r04/g1_tests_det.log gpurun_out/r04/g1_tests_dist.log; cat gpurun_out/r04/g1_ab_*.log; head -c 1500 gpurun_out/r04/g1_bitwise_default.json; tail -5 gpurun_out/r04/g1_bitwise_default.err; head -c 1500 gpurun_out/r04/g1_bitwise_replicated.json; tail -5 gpurun_out/r04/g1_bitwise_replicated.err
