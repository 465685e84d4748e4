mkdir -p gpurun_out/r04
( timeout 1500 python -m pytest tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -8 ) 2>&1 | sed "s/^/dist: /"
( MLP_PB_DET=1 timeout 600 python -m pytest tests/test_late_regime.py tests/test_lowrank.py -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | sed "s/^/det1 late+lowrank: /"
( timeout 600 python -m pytest tests/test_late_regime.py tests/test_hyper.py tests/test_sparse_row.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | sed "s/^/default late+hyper+str+abi: /"
