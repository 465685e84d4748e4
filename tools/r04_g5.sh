mkdir -p gpurun_out/r04
( MLP_POOL_POISON=1 MLP_PB_DET=0 timeout 300 python -m pytest tests/test_late_regime.py -x -q -m gpu -k "stepped" 2>&1 | grep -E "passed|failed|Error|non-finite" | head -5 ) 2>&1 | sed "s/^/poison det0: /"
( MLP_POOL_POISON=1 MLP_PB_DET=1 timeout 300 python -m pytest tests/test_late_regime.py -x -q -m gpu -k "stepped" 2>&1 | grep -E "passed|failed|Error|non-finite" | head -5 ) 2>&1 | sed "s/^/poison det1: /"
( MLP_POOL_POISON=1 MLP_PB_DET=1 timeout 300 python tools/experiments/nan_probe.py 2>&1 | grep -E "non-finite: [1-9]|Error" | head ) 2>&1 | sed "s/^/poison probe det1: /"
( MLP_POOL_POISON=1 MLP_PB_DET=0 timeout 300 python tools/experiments/nan_probe.py 2>&1 | grep -E "non-finite: [1-9]|Error" | head ) 2>&1 | sed "s/^/poison probe det0: /"
( MLP_POOL_POISON=1 timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_dist_gpu.py -k "not config4 and not late and not cfg4" 2>&1 | tail -15 ) 2>&1 | sed "s/^/poison suite: /"
