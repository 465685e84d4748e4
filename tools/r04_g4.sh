mkdir -p gpurun_out/r04
for i in 1 2 3; do
( MLP_PB_DET=1 timeout 300 python -m pytest tests/test_late_regime.py -x -q -m gpu -k "stepped" 2>&1 | grep -E "passed|failed|AssertionError|non-finite" | head -5 ) 2>&1 | sed "s/^/det1 run $i: /"
( MLP_PB_DET=0 timeout 300 python -m pytest tests/test_late_regime.py -x -q -m gpu -k "stepped" 2>&1 | grep -E "passed|failed|AssertionError|non-finite" | head -5 ) 2>&1 | sed "s/^/det0 run $i: /"
done
