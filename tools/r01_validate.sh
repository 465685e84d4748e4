#!/bin/bash
# Round-1 validation chain on the GPU box: delayed-update tests, full GPU suite, late-regime
# comparison (in-place vs delayed update), TSP driver timing.  Logs under gpurun_out/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lowrank.py -x -q -m gpu > gpurun_out/v_lowrank.log 2>&1; echo "lowrank rc=$?"
tail -3 gpurun_out/v_lowrank.log
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/v_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -3 gpurun_out/v_gpu.log
timeout 300 python tools/gpu_perf.py 100000 100000 100 250 10 prof > gpurun_out/v_early.log 2>&1; grep chunk gpurun_out/v_early.log | cut -c1-250
MLP_LOWRANK=0 timeout 600 python tools/gpu_perf.py 100000 100000 100 4000 12 prof > gpurun_out/v_late_inplace.log 2>&1
timeout 600 python tools/gpu_perf.py 100000 100000 100 4000 12 prof > gpurun_out/v_late_lowrank.log 2>&1
grep chunk gpurun_out/v_late_inplace.log | tail -4
grep chunk gpurun_out/v_late_lowrank.log | tail -4
timeout 900 python examples/tsp.py tests/golden/bn130.tsp --backend hip > gpurun_out/v_tsp_hip.log 2>&1; tail -1 gpurun_out/v_tsp_hip.log
timeout 900 python examples/tsp.py tests/golden/bn130.tsp --backend oracle > gpurun_out/v_tsp_oracle.log 2>&1; tail -1 gpurun_out/v_tsp_oracle.log
