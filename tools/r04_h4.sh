timeout 900 python -m pytest tests/test_small_basis.py -x -q -m gpu -s > gpurun_out/h4_sb.log 2>&1; grep -E "passed|failed|through|Error|assert" gpurun_out/h4_sb.log | tail -12
timeout 900 python -m pytest tests/test_reinvert_parity.py -x -q -m gpu > gpurun_out/h4_ri.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/h4_ri.log | tail -8
timeout 600 python tools/reinvert_timing.py mid 2 2>&1 | grep -v Warn | tail -3
timeout 600 python tools/reinvert_timing.py late 2 2>&1 | grep -v Warn | tail -3
timeout 1500 python -m pytest tests/test_late_regime.py tests/test_lowrank.py tests/test_basis.py -x -q -m gpu -s > gpurun_out/h4_late.log 2>&1; grep -E "passed|failed|Error|assert|pivots, objective" gpurun_out/h4_late.log | tail -8
