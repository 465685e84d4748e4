export TMPDIR=/tmp
timeout 400 python tools/ab_env.py late "MLP_TK_RIDE=0" "MLP_TK_RIDE=1" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
timeout 300 python tools/ab_env.py mid "MLP_TK_RIDE=0" "MLP_TK_RIDE=1" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
timeout 1500 python -m pytest tests/test_late_regime.py tests/test_lowrank.py -x -q -m gpu > gpurun_out/h9_late.log 2>&1; grep -E "passed|failed|^E" gpurun_out/h9_late.log | tail -6
