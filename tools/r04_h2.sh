timeout 600 python tools/ab_env.py late "MLP_VBRANCH=0" "MLP_VBRANCH=1" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
timeout 300 python tools/ab_env.py mid "MLP_VBRANCH=0" "MLP_VBRANCH=1" --reps 1 --pivots 1024 2>&1 | grep -v Warn | cut -c1-200
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/h2_tests.log 2>&1; grep -E "passed|failed" gpurun_out/h2_tests.log | tail -3
