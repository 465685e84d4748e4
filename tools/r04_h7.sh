timeout 600 python -m pytest tests/test_small_basis.py -x -q -m gpu > gpurun_out/h7_sb.log 2>&1; grep -E "passed|failed|assert" gpurun_out/h7_sb.log | tail -4
for hv in 262144 524288 1048576 2097152 4194304; do echo "MLP_HYPER_HEAVY=$hv"; MLP_HYPER_HEAVY=$hv timeout 120 python tools/hyper_profile.py 2>&1 | grep -E "MLP_HYPER=1:" | cut -c1-330; done
timeout 300 python bench.py --no-full-solve --no-windows --no-factor-transport --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
timeout 1500 python -m pytest tests/test_late_regime.py tests/test_lowrank.py tests/test_dist_gpu.py tests/test_basis.py tests/test_reinvert_parity.py -x -q -m gpu > gpurun_out/h7_late.log 2>&1; grep -E "passed|failed|^E" gpurun_out/h7_late.log | tail -6
timeout 300 python tools/ab_env.py late "MLP_X=0" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
timeout 300 python tools/ab_env.py mid "MLP_X=0" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
