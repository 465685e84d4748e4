timeout 900 python -m pytest tests/test_small_basis.py -x -q -m gpu -s > gpurun_out/h3_sb.log 2>&1; grep -E "passed|failed|through|Error|assert" gpurun_out/h3_sb.log | tail -12
for i in 1 2; do
for sb in 1 0; do
echo "MLP_SMALL_BASIS=$sb"; MLP_SMALL_BASIS=$sb timeout 300 python bench.py --no-full-solve --no-windows --no-factor-transport --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('timed_window'))"
done; done
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_sparse_row.py tests/test_stage_parity.py tests/test_hyper.py tests/test_mps_tsp.py -x -q -m gpu > gpurun_out/h3_tests.log 2>&1; grep -E "passed|failed" gpurun_out/h3_tests.log | tail -3
