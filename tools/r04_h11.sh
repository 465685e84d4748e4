timeout 900 python -m pytest tests/test_hyper.py tests/test_hip_parity.py tests/test_sparse_row.py tests/test_mps_tsp.py -x -q -m gpu > gpurun_out/h11_tests.log 2>&1; grep -E "passed|failed|^E|config 3" gpurun_out/h11_tests.log | tail -6
timeout 200 python tools/hyper_profile.py 2>&1 | grep -E "MLP_HYPER=1:|MLP_HYPER=0|CPU" | cut -c1-200
