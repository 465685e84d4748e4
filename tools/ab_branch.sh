for w in late mid; do
  python tools/window_profile.py $w 512 2>&1 | grep pivots/s
  MLP_BRANCH=0 python tools/window_profile.py $w 512 2>&1 | grep pivots/s | sed "s/^/no branch: /"
done
timeout 900 python -m pytest tests/test_lowrank.py tests/test_stage_parity.py tests/test_hip_parity.py tests/test_dist_gpu.py tests/test_basis.py -x -q -m gpu 2>&1 | tail -3
