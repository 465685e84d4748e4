export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_small_basis.py -x -q -m gpu > gpurun_out/h5_sb.log 2>&1; grep -E "passed|failed|assert" gpurun_out/h5_sb.log | tail -4
timeout 600 python -m pytest tests/test_late_regime.py -x -q -m gpu -k "v_branch" > gpurun_out/h5_vb.log 2>&1; grep -E "passed|failed|^E" gpurun_out/h5_vb.log | tail -6
for w in early late; do rm -rf /tmp/prof_$w; ( cd /tmp && MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $GRAFT_REPO_ROOT/tools/window_profile.py $w 160 > /dev/null 2>&1 ); python tools/prof_summary.py /tmp/prof_$w gpurun_out/r04b_${w}_kernel_stats.csv 100 | head -24; done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/h5_tests.log 2>&1; grep -E "passed|failed" gpurun_out/h5_tests.log | tail -3
