export TMPDIR=/tmp
timeout 300 python tools/ab_env.py late "MLP_X=0" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
timeout 300 python tools/ab_env.py mid "MLP_X=0" --reps 2 --pivots 512 2>&1 | grep -v Warn | cut -c1-200
for w in late mid; do rm -rf /tmp/prof_$w; ( cd /tmp && MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $GRAFT_REPO_ROOT/tools/window_profile.py $w 256 > /dev/null 2>&1 ); python tools/prof_summary.py /tmp/prof_$w gpurun_out/r04c_${w}_kernel_stats.csv 200 | grep -v "k_inv\|k_fold" | head -16; done
timeout 1500 python -m pytest tests/test_late_regime.py tests/test_lowrank.py tests/test_dist_gpu.py tests/test_stage_parity.py -x -q -m gpu > gpurun_out/h8_late.log 2>&1; grep -E "passed|failed|^E" gpurun_out/h8_late.log | tail -6
