for w in late mid; do
  python tools/window_profile.py $w 512 2>&1 | grep pivots/s
  MLP_SWEEP_ORDER=pos python tools/window_profile.py $w 512 2>&1 | grep pivots/s | sed "s/^/position order: /"
done
python tools/window_profile.py early 2000 200 2>&1 | grep pivots/s
MLP_SWEEP_ORDER=pos python tools/window_profile.py early 2000 200 2>&1 | grep pivots/s | sed "s/^/position order: /"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_golden_oracle.py tests/test_stage_parity.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for w in late; do rm -rf /tmp/prof_$w; MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $GRAFT_REPO_ROOT/tools/window_profile.py $w 512 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_$w $GRAFT_REPO_ROOT/gpurun_out/r02s_${w}_kernel_stats.csv 200 > /dev/null; done
cd $GRAFT_REPO_ROOT; grep -v "rocsolver\|Cijk\|rocblas" gpurun_out/r02s_late_kernel_stats.csv | head -14
