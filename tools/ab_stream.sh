# A/B of the large-nucleus kernels in situ (mid / late windows of bench.py); run on the GPU box
for w in late mid; do
  python tools/window_profile.py $w 512 2>&1 | grep pivots/s >> gpurun_out/r02j_ab.log
  MLP_OLD_FOLD=1 python tools/window_profile.py $w 512 2>&1 | grep pivots/s | sed "s/^/old fold: /" >> gpurun_out/r02j_ab.log
done
python tools/window_profile.py early 2000 200 2>&1 | grep pivots/s >> gpurun_out/r02j_ab.log
cd /tmp && export TMPDIR=/tmp
for w in late mid early; do
  rm -rf /tmp/prof_$w
  MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $GRAFT_REPO_ROOT/tools/window_profile.py $w 512 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_$w $GRAFT_REPO_ROOT/gpurun_out/r02j_${w}_kernel_stats.csv 200 > /dev/null
done
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lowrank.py tests/test_stage_parity.py tests/test_hip_parity.py tests/test_golden_oracle.py -x -q -m gpu 2>&1 | tail -3
cat gpurun_out/r02j_ab.log
for f in gpurun_out/r02j_*_kernel_stats.csv; do echo $f; grep -v "rocsolver\|Cijk\|rocblas" $f | head -14; done
