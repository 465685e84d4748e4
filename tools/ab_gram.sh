#!/bin/bash
# A/B of the Gram mode against the streaming pass on the committed mid-solve bases of config 4,
# then the per-kernel picture of the late window with the mode on.
for w in late mid; do
  for g in 1 0; do
    MLP_GRAM=$g python tools/window_profile.py $w 512 64 2>&1 | grep pivots/s | sed "s/^/gram $g: /"
  done
done
MLP_GRAM=1 python tools/window_profile.py early 2000 200 2>&1 | grep pivots/s | sed "s/^/gram 1: /"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gram
MLP_IMPORT_TORCH=1 MLP_GRAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gram -o t -- python $GRAFT_REPO_ROOT/tools/window_profile.py late 512 64 > $GRAFT_REPO_ROOT/gpurun_out/gram_late_run.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_gram $GRAFT_REPO_ROOT/gpurun_out/gram_late_kernel_stats.csv 40 > /dev/null
head -30 $GRAFT_REPO_ROOT/gpurun_out/gram_late_kernel_stats.csv | cut -c1-150
