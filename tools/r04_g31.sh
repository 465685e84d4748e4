ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for vb in 1 0; do
  rm -rf /tmp/prof_tl
  MLP_VBRANCH=$vb MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o t -- python $ROOT/tools/window_profile.py late 256 32 > $ROOT/gpurun_out/r04_tl_vb$vb.log 2>&1
  echo "== vbranch $vb"
  python $ROOT/tools/experiments/timeline_dump.py /tmp/prof_tl 600 0 2>&1 | tail -20
done
