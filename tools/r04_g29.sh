ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for mode in graph eager; do
  rm -rf /tmp/prof_tl
  if [ $mode = eager ]; then export MLP_NO_GRAPH=1; else unset MLP_NO_GRAPH; fi
  MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o t -- python $ROOT/tools/window_profile.py late 64 32 > $ROOT/gpurun_out/r04_tl_$mode.log 2>&1
  echo "== $mode"
  python $ROOT/tools/experiments/timeline_dump.py /tmp/prof_tl 600 30
done
