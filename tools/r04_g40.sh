ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fac
FAMILY=mixed MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fac -o t -- python $ROOT/tools/experiments/fac_profile.py 60000 100000 4 0 36000 2000 2>&1 | grep "pivots"
python $ROOT/tools/prof_summary.py /tmp/prof_fac $ROOT/gpurun_out/r04_mixed60k_late_kernel_stats.csv 14 | head -20
