ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fac
FAMILY=mixed MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fac -o t -- python $ROOT/tools/experiments/fac_profile.py 60000 100000 4 0 36000 2000 2>&1 | grep "pivots"
python $ROOT/tools/experiments/tail_stats.py /tmp/prof_fac 0.2
