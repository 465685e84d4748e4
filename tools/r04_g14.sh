cd /tmp && export TMPDIR=/tmp
( MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fac200k -o t -- python /root/repo/tools/experiments/fac_profile.py 100000 100000 4 0.4 110000 8000 2>&1 | grep -v "Warn" | tail -4 | cut -c1-200 )
cd /root/repo
mkdir -p gpurun_out/r04
python tools/prof_summary.py /tmp/prof_fac200k gpurun_out/r04/fac200k_late_kernel_stats.csv 16 | cut -c1-200
