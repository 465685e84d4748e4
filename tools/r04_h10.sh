export TMPDIR=/tmp
for hb in 2 3 4 5 6 8; do echo "MLP_HYPER_BACKOFF=$hb"; MLP_HYPER_BACKOFF=$hb timeout 120 python tools/hyper_profile.py 2>&1 | grep -E "MLP_HYPER=1:|CPU" | cut -c1-330; done
rm -rf /tmp/prof_c3; ( cd /tmp && MLP_HYPER=0 MLP_IMPORT_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o t -- python $GRAFT_REPO_ROOT/tools/cfg3_profile.py > /dev/null 2>&1 ); python tools/prof_summary.py /tmp/prof_c3 gpurun_out/r04_cfg3_multikernel_kernel_stats.csv 1000 | head -24
