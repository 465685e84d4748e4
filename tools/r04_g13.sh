( timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "pump or rccl" 2>&1 | grep -v "Warn\|^\[W" | tail -15 ) 2>&1 | sed "s/^/pump: /"
( timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -v "Warn\|^\[W" | tail -5 ) 2>&1 | sed "s/^/factor: /"
cd /tmp && export TMPDIR=/tmp
( MLP_FACTOR=1 timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04/prof_fac40k -o fac40k -- python /root/repo/tools/transport_200k.py 20000 20000 4 0.4 --paths factor 2>&1 | tail -3 )
cd /root/repo
ls -R gpurun_out/r04/prof_fac40k | head; ls /tmp | head -20
f=$(find gpurun_out/r04/prof_fac40k /tmp -name "*kernel_stats.csv" 2>/dev/null | head -1); echo "stats file: $f"; head -22 "$f" | cut -c1-220
