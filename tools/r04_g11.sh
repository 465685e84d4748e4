mkdir -p /root/repo/gpurun_out/r04
( timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu -s 2>&1 | grep -v "Warn\|^\[W" | tail -25 ) 2>&1 | sed "s/^/factor: /"
( timeout 900 python tools/transport_200k.py 20000 20000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | tail -4 ) 2>&1 | sed "s/^/40k: /"
( timeout 1500 python tools/transport_200k.py 100000 100000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | tail -9 ) 2>&1 | sed "s/^/200k: /"
( timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu -k "pump or rccl" 2>&1 | grep -v "Warn\|^\[W" | tail -25 ) 2>&1 | sed "s/^/pump: /"
cd /tmp && export TMPDIR=/tmp
( MLP_FACTOR=1 timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04/prof_fac40k -o fac40k -- python /root/repo/tools/transport_200k.py 20000 20000 4 0.4 --paths factor > /dev/null 2>&1 )
cd /root/repo
f=$(find gpurun_out/r04/prof_fac40k -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; head -22 "$f" | cut -c1-220
