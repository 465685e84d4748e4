mkdir -p gpurun_out/r04
( timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu -s 2>&1 | grep -v "Warn\|^\[W" | tail -30 ) 2>&1 | sed "s/^/factor: /"
( timeout 900 python tools/transport_200k.py 20000 20000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | tail -6 ) 2>&1 | sed "s/^/40k: /"
cd /tmp && export TMPDIR=/tmp
( MLP_FACTOR=1 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_fac40k -o fac40k -- python $GRAFT_REPO_ROOT/tools/transport_200k.py 20000 20000 4 0.4 --paths factor 2>&1 | tail -3 )
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r04/prof_fac40k -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; head -25 "$f" | cut -c1-200
( timeout 1500 python tools/transport_200k.py 100000 100000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | tail -10 ) 2>&1 | sed "s/^/200k: /"
