mkdir -p gpurun_out/r04
( timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu -s 2>&1 | grep -v "Warn\|^\[W" | tail -40 ) 2>&1 | sed "s/^/factor: /"
( MLP_FINAL_REFRESH=1000 timeout 600 python tools/shard_full_solve.py 2 3000 3000 12 2>&1 | grep "^{" | cut -c1-700 ) 2>&1 | sed "s/^/polish-sharded: /"
