mkdir -p gpurun_out/r04
( timeout 3300 python tools/shard_full_solve.py 2 ) > gpurun_out/r04/g7_sharded_full_solve.log 2>&1
grep -v "Warn\|^\[W" gpurun_out/r04/g7_sharded_full_solve.log | tail -25
