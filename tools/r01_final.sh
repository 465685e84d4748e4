#!/bin/bash
# Round-1 measurement set: bench line, rocprofv3 kernel stats of the same command, PMC calibration,
# 2-rank (oversubscribed) run of the sharded bench path.  Outputs under gpurun_out/.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r01g}
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o b -- python $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/prof_$TAG.log 2>&1; \
  f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${TAG}_bench_kernel_stats.csv && head -14 "$f" | cut -c1-160 )
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_${TAG}_2rank.json 2> gpurun_out/bench_${TAG}_2rank.err; echo "2-rank rc=$?"; cat gpurun_out/bench_${TAG}_2rank.json | cut -c1-700
bash tools/pmc_traffic.sh 2>&1 | tail -12
