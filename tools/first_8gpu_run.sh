#!/bin/bash
# The first run on a multi-GPU node: nothing of DESIGN.md §6 has met a second device yet, so the order is
#   1. the sharded GPU tests (incl. the RCCL transport between two devices, which is skipped on a one-GPU box),
#   2. the bitwise replica gate on all ranks (tools/shard_bitwise.py: SHA-1 digests of what every rank holds, every 500 pivots),
#   3. bench.py --gpus N for N in NGPUS (default "1 2 4 8"; it starts its N ranks itself and refuses a mismatched world),
#   4. ONE table from gpurun_out/bench_detail_n<N>.json (tools/scale_table.py): value, scaling label, late sharded us per pivot,
#      pricing path us per pivot, pricing speed-up against the unsharded run of the same invocation.
# On a one-GPU test rig:  MLP_OVERSUBSCRIBE=1 NGPUS="1 2" GATE_RANKS=2 tools/first_8gpu_run.sh   (every figure is then time-slicing noise;
# the point is that every step runs and the table has its columns).
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/first_multi_gpu}
NGPUS=${NGPUS:-"1 2 4 8"}
GATE_RANKS=${GATE_RANKS:-8}
BENCH_ARGS=${BENCH_ARGS:-"--steps 20 --warmup 5 --no-full-solve --no-factor-transport"}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
rc_all=0
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  echo "== 1. sharded GPU tests" | tee "$OUT/steps.log"
  timeout 3600 python -m pytest tests/test_dist_gpu.py -x -q -m "gpu and not slow" > "$OUT/tests.log" 2>&1; rc=$?
  tail -3 "$OUT/tests.log" | tee -a "$OUT/steps.log"; [ $rc -ne 0 ] && rc_all=1
fi
if [ "${SKIP_GATE:-0}" != "1" ]; then
  echo "== 2. bitwise replica gate, $GATE_RANKS ranks, 2 000 pivots from the late basis" | tee -a "$OUT/steps.log"
  timeout 3000 python tools/shard_bitwise.py "$GATE_RANKS" 2000 500 default > "$OUT/bitwise_gate.log" 2>&1; rc=$?
  grep '^{' "$OUT/bitwise_gate.log" | tail -1 | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print("gate ok:", r["ok"], "| checkpoints:", [(c["pivots"], c["replicas_identical"]) for c in r["checkpoints"]])' 2>/dev/null | tee -a "$OUT/steps.log"
  [ $rc -ne 0 ] && { echo "gate FAILED (rc $rc): see $OUT/bitwise_gate.log" | tee -a "$OUT/steps.log"; rc_all=1; }
fi
echo "== 3. bench.py --gpus N" | tee -a "$OUT/steps.log"
for n in $NGPUS; do
  timeout 3000 python bench.py --gpus "$n" $BENCH_ARGS > "$OUT/bench_n$n.json" 2> "$OUT/bench_n$n.err"; rc=$?
  echo "   N=$n rc=$rc $(head -c 160 "$OUT/bench_n$n.json")" | tee -a "$OUT/steps.log"
  [ $rc -ne 0 ] && rc_all=1
  [ -f "gpurun_out/bench_detail_n$n.json" ] && cp "gpurun_out/bench_detail_n$n.json" "$OUT/"
done
echo "== 4. table" | tee -a "$OUT/steps.log"
python tools/scale_table.py "$OUT" $NGPUS | tee "$OUT/scale_table.md"
exit $rc_all
