#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for cfg in "16 1" "32 1" "64 1" "16 2" "32 4" "64 8"; do
  set -- $cfg
  v=$(MLP_BATCH=$1 MLP_GRAPH_ITERS=$2 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f piv/s %.1f us' % (d['value'], d['ms_per_step']*1e3))")
  echo "batch=$1 graph_iters=$2: $v"
done
