#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 600 python tools/gpu_perf.py 100000 100000 100 4000 12 prof > gpurun_out/x_$tag.log 2>&1; echo "== $tag"; grep chunk gpurun_out/x_$tag.log | tail -2 | cut -c1-200; }
run lr16_pad16 MLP_LDPAD=16
TAG=late bash tools/late_profile.sh 4000 12 MLP_LDPAD=16 MLP_IMPORT_TORCH=1 MLP_NO_REINV=1 | head -16
