#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_lowrank.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
TAG=late bash tools/late_profile.sh 4000 12 MLP_IMPORT_TORCH=1 MLP_NO_REINV=1 | head -7
grep chunk gpurun_out/late_run.log | tail -2 | cut -c1-150
TAG=late32 bash tools/late_profile.sh 4000 12 MLP_IMPORT_TORCH=1 MLP_NO_REINV=1 MLP_LOWRANK=32 | head -7
grep chunk gpurun_out/late32_run.log | tail -2 | cut -c1-150
