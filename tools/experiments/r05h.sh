#!/bin/bash
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
timeout 900 python -m pytest tests/test_factor.py tests/test_hyper.py tests/test_abi.py -x -q -m gpu 2>&1 | grep -v "^W2026" | tail -3
for f in 1 0; do echo "== MLP_FACTOR_FUSE=$f"; MLP_FACTOR_FUSE=$f timeout 300 python tools/experiments/factor_once.py transport 100000 100000 4 0 20000 2>&1 | grep -v Warn | tail -3 | cut -c1-200; done
