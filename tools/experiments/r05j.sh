#!/bin/bash
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -v "^W2026" | tail -3
for f in 1 0; do echo "== MLP_FACTOR_MULTI=$f"; MLP_FACTOR_MULTI=$f timeout 300 python tools/experiments/factor_once.py transport 100000 100000 4 0 50000 2>&1 | grep -v Warn | tail -2 | cut -c1-200; done
for f in 1 0; do echo "== MLP_FACTOR_MULTI=$f"; MLP_FACTOR_MULTI=$f timeout 300 python tools/experiments/factor_once.py mixed 100000 160000 4 0 20000 2>&1 | grep -v Warn | tail -2 | cut -c1-200; done
