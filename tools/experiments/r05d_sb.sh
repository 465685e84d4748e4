#!/bin/bash
# sparse bump factor: tests, then the config-3 family at 60 000 rows on the factor
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu -s 2>&1 | grep -v "^W2026" | tail -40 > gpurun_out/r05d_tests.log; tail -25 gpurun_out/r05d_tests.log
timeout 300 python tools/experiments/factor_once.py mixed 60000 100000 4 0 10000 2>&1 | grep -v Warn | tail -8 | cut -c1-300
