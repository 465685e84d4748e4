#!/bin/bash
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -v "^W2026" | tail -4
timeout 300 python tools/experiments/factor_once.py mixed 100000 160000 4 0 10000 2>&1 | grep -v Warn | tail -4 | cut -c1-220
ORACLE=0 timeout 300 python tools/experiments/staircase_probe.py 2>&1 | grep -v "^W2026" | cut -c1-300 | tail -3
timeout 600 python tools/experiments/factor_timeline.py 100000 160000 62000 2>&1 | grep -v "^W2026" | cut -c1-400 | tail -3
