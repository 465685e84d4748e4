#!/bin/bash
# bumps below 48 columns: dense carrier (grid-barrier Gauss-Jordan, one launch) against the dense tail of the sparse factor (register Gauss-Jordan)
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd $ROOT
for f in 48 2; do echo "== MLP_FACTOR_SB_FROM=$f"; MLP_FACTOR_SB_FROM=$f timeout 300 python tools/experiments/factor_once.py mixed 60000 100000 4 0 10000 2>&1 | grep -v Warn | tail -4 | cut -c1-220; done
for f in 48 2; do echo "== MLP_FACTOR_SB_FROM=$f (config 3, factor forced)"; MLP_FACTOR=1 MLP_FACTOR_SB_FROM=$f timeout 300 python tools/experiments/factor_once.py mixed 6000 10000 4 0 2000 2>&1 | grep -v Warn | tail -2 | cut -c1-220; done
