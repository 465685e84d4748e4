( timeout 900 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -v "Warn\|^\[W" | tail -4 ) 2>&1 | sed "s/^/factor: /"
( timeout 1500 python tools/transport_200k.py 100000 100000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | tail -8 ) 2>&1 | sed "s/^/200k skip: /"
( MLP_FACTOR_SKIP=0 timeout 1500 python tools/transport_200k.py 100000 100000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | tail -8 | head -1 ) 2>&1 | sed "s/^/200k noskip: /"
( timeout 900 python tools/transport_200k.py 20000 20000 4 0.4 --paths factor 2>&1 | grep -v "Warn\|^\[W" | grep "^factor" ) 2>&1 | sed "s/^/40k: /"
