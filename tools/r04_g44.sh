( timeout 1500 python -m pytest tests/test_factor.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5 ) 2>&1 | sed "s/^/tests: /"
FAMILY=mixed timeout 600 python tools/experiments/fac_timeline.py 60000 100000 4 37000 3 2>&1 | grep -v Warn | cut -c1-250
( timeout 600 python tools/transport_200k.py 30000 50000 4 0 --family mixed --paths factor --chunk 4000 2>&1 | grep -v Warn | cut -c1-330 ) 2>&1 | sed "s/^/30k: /" | tail -4
( timeout 900 python tools/transport_200k.py 60000 100000 4 0 --family mixed --paths factor --chunk 5000 2>&1 | grep -v Warn | cut -c1-330 ) 2>&1 | sed "s/^/60k: /"
