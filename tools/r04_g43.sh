FAMILY=mixed timeout 600 python tools/experiments/fac_timeline.py 60000 100000 4 37000 4 2>&1 | grep -v Warn | cut -c1-250
