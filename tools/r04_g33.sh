( timeout 600 python tools/transport_200k.py 30000 50000 4 0 --family mixed --paths factor,oracle 2>&1 | grep -v Warn | cut -c1-400 ) 2>&1 | sed "s/^/30k: /"
( timeout 1200 python tools/transport_200k.py 60000 100000 4 0 --family mixed --paths factor,oracle 2>&1 | grep -v Warn | cut -c1-400 ) 2>&1 | sed "s/^/60k: /"
