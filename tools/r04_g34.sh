( timeout 600 python tools/transport_200k.py 30000 50000 4 0 --family mixed --paths factor --chunk 1000 2>&1 | grep -v Warn | cut -c1-300 ) 2>&1 | sed "s/^/30k: /"
