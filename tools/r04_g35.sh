for c in 20000 4000; do ( timeout 600 python tools/experiments/chunk_probe.py 30000 50000 $c 2>&1 | grep -v Warn | cut -c1-1500 ) 2>&1 | sed "s/^/c$c: /"; done
