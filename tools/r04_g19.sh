mkdir -p gpurun_out/r04
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "Warn\|^\[W" | tail -12 ) 2>&1 | sed "s/^/suite: /"
( timeout 600 python bench.py --steps 20 --warmup 5 --no-full-solve --no-factor-transport 2>/dev/null | cut -c1-1500 ) 2>&1 | sed "s/^/bench-quick: /"
