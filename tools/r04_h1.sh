( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) 2>&1 | sed "s/^/tests: /"
( timeout 900 python bench.py --no-full-solve --steps 20 --warmup 5 2>gpurun_out/h1_bench.err | tail -1 > gpurun_out/h1_bench.json ); tail -c 3000 gpurun_out/h1_bench.json
