mkdir -p gpurun_out/r04
( timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 20 --warmup 5 2>gpurun_out/r04/bench_n2.err | tail -1 | cut -c1-3000 ) 2>&1 | sed "s/^/bench-n2: /"
tail -5 gpurun_out/r04/bench_n2.err | cut -c1-300
