timeout 600 python -m pytest tests/test_small_basis.py -x -q -m gpu > gpurun_out/h6_sb.log 2>&1; grep -E "passed|failed|assert" gpurun_out/h6_sb.log | tail -4
timeout 600 python tools/small_basis_curve.py 2>&1 | grep -v Warn | cut -c1-300
timeout 1200 python -m pytest tests/test_dist_gpu.py tests/test_stress.py tests/test_sweep_order.py -x -q -m gpu > gpurun_out/h6_dist.log 2>&1; grep -E "passed|failed|^E" gpurun_out/h6_dist.log | tail -6
mkdir -p gpurun_out/r04
( timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 20 --warmup 5 --no-full-solve --no-factor-transport 2>gpurun_out/r04/bench_n2.err | tail -1 > gpurun_out/h6_bench_n2.json ); python -c "
import json; d=json.load(open('gpurun_out/h6_bench_n2.json')); print({k:d[k] for k in d if k in ('value','ms_per_step','value_vs_1gpu','pricing_speedup_vs_1gpu')}); print(d['config']); print(d.get('roofline',{}).get('ftran'))"
timeout 300 python tools/hyper_profile.py 2>&1 | grep -v Warn | cut -c1-600
