timeout 400 python -m pytest tests/test_dist_gpu.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 400 --warmup 100 --no-cpu-baseline --no-windows > gpurun_out/r02p_bench_2rank_nowin.json 2> gpurun_out/r02p_bench_2rank_nowin.err
echo "2-rank (no windows) rc=$?"; cut -c1-600 gpurun_out/r02p_bench_2rank_nowin.json
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 400 --warmup 100 --no-cpu-baseline > gpurun_out/r02p_bench_2rank_oversubscribed.json 2> gpurun_out/r02p_bench_2rank.err
echo "2-rank rc=$?"
tail -3 gpurun_out/r02p_bench_2rank.err | cut -c1-300
python - <<PY
import json
d = json.load(open("gpurun_out/r02p_bench_2rank_oversubscribed.json"))
print("2 ranks:", d["value"], d["scaling"], d["config"]["parallelism"][:260])
w = d.get("windows", {}).get("late_sharded")
print("late sharded:", {k: w[k] for k in w if k not in ("kernels", "sampling")} if w else None)
if w and "kernels" in w: print({k: (round(v.get("avg_us", 0), 1), round(v.get("frac", 0), 3)) for k, v in w["kernels"].items()})
PY
python tools/window_profile.py early 2000 200
python tools/window_profile.py late 256
