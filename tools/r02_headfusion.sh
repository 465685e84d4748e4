python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r02o_gpu_tests.log
python tools/config3_profile.py > gpurun_out/r02o_config3.log 2>&1
MLP_NO_HEAD_FUSION=1 python tools/config3_profile.py 2>&1 | sed "s/^/no head fusion: /" >> gpurun_out/r02o_config3.log
python tools/window_profile.py early 2000 200 > gpurun_out/r02o_early.log 2>&1
MLP_NO_HEAD_FUSION=1 python tools/window_profile.py early 2000 200 2>&1 | sed "s/^/no head fusion: /" >> gpurun_out/r02o_early.log
python tools/config_times.py >> gpurun_out/r02o_config3.log 2>&1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 400 --warmup 100 --no-cpu-baseline > gpurun_out/r02o_bench_2rank_oversubscribed.json 2> gpurun_out/r02o_bench_2rank.err
echo "2-rank rc=$?"
cat gpurun_out/r02o_gpu_tests.log gpurun_out/r02o_config3.log gpurun_out/r02o_early.log
tail -5 gpurun_out/r02o_bench_2rank.err
python - <<PY
import json
d = json.load(open("gpurun_out/r02o_bench_2rank_oversubscribed.json"))
print("2 ranks:", d["value"], d["scaling"], d["config"]["parallelism"][:200])
w = d.get("windows", {}).get("late_sharded")
print("late sharded:", {k: w[k] for k in w if k not in ("kernels", "sampling")} if w else None)
if w and "kernels" in w: print({k: (round(v.get("avg_us", 0), 1), round(v.get("frac", 0), 3)) for k, v in w["kernels"].items()})
PY
