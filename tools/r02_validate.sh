# Round-2 validation on the GPU box: full GPU suite, the bench line (all windows), per-window rocprofv3 kernel stats.
TAG=${1:-r02k}
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/${TAG}_gpu_tests.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_args.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
for w in late mid early; do
  rm -rf /tmp/prof_$w
  MLP_IMPORT_TORCH=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o t -- python $GRAFT_REPO_ROOT/tools/window_profile.py $w 512 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${w}_run.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_$w $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${w}_kernel_stats.csv 200 > /dev/null
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/${TAG}_gpu_tests.log
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("early", d["value"], d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["launches"])
for w in ("mid", "late"):
    x = d["windows"][w]
    print(w, x["pivots_per_s"], x["us_per_pivot"], {k: (round(v.get("avg_us", 0), 1), round(v.get("frac", 0), 3)) for k, v in x["kernels"].items()})
print("cpu", d["cpu_baseline"]["value"])
PY
for w in late mid early; do grep pivots/s gpurun_out/${TAG}_${w}_run.log; grep -v "rocsolver\|Cijk\|rocblas" gpurun_out/${TAG}_${w}_kernel_stats.csv | head -13; done
