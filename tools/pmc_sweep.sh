#!/bin/bash
# PMC passes for the dominant kernels (run on the GPU box from the repo root; writes gpurun_out/pmc_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/tools/gpu_perf.py 100000 100000 100 300 2 > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "== $pass"
  python - <<PY
import csv, glob, collections
fs = glob.glob("$R/gpurun_out/pmc_$tag/*counter_collection.csv")
if not fs: print("no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(fs[0])):
    k = row["Kernel_Name"].split("(")[0][-40:]
    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    if "sweep" in k or "fused_w" in k or "ratio_primal_p1" in k:
        print(k, {c: (sum(v)/len(v), len(v)) for c, v in acc[k].items()})
PY
done
