#!/bin/bash
# Round 6: + the pulled F product (k_fpull_p1, k_fpull_p2) and the kernels of the early window as it now runs (k_primal_head, k_update_pivot).
# HBM traffic of the dominant kernels of the early and the late window from the PMC counters, as MI355X_MICROARCH.md
# prescribes: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; traffic = 2*FETCH_SIZE + WRITE_SIZE
# (KB -> bytes; the factor 2 and the exactness of WRITE_SIZE were calibrated in round 1, profiles/r01_pmc_calibration.md).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_r06"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in early late; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/${w}_$c"
    # counters only on the kernels of the pivot loop: with the ~41 000 launches of the blocked re-inversion at load time (mlp::k_inv_*) in the
    # collection rocprofv3 itself crashes (segmentation fault) on the late window
    MLP_IMPORT_TORCH=1 timeout 900 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_stream_w|k_fold_w|k_ftran_gather|k_fpull_p1|k_fpull_p2|k_push_stage1|k_push_combine|k_sweep_band|k_ratio_primal|k_update_pivot|k_post_fused|k_primal_head|k_row_pull" --output-format csv -d "$OUT/${w}_$c" -o p -- python $ROOT/tools/window_profile.py $w $([ $w = early ] && echo "40 5" || echo "160 32") > "$OUT/${w}_$c.log" 2>&1
  done
done
python - "$OUT" "$ROOT/gpurun_out/r06_pmc_traffic.json" <<'PY'
import csv, glob, sys, json, collections
out, dst = sys.argv[1], sys.argv[2]
doc = dict(workload=dict(rows=100000, cols=100000, nnz_per_row=100, seed=4, pivots="early: 5..45 from the slack basis (small-nucleus primal head + update kernel); late: 32..192 from tests/golden/cfg4_basis_p240000"),
           method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/window_profile.py; traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes)",
           kernels={})
def collect(w):
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == c:
                    acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            big = [x for x in v if x >= 0.5 * max(v)] or v   # the launches that do the work (the fold kernel is launched every pivot and exits at once on 31 of 32)
            res[k][c] = (sum(v) / len(v), len(v), max(v), sum(big) / len(big), len(big))
    return res
for w in ("early", "late"):
    for k, d in collect(w).items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        name = None
        if "k_sweep_band" in k: name = "sweep" if w == "early" else "sweep_late"
        elif "k_stream_w<true" in k: name = "stream_late"
        elif "k_stream_w<false" in k: name = "stream_dense_ftran_late"   # the dense-rhs FTRAN x_B = B^-1 (b - N x_N) at load time (tau side of the strip kernel)
        elif "k_fold_w2" in k: name = "fold_late"      # (the default fold kernel: U through the scalar unit)
        elif "k_fold_w<" in k: name = "fold_lds_late"
        elif "k_fused_w" in k and w == "early": name = "fused"
        elif "k_push_stage1" in k: name = "push_stage1_" + w
        elif "k_row_pull" in k: name = "row_pull_" + w
        elif "k_row_touch" in k: name = "row_touch_" + w
        elif "k_ftran_gather" in k: name = "ftran_gather_" + w
        elif "k_push_combine" in k: name = "push_combine_" + w
        elif "k_ftran_prep" in k: name = "ftran_prep_" + w
        elif "k_primal_head" in k: name = "primal_head_" + w
        elif "k_update_pivot" in k: name = "update_" + w
        elif "k_ratio_primal_fused" in k: name = "ratio_primal_" + w
        elif "k_fpull_p1" in k: name = "fpull_p1_" + w      # pull of the F product + Harris pass 1 (algorithmic: 12 nnz(nucleus columns) + 56 m)
        elif "k_fpull_p2" in k: name = "fpull_p2_" + w      # Harris pass 2 + decision | t_K | append
        elif "k_post_fused" in k: name = "post_" + w
        if name is None or (name in doc["kernels"]):
            continue
        f, wr = d["FETCH_SIZE"], d["WRITE_SIZE"]
        rec = dict(kernel=k, launches=f[1], fetch_size_kb=f[0], write_size_kb=wr[0], hbm_bytes_per_launch=(2.0 * f[0] + wr[0]) * 1024.0,
                   working_launches=f[4], hbm_bytes_per_working_launch=(2.0 * f[3] + wr[3]) * 1024.0)  # launches that exit at once (a skipped pass, a non-folding pivot) left out
        if name.startswith("fold"):  # most launches exit at once (a fold every 32nd pivot): report the launches that fold, averaged
            rec["folding_launches"] = f[4]
            rec["hbm_bytes_per_folding_launch"] = (2.0 * f[3] + wr[3]) * 1024.0
            rec["algorithmic_bytes_per_folding_launch"] = "16 k^2 (read + write of the nucleus inverse), k = 20 493 + pivots since the load"
        if "push_stage1" in name and "_det" in k:
            continue
        doc["kernels"][name] = rec
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
