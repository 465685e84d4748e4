"""Print the start / end (us, relative) of the kernels of a few consecutive pivots from a rocprofv3 --kernel-trace CSV
(diagnostic: do the side-stream kernels of the v branch overlap the ratio test and the BTRAN?)."""
import csv, glob, sys
d = sys.argv[1]
n_skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n_show = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "mlp::" in r["Kernel_Name"]]
sel = rows[n_skip:n_skip + n_show]
t0 = int(sel[0]["Start_Timestamp"]) if sel else 0
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f q=%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
# mean pivot period (between consecutive k_update_pivot ends) and the neighbourhood of a folding pivot
ends = [int(r["End_Timestamp"]) for r in rows if "k_update_pivot" in r["Kernel_Name"]]
per = [(b - a) / 1e3 for a, b in zip(ends, ends[1:])]
per_s = sorted(per)
print("pivots %d  mean period %.1f us  median %.1f  p90 %.1f  max %.1f" % (len(per), sum(per) / len(per), per_s[len(per) // 2], per_s[int(len(per) * 0.9)], per_s[-1]))
idx = [i for i, r in enumerate(rows) if "k_fold_w2" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 500000]
if len(idx) > 2:
    i0 = idx[2]
    t0 = int(rows[i0 - 8]["Start_Timestamp"])
    for r in rows[i0 - 8:i0 + 10]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%9.1f %9.1f %7.1f q=%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
