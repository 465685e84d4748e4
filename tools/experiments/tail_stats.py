"""Per-kernel statistics of the LAST `frac` of a rocprofv3 --kernel-trace CSV by time (the window a profile script ends with)."""
import csv, glob, sys, collections
d, frac = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    if int(r["Start_Timestamp"]) < cut: continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc[r["Kernel_Name"][:60]]; a[0] += 1; a[1] += dur; a[2] = max(a[2], dur)
tot = sum(a[1] for a in acc.values())
print("window %.1f ms, kernels busy %.1f ms" % ((t1 - cut) / 1e6, tot / 1e3))
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-62s calls %6d  total %9.1f us  avg %8.2f  max %8.1f  %5.1f%%" % (k, a[0], a[1], a[1] / a[0], a[2], 100 * a[1] / tot))
