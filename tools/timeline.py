"""Per-queue timeline of one training step from a rocprofv3 --kernel-trace CSV (steps are delimited by adamw_kernel).
usage: python tools/timeline.py <dir with *_kernel_trace.csv> [--full]"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step starts with the weight split (wabsmax, plane engine) -- with per-bucket AdamW (data parallel) there are several adamw
# launches per step, so adamw only delimits steps when no wabsmax is in the trace
idx = [i for i, r in enumerate(rows) if "wabsmax" in r["Kernel_Name"]]
if len(idx) >= 3:
    a, b = idx[-3], idx[-2]
    step = rows[a:b]
else:
    idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print("step wall %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
qs = collections.defaultdict(list)
for r in step:
    qs[r["Queue_Id"]].append(r)
for q, l in qs.items():
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in l)
    u, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print("queue %s: %d kernels, busy %.3f ms, first %.3f last %.3f" % (q, len(l), u / 1e6, (iv[0][0] - t0) / 1e6, (max(e for _, e in iv) - t0) / 1e6))
agg = collections.defaultdict(float)
for r in step:
    agg[(r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void segmm::", "")[:48])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (q, k), v in sorted(agg.items(), key=lambda x: -x[1])[:16]:
    print("  q%s %-50s %7.3f ms" % (q, k, v))
if "--full" in sys.argv:
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%7.3f %7.3f q%s %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Queue_Id"], r["Kernel_Name"][:60].replace("void segmm::", "")))
