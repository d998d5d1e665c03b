"""Per-view timeline from a rocprofv3 --kernel-trace CSV: when does each kernel of one steady-state view start and end?
usage: python tools/timeline.py <dir with *kernel_trace.csv> [skip_views]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gwbp::", "").replace("(anonymous namespace)::", "")[-44:], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
proj = [i for i, r in enumerate(rows) if "k_project" in r[2]]
k = proj[len(proj) * 2 // 3]
t0 = rows[k][0]
k1 = proj[proj.index(k) + 2] if proj.index(k) + 2 < len(proj) else len(rows)
print("two views starting at a k_project, times in us relative to it")
for s, e, n, q in rows[k:k1]:
    print("%9.1f %9.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
