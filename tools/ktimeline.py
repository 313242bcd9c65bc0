"""Print one steady-state period of a rocprofv3 kernel trace: kernel, duration, gap to the previous one.
usage: python tools/ktimeline.py DIR [anchor-substring] [period-index]"""
import sys, glob, os, csv
fs = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(fs[-1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "f16p"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
a, b = idx[k], idx[k + 1]
prev_end = int(rows[a - 1]["End_Timestamp"]) if a else int(rows[a]["Start_Timestamp"])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    prev_end = e
