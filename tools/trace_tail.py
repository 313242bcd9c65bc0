"""Per-kernel totals of the part of a rocprofv3 kernel trace that follows its longest idle gap (the measured forwards
of tools/dn_prof.py).  usage: python tools/trace_tail.py DIR REPS [top]"""
import collections, csv, glob, os, re, sys
fs = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(fs[-1])), key=lambda r: int(r["Start_Timestamp"]))
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), i) for i in range(len(rows) - 1)]
big = [i for g, i in gaps if g > 100e6]            # the 200 ms pause (any earlier long gaps are compilations / searches)
cut = (big[-1] if big else max(gaps)[1]) + 1
tail = rows[cut:]
reps = int(sys.argv[2])
acc = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    m = re.search(r"(nw_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
    k = m.group(1) if m else re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
    acc[k][0] += 1
    acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3
print(f"{len(tail)} launches after the gap; kernel time {tot / reps / 1e3:.3f} ms per forward, span {span / reps / 1e3:.3f} ms per forward")
for k, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{us / reps:9.1f} us  {n / reps:6.1f} calls  {us / n:8.2f} us/call  {k}")
