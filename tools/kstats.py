"""Median duration per kernel (and grid) from a rocprofv3 --kernel-trace --output-format csv directory.
python tools/kstats.py DIR [substring ...]   (only kernels whose name contains one of the substrings)"""
import csv, glob, re, sys, collections
d = sys.argv[1]
subs = sys.argv[2:]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if subs and not any(s in n for s in subs):
        continue
    m = re.search(r"(nw_\w+)(<[^>]*>)?", n)
    key = (m.group(0) if m else re.sub(r"\(.*", "", n)[:60])
    key += f" wg={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)},{r['Grid_Size_Y']},{r['Grid_Size_Z']}"
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:78s} n={len(v):5d} med={v[len(v) // 2]:8.2f} min={v[0]:8.2f} us")
