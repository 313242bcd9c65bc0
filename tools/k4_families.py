"""Per-step kernel families of a K4 trace (tools/prof_train.sh): python tools/k4_families.py DIR [steps=9] [top=40]"""
import csv, glob, collections, os, re, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 9; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True), key=lambda p: -os.path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
fam = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name']
    if 'copyBuffer' in n:                     # model setup (.to(device)), not the step
        continue
    m = re.search(r"(nw_\w+)", n)
    k = m.group(1) if m else re.sub(r"\s+", " ", n)[:110] + " g=%d" % (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']))
    fam[k][0] += 1; fam[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in fam.values())
print("kernel time per step %.2f ms, launches per step %.0f" % (tot / steps / 1e3, sum(v[0] for v in fam.values()) / steps))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"n/step={v[0]/steps:6.1f} us/step={v[1]/steps:8.1f} avg={v[1]/v[0]:6.1f} {k}")
