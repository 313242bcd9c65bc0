import csv,glob,re,sys
import os
f=sorted(glob.glob(sys.argv[1]+"/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)): print(re.search(r"(nw_[a-z_0-9]+)",r["Name"]).group(1), "%.2f us"%(float(r["AverageNs"])/1e3), r["Calls"])
