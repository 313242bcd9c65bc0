import sys, glob, os, csv
fs = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
if not fs:
    sys.exit("no kernel_stats.csv under " + sys.argv[1])
for r in list(csv.DictReader(open(fs[-1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-70s calls %5s avg %10.2f us  %6s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
