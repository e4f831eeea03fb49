"""prints the rocprofv3 kernel_stats.csv found under a directory: short kernel name, calls, average us, total us per frame"""
import csv, glob, os, sys
d, frames = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "").replace("svoslam::", "")
    if "at::native" in n or "rocclr" in n:
        continue
    print("%-46s calls %5s  avg %8.2f us  per frame %8.2f us" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / frames))
