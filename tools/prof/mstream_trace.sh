#!/bin/bash
# one steady-state frame of the bench under rocprofv3 --kernel-trace: every kernel between two consecutive march launches, with its queue, start and duration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/mstream"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
S=/tmp/mt; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $S -o k -- python $R/bench.py --no-cpu-baseline --no-stage-pass "$@" > $OUT/bench.log 2>&1
t=$(find $S -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY' | tee $OUT/frame.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "svoslam" in r["Kernel_Name"]]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
m = [i for i, r in enumerate(rows) if "cone_trace" in r["Kernel_Name"]]
a, b = m[-12], m[-11]
t0 = rows[a]["s"]
q = rows[a]["Queue_Id"]
print("# kernels from one march launch to the next (frame ~%d of %d); times in us from the first march's start; * = same queue as the march" % (len(m) - 12, len(m)))
for r in rows[a:b + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svoslam::", "")[:44]
    print("%s q%-3s %-44s start %8.1f  dur %7.1f  end %8.1f" % ("*" if r["Queue_Id"] == q else " ", r["Queue_Id"], n, (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, (r["e"] - t0) / 1e3))
PY
