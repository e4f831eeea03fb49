#!/bin/bash
# the driver's command under rocprofv3 --kernel-trace: the map stream's kernels of the last 27 frames (2 history + 5 warm-up + 20 timed), durations in us
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/mstream"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
S=/tmp/mt20; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $S -o k -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stage-pass > $OUT/bench20.log 2>&1
t=$(find $S -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY' | tee $OUT/frames20.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "svoslam" in r["Kernel_Name"]]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
m = [i for i, r in enumerate(rows) if "cone_trace" in r["Kernel_Name"]]
q = rows[m[-1]]["Queue_Id"]
mq = [r for r in rows if r["Queue_Id"] == q]
mi = [i for i, r in enumerate(mq) if "cone_trace" in r["Kernel_Name"]]
print("# map-stream kernels per frame: start of the march (ms from the first listed), then durations (us) of apply / grid update / brick rebuild / march, and the march-to-march period")
t0 = mq[mi[-27]]["s"]
prev = None
for j in mi[-27:]:
    d = {}
    k = j - 1
    while k >= 0 and "cone_trace" not in mq[k]["Kernel_Name"]:
        n = mq[k]["Kernel_Name"]
        for key in ("commit_apply", "pool_grid_update", "brick_rebuild", "fill_mip", "mip_straddle", "split_all"):
            if key in n: d[key] = (mq[k]["e"] - mq[k]["s"]) / 1e3
        k -= 1
    r = mq[j]
    print("march start %8.3f ms  apply %5.1f update %5.1f rebuild %5.1f march %6.1f  period %6.1f" % ((r["s"] - t0) / 1e6, d.get("commit_apply", 0), d.get("pool_grid_update", 0),
          d.get("brick_rebuild", 0), (r["e"] - r["s"]) / 1e3, ((r["e"] - prev) / 1e3) if prev else 0))
    prev = r["e"]
PY
grep -o '"value": [0-9.]*' $OUT/bench20.log | head -1
