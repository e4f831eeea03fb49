#!/bin/bash
# pool_refresh_kernel under variants of its launch shape (SVOSLAM_REFRESH_BLOCKS x SVOSLAM_REFRESH_CHAINS): mean duration over the
# last 100 frames of the bench under rocprofv3 --kernel-trace, and the march-start period of the same frames
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/refresh_ab"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-2048:4 4096:2 4096:4 1024:8 2048:8 2048:2 8192:1 1024:4}; do
  b=${v%%:*}; c=${v##*:}
  S=/tmp/rf; rm -rf $S; mkdir -p $S
  SVOSLAM_REFRESH_BLOCKS=$b SVOSLAM_REFRESH_CHAINS=$c timeout 600 rocprofv3 --kernel-trace --output-format csv -d $S -o k -- python $R/bench.py --no-cpu-baseline --no-stage-pass "$@" > $OUT/bench_$b_$c.log 2>&1
  t=$(find $S -name "*kernel_trace.csv" | head -1)
  python3 - $t $b $c <<'PY' | tee -a $OUT/summary.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "svoslam" in r["Kernel_Name"]]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rf = [(r["e"] - r["s"]) / 1e3 for r in rows if "pool_refresh" in r["Kernel_Name"]][-100:]
m = [r["s"] for r in rows if "cone_trace" in r["Kernel_Name"]][-101:]
md = [(r["e"] - r["s"]) / 1e3 for r in rows if "cone_trace" in r["Kernel_Name"]][-100:]
print("blocks %5s chains %s: refresh %.1f us (min %.1f), march %.1f us, period %.1f us" % (sys.argv[2], sys.argv[3], sum(rf) / len(rf), min(rf), sum(md) / len(md), (m[-1] - m[0]) / 1e3 / (len(m) - 1)))
PY
done
