#!/bin/bash
# rocprofv3 kernel statistics of the default bench line (or "$@"), top kernels printed and kept under gpurun_out/kstats
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/kstats"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
S=/tmp/ks; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o k -- python $R/bench.py --no-cpu-baseline --allow-missing-traffic --no-stage-pass "$@" > $OUT/bench.log 2>&1
f=$(find $S -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-70s calls %6s avg %9.1f us  total %9.1f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -1 $OUT/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
