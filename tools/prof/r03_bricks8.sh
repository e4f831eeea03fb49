#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_bricks8"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for b in 1 0; do
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    S=/tmp/pm_$b; rm -rf $S; mkdir -p $S
    DIAG_CFG4=1 SVOSLAM_MARCH_BRICKS=$b timeout 300 rocprofv3 --pmc $set --kernel-include-regex "cone_trace" --kernel-trace --output-format csv -d $S -o p -- python $R/tools/prof/render_only.py 45 > $OUT/pmc_run_$b.log 2>&1
    f=$(find $S -name "*counter_collection.csv" | sort | tail -1)
    python - "$f" "$b" >> $OUT/pmc_cfg4.txt <<'PY'
import csv, sys, collections
f, b = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("bricks=%s %s %s calls=%d mean=%.1f" % (b, k, c, len(v), sum(v) / len(v)))
PY
  done
done
cat $OUT/pmc_cfg4.txt
