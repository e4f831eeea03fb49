#!/bin/bash
# SQ counters of the brick march alone on the 300-frame cfg3 map (tools/prof/render_only.py 300), for the plain kernel and the march in bursts:
#   bash tools/prof/march_sq_counters.sh gpurun_out/r06/r06_march_sq_counters_cfg3.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=${1:-$R/gpurun_out/march_sq_counters.txt}
case $OUT in /*) ;; *) OUT=$PWD/$OUT;; esac
mkdir -p "$(dirname "$OUT")"
cd /tmp && export TMPDIR=/tmp
: > $OUT
for a in -1 90 0; do
  echo "# SVOSLAM_CONFIG=march_ahead=$a: $([ $a = -1 ] && echo 'cone_trace_brick_kernel<.., 0> (one sample per iteration)' || echo "cone_trace_brick_kernel<.., 3>: bursts of three samples past step $a")" >> $OUT
  SVOSLAM_CONFIG=march_ahead=$a python $R/tools/prof/render_only.py 300 2>&1 | grep "mode 0" >> $OUT
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS"; do
    D=/tmp/pmc_m_$$; rm -rf $D; mkdir -p $D
    SVOSLAM_CONFIG=march_ahead=$a timeout 600 rocprofv3 --pmc $set --kernel-include-regex cone_trace_brick --kernel-trace --output-format csv -d $D -o p -- python $R/tools/prof/render_only.py 300 > /tmp/pm.log 2>&1 || { echo "FAILED $set: $(tail -2 /tmp/pm.log)" >> $OUT; continue; }
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-24s calls=%d mean=%.0f" % (k, len(v), sum(v) / len(v)))
PY
  done
done
cat $OUT
