#!/bin/bash
# RECORD of a round-5 A/B call: the svoslam_config switch it sets existed only at the commit of the experiment (git log); the library now ignores it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in 0 2; do
  echo "# svoslam_config.track_recompute=$v : rocprofv3 --pmc ... --kernel-include-regex track_persistent -- python tools/prof/track_only.py 6 cfg4 (the streaming tracker alone at 1920x1080; mean of 5 launches; quad-cycles)" >> $O/trk_recompute_sq.txt
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
    D=/tmp/sq_t; rm -rf $D; mkdir -p $D
    SVOSLAM_CONFIG=track_recompute=$v timeout 300 rocprofv3 --pmc $set --kernel-include-regex track_persistent --kernel-trace --output-format csv -d $D -o p -- python $R/tools/prof/track_only.py 6 cfg4 > /tmp/sq_t.log 2>&1 || { echo "FAILED $set"; continue; }
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    python3 -c "
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$f')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print('%-24s calls=%d mean=%.0f' % (k, len(v), sum(v) / len(v)))" >> $O/trk_recompute_sq.txt
  done
done
cat $O/trk_recompute_sq.txt
