#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  D=/tmp/sq_r; rm -rf $D; mkdir -p $D
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex scanline_kernel --kernel-trace --output-format csv -d $D -o p -- python $R/tools/mesh_bench.py --config cfg5 --reps 1 > /tmp/sq_r.log 2>&1 || { echo "FAILED $set"; tail -3 /tmp/sq_r.log; continue; }
  f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
  python3 -c "
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$f')): acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()): print('%-42s %-24s calls=%d mean=%.0f' % (k[0], k[1], len(v), sum(v) / len(v)))" | tee -a $O/raster_sq.txt
done
