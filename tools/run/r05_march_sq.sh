#!/bin/bash
# SQ counters of cone_trace_brick_kernel alone on the 300-frame cfg3 map (as profiles/r04_march_sq_counters_cfg3.txt), with the round's
# residency settings (two workgroups per CU, tiles costliest-first, paired strips).  Counter passes only (no sys / hip trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
OUT=$O/r05_march_sq_counters_cfg3.txt
echo "# rocprofv3 --pmc <4 SQ counters per pass> --kernel-include-regex cone_trace_brick --kernel-trace -- python tools/prof/render_only.py 300: cone_trace_brick_kernel<512, true, 0> alone on the 300-frame cfg3 map; mean per launch" > $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD"; do
  D=/tmp/sq_m; rm -rf $D; mkdir -p $D
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex cone_trace_brick --kernel-trace --output-format csv -d $D -o p -- python $R/tools/prof/render_only.py 300 > /tmp/sq_m.log 2>&1 || { echo "FAILED $set" >> $OUT; continue; }
  f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
  python3 -c "
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$f')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print('%-24s calls=%d mean=%.0f' % (k, len(v), sum(v) / len(v)))" >> $OUT
done
cat $OUT
