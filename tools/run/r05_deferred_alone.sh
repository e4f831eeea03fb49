#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
python tools/prof/deferred_commit_alone.py 2>&1 | grep -v amdgpu.ids | tee $O/deferred_alone.txt
cd /tmp; export TMPDIR=/tmp
S=/tmp/ksd; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o k -- python $R/tools/prof/deferred_commit_alone.py 100 10 > /tmp/ksd.log 2>&1
f=$(find $S -name "*kernel_stats.csv" | head -1)
python - <<PY | tee -a $O/deferred_alone.txt
import csv
for r in csv.DictReader(open("$f")):
    n=r["Name"].replace("void ","").replace("svoslam::","")
    if any(k in n for k in ("fill_mip","mip_straddle","split_all","commit_apply","plan_")): print("%-60s calls %5s avg %8.1f us min %8.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
