#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
S=/tmp/ksm; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o k -- python $R/tools/mesh_bench.py --config cfg5 --reps 1 > $O/ks_mesh.log 2>&1
f=$(find $S -name "*kernel_stats.csv" | head -1); cp $f $O/mesh_cfg5_kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/mesh_cfg5_kernel_stats.csv")))[:24]:
    print("%-64s calls %4s avg %10.1f us total %9.2f ms" % (r["Name"].replace("void ","").replace("svoslam::","")[:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
