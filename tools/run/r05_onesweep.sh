#!/bin/bash
# RECORD of a round-5 A/B call: the svoslam_config switch it sets existed only at the commit of the experiment (git log); the library now ignores it
# one-sweep sort: parity tests of the paths that sort, then A/B of the bench lines (sequential stage pass included)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_pipeline.py tests/test_gpu_sharded.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for v in 1 0; do
  SVOSLAM_CONFIG=sort_onesweep=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>$O/cfg3_$v.err | grep '^{"metric"' | tail -1 > $O/cfg3_os$v.json
  SVOSLAM_CONFIG=sort_onesweep=$v timeout 300 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --lean 2>$O/cfg4_$v.err | grep '^{"metric"' | tail -1 > $O/cfg4_os$v.json
done
python - <<PY
import json
for w in ("cfg3","cfg4"):
    for v in (1,0):
        try:
            d=json.load(open("$O/%s_os%d.json"%(w,v)))
            print(w, "onesweep", v, "value %.1f" % d["value"], [round(x) for x in d["runs"]], "seq:", {k:round(x,4) for k,x in d.get("stages_sequential",{}).items() if k!="note"})
        except Exception as e:
            print(w, v, "FAILED", e)
PY
