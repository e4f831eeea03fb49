#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_eighth; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_pipeline.py tests/test_gpu_sharded.py -m gpu -x -q --timeout 300 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -6 > $OUT/pytest.log; cat $OUT/pytest.log
for m in keys points; do
  SVOSLAM_FORCE_DIST=1 SVOSLAM_BAND_FUSION=$m python bench.py --steps 60 --warmup 5 --no-cpu-baseline --exchange allreduce --map-frames 0 2>$OUT/err_$m.log | line > $OUT/cfg3_forced_allreduce_$m.json
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print("%-40s %8.1f fps %.3f ms/frame" % (os.path.basename(f), d["value"], d["ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
tail -3 $OUT/err_keys.log
