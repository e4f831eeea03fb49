#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_fourth; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
timeout 600 python -m pytest tests/test_gpu_io.py tests/test_gpu_configs.py tests/test_gpu_sensor.py tests/test_gpu_sharded.py -m gpu -x -q --timeout 200 2>&1 | tail -8 > $OUT/pytest_default.log; tail -3 $OUT/pytest_default.log
for h in 0 2; do
  SVOSLAM_TRACK_HYBRID=$h timeout 400 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "cfg4" --timeout 200 2>&1 | tail -4 > $OUT/pytest_hybrid$h.log; echo "hybrid $h: $(tail -1 $OUT/pytest_hybrid$h.log)"
done
for rep in 1 2; do for h in 0 1 2; do
  SVOSLAM_TRACK_HYBRID=$h python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $OUT/cfg4_hybrid${h}_$rep.json
done; done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/cfg4_*.json")):
    try:
        d = json.load(open(f)); r = {s["stage"]: s for s in d["roofline_stages"]}
        print("%-28s %8.1f fps  tracker %.3f ms march %.3f ms" % (os.path.basename(f), d["value"], r["tracker"]["kernel_ms"], r["march"]["kernel_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
