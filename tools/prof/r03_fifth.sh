#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_fifth; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
for v in "SVOSLAM_TRACK_STREAM=1" "SVOSLAM_TRACK_STREAM=1 SVOSLAM_TRACK_STREAM_WAVES=2"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v timeout 400 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sensor.py -m gpu -x -q -k "cfg4 or tracker or camera" --timeout 200 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -4 > $OUT/pytest_$tag.log; echo "$v: $(tail -2 $OUT/pytest_$tag.log)"
done
for rep in 1 2; do for v in "X=0" "SVOSLAM_TRACK_STREAM=1" "SVOSLAM_TRACK_STREAM=1 SVOSLAM_TRACK_STREAM_WAVES=2" "SVOSLAM_TRACK_STREAM=1 SVOSLAM_RUNNER_PRIO=2"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $OUT/cfg4_${tag}_$rep.json
done; done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/cfg4_*.json")):
    try:
        d = json.load(open(f)); r = {s["stage"]: s for s in d["roofline_stages"]}
        print("%-70s %8.1f fps  tracker %.3f ms march %.3f ms" % (os.path.basename(f), d["value"], r["tracker"]["kernel_ms"], r["march"]["kernel_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
SVOSLAM_TRACK_STREAM=1 python tools/prof/track_only.py 12 cfg4 2>&1 | tail -1
