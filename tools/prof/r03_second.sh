#!/bin/bash
# round-3 second GPU call: where does the suite crash; A/B of the commit's new kernels (sequential stage pass of bench.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_second"; mkdir -p "$OUT"
cd $R
( time timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -v --deselect tests/test_gpu_fullsize.py 2>&1 | grep -v "^  File" | cut -c1-220 ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( time timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -v 2>&1 | cut -c1-300 | tail -30 ) > $OUT/pytest_fullsize.log 2>&1
tail -5 $OUT/pytest_fullsize.log
line() { grep '^{"metric"' | tail -1; }
for v in "new" "SVOSLAM_STRADDLE=1" "SVOSLAM_FILL_RESUME=0" "SVOSLAM_STRADDLE=1 SVOSLAM_FILL_RESUME=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  e=""; [ "$v" != "new" ] && e="$v"
  env $e python bench.py --steps 40 --warmup 5 --allow-missing-traffic --no-cpu-baseline 2>/dev/null | line > $OUT/ab_cfg3_$tag.json
  env $e python bench.py --workload cfg4 --steps 30 --warmup 5 --allow-missing-traffic --no-cpu-baseline 2>/dev/null | line > $OUT/ab_cfg4_$tag.json
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/ab_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("stages_sequential") or {}
        print("%-60s %8.1f fps  seq: sort %.1f plan %.1f commit %.1f march %.1f us" % (os.path.basename(f), d["value"], 1e3*s.get("fuse_sort_ms",0), 1e3*s.get("fuse_plan_ms",0), 1e3*s.get("fuse_commit_ms",0), 1e3*s.get("march_ms",0)))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
