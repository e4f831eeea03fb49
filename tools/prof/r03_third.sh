#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_third; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
for rep in 1 2; do
for v in "X=0" "SVOSLAM_STRADDLE=1" "SVOSLAM_FILL_RESUME=0"; do
  tag=$(echo "$v" | tr ' =' '__')_$rep
  env $v python bench.py --steps 60 --warmup 5 --allow-missing-traffic --no-cpu-baseline 2>/dev/null | line > $OUT/ab_cfg3_$tag.json
  env $v python bench.py --workload cfg4 --steps 30 --warmup 5 --allow-missing-traffic --no-cpu-baseline 2>/dev/null | line > $OUT/ab_cfg4_$tag.json
done
done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/ab_*.json")):
    try:
        d = json.load(open(f))
        s = d.get("stages_sequential") or {}
        print("%-44s %8.1f fps  seq: sort %.1f plan %.1f commit %.1f march %.1f us" % (os.path.basename(f), d["value"], 1e3*s.get("fuse_sort_ms",0), 1e3*s.get("fuse_plan_ms",0), 1e3*s.get("fuse_commit_ms",0), 1e3*s.get("march_ms",0)))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
bash tools/prof/profile_round.sh r03 quick > $OUT/profile_round.log 2>&1
tail -60 $OUT/profile_round.log
