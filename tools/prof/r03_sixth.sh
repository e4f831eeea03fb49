#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_sixth; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
for v in "X=0" "SVOSLAM_TRACK_STREAM=1" "SVOSLAM_TRACK_HYBRID=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $OUT/cfg4_${tag}.json
  env $v python bench.py --workload cfg4 --steps 30 --warmup 5 --no-cpu-baseline --no-overlap 2>/dev/null | line > $OUT/cfg4_seq_${tag}.json
done
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-overlap 2>/dev/null | line > $OUT/cfg3_seq.json
python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | line > $OUT/cfg3.json
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); r = {s["stage"]: s for s in d["roofline_stages"]}
        print("%-40s %8.1f fps %.3f ms/frame  live: tracker %.3f march %.3f  alone: %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r.get("tracker", {}).get("kernel_ms", 0), r["march"]["kernel_ms"],
              {k: round(v, 3) for k, v in (d.get("stages_sequential") or {}).items() if k != "note"}))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
