#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_emu; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
for e in 0/2 0/4 0/8 3/8; do
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>/dev/null | line > $OUT/cfg3_rank_$(echo $e | sed "s#/#_of_#").json
done
SVOSLAM_RUNNER_TIMELINE=1 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --emulate-rank 3/8 --stages 2>/dev/null | line > $OUT/cfg3_rank_3_of_8_stages.json
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print("%-40s %8.1f fps %.3f ms/frame march %.3f  stages %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["roofline_stages"][0]["kernel_ms"], d.get("stages")))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
