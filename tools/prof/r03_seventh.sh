#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_seventh; mkdir -p $OUT; cd $R
line() { grep '^{"metric"' | tail -1; }
timeout 800 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q --timeout 300 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -6 > $OUT/pytest.log; cat $OUT/pytest.log
for s in 1 0; do for e in 0/2 0/4 0/8 3/8; do
  SVOSLAM_SHARD_SORT=$s python bench.py --steps 100 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>$OUT/err_${s}.log | line > $OUT/cfg3_sort${s}_rank_$(echo $e | sed "s#/#_of_#").json
done; done
for s in 1 0; do for e in 0/2 3/8; do
  SVOSLAM_SHARD_SORT=$s python bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>/dev/null | line > $OUT/cfg4_sort${s}_rank_$(echo $e | sed "s#/#_of_#").json
done; done
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print("%-40s %8.1f fps %.3f ms/frame march %.3f" % (os.path.basename(f), d["value"], d["ms_per_step"], d["roofline_stages"][0]["kernel_ms"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
tail -3 $OUT/err_1.log
