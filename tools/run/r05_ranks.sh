#!/bin/bash
# one rank of 8 emulated on one GPU: owner-sorted frames and graph replay on / off, both workloads; single-GPU lines of the same box beside them
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
L() { grep '^{"metric"' | tail -1; }
timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | L > $O/cfg3_single.json
timeout 300 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --lean 2>/dev/null | L > $O/cfg4_single.json
for ss in 0 1; do for g in 0 1; do
  SVOSLAM_SHARD_SORT=$ss SVOSLAM_CONFIG=graphs=$g timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --emulate-rank 3/8 2>$O/e3.err | L > $O/cfg3_rank3of8_sort${ss}_graphs$g.json
  SVOSLAM_SHARD_SORT=$ss SVOSLAM_CONFIG=graphs=$g timeout 300 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --emulate-rank 3/8 2>$O/e4.err | L > $O/cfg4_rank3of8_sort${ss}_graphs$g.json
done; done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.load(open(f)); print("%-44s %8.1f  %s" % (os.path.basename(f), d["value"], [round(x) for x in d["runs"]]))
    except Exception as e: print(os.path.basename(f), "FAILED", e)
PY
