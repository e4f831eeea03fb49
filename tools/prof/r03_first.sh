#!/bin/bash
# round-3 first GPU call: the suite on the reworked measurement code, the new bench line, and how long a PMC pass takes
# when only the library's kernels are counted (--kernel-include-regex) -- decides the shape of profile_round.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/r03_first"; mkdir -p "$OUT"
cd $R
( time python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
python bench.py --allow-missing-traffic > $OUT/bench_default.log 2>&1
python bench.py --steps 20 --warmup 5 --allow-missing-traffic --no-cpu-baseline > $OUT/bench_driver.log 2>&1
python bench.py --workload cfg4 --steps 40 --warmup 5 --allow-missing-traffic --no-cpu-baseline > $OUT/bench_cfg4.log 2>&1
cd /tmp; export TMPDIR=/tmp
S=/tmp/pm1; mkdir -p $S
( time SVOSLAM_GRAPHS=0 SVOSLAM_TRACK_CHAIN=1 timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "svoslam" --kernel-trace --output-format csv -d $S -o p -- \
    python $R/bench.py --no-overlap --steps 6 --warmup 2 --map-frames 40 --no-cpu-baseline --allow-missing-traffic ) > $OUT/pmc_probe.log 2>&1
f=$(find $S -name "*counter_collection.csv" | sort | tail -1)
[ -n "$f" ] && { wc -l $f; head -3 $f; grep -c cone_trace $f; } >> $OUT/pmc_probe.log 2>&1
# the one-launch tracker under counter collection (its workgroups wait for each other: does it survive serialised dispatch?)
S=/tmp/pm2; mkdir -p $S
( time timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "track_persistent" --kernel-trace --output-format csv -d $S -o p -- \
    python $R/tools/prof/track_only.py 8 ) > $OUT/pmc_persistent_probe.log 2>&1
f=$(find $S -name "*counter_collection.csv" | sort | tail -1)
[ -n "$f" ] && { wc -l $f; grep track_persistent $f | head -3; } >> $OUT/pmc_persistent_probe.log 2>&1
tail -3 $OUT/*.log
