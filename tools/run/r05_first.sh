#!/bin/bash
# round-5 first GPU call: suite, the driver's bench command with other_configs, sequential kernel statistics of both workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/bench_driver.time
tail -c 600 $O/bench_driver.err
cat $O/bench_driver.time
cd /tmp; export TMPDIR=/tmp
for W in cfg3 cfg4; do
  S=/tmp/ks_$W; rm -rf $S; mkdir -p $S
  A="--steps 10 --warmup 2 --map-frames 300 --repeats 1"; [ $W = cfg4 ] && A="--steps 10 --warmup 10 --map-frames 0 --repeats 1"
  SVOSLAM_CONFIG=track_mode=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $S -o k -- python $R/bench.py --workload $W $A --no-overlap --no-cpu-baseline --allow-missing-traffic --no-other-configs > $O/ks_$W.log 2>&1
  f=$(find $S -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/seq_kernel_stats_$W.csv
done
python - <<PY
import csv
for w in ("cfg3","cfg4"):
    try:
        rows=list(csv.DictReader(open("$O/seq_kernel_stats_%s.csv"%w)))
    except Exception as e:
        print(w, e); continue
    print("==", w)
    for r in rows[:30]:
        if "at::native" in r["Name"] or "rocprim" in r["Name"]: continue
        print("%-60s calls %5s avg %8.1f us" % (r["Name"].replace("void ","").replace("svoslam::","")[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
