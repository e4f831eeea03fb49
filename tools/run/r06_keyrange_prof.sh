#!/bin/bash
# kernel statistics of an emulated rank 3/8 with the key-range fusion (cfg4, cfg3)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for wl in cfg4 cfg3; do
  st=20; [ $wl = cfg3 ] && st=60
  D=/tmp/kr_$wl; rm -rf $D
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $R/bench.py --workload $wl --steps $st --warmup 5 --repeats 1 --no-cpu-baseline --no-other-configs --lean --exchange keyrange --emulate-rank 3/8 > $O/prof_$wl.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_keyrange_3of8_$wl.csv
  t=$(find $D -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 - "$t" "$st" > $O/kernel_tail_keyrange_3of8_$wl.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed frames = the last part of the trace: take the last 30 % of dispatches
cut = int(len(rows) * 0.7)
acc = collections.defaultdict(lambda: [0, 0.0])
t0, t1 = int(rows[cut]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
for r in rows[cut:]:
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][0] += 1; acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("# last 30 %% of dispatches: %d kernels over %.3f ms wall; per kernel: calls, total us, mean us" % (len(rows) - cut, (t1 - t0) / 1e6))
for k, (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-72s %6d %10.1f %8.2f" % (k, c, us, us / c))
PY
done
tail -45 $O/kernel_tail_keyrange_3of8_cfg4.txt
