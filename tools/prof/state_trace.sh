#!/bin/bash
# the two steady states of the cfg3 schedule side by side: the bench under rocprofv3 --kernel-trace with the host 1 and 4 commits ahead
# (4 = always the slow state), three consecutive frames near the end of each run as one table per queue: start, duration, gap to the
# previous kernel of the same queue; then per queue and frame the busy time and the idle time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/state"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for lead in ${LEADS:-1 4}; do
  S=/tmp/st$lead; rm -rf $S; mkdir -p $S
  SVOSLAM_CONFIG=runner_lead=$lead timeout 600 rocprofv3 --kernel-trace --output-format csv -d $S -o k -- python $R/bench.py --no-cpu-baseline --no-stage-pass "$@" > $OUT/bench_lead$lead.log 2>&1
  grep '^{"metric"' $OUT/bench_lead$lead.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('# runner_lead=$lead under rocprofv3 --kernel-trace:', round(d['value'],1), 'frames/s')" | tee $OUT/rate_lead$lead.txt
  t=$(find $S -name "*kernel_trace.csv" | head -1)
  cp $OUT/rate_lead$lead.txt $OUT/frames_lead$lead.txt
  python3 - $t <<'PY' >> $OUT/frames_lead$lead.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "svoslam" in r["Kernel_Name"]]
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
m = [i for i, r in enumerate(rows) if "cone_trace" in r["Kernel_Name"]]
per = [(rows[m[i + 1]]["s"] - rows[m[i]]["s"]) / 1e3 for i in range(len(m) - 40, len(m) - 2)]
print("# march-start periods of the last frames, us:", " ".join("%.0f" % p for p in per))
a, b = m[-14], m[-11]
t0 = rows[a]["s"]
sel = rows[a:b + 1]
qs = sorted(set(r["Queue_Id"] for r in sel))
short = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("svoslam::", "")[:40]
for q in qs:
    print("\n## queue", q)
    last = None
    for r in sel:
        if r["Queue_Id"] != q: continue
        gap = (r["s"] - last) / 1e3 if last is not None else 0.0
        print("%-40s start %8.1f dur %7.1f gap %7.1f" % (short(r), (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap))
        last = r["e"]
span = (rows[b]["s"] - t0) / 1e3
print("\n# three frames = %.1f us; per queue: busy us, kernels, sum of gaps < 30 us (launch hand-offs), sum of longer gaps (waits for other streams)" % span)
for q in qs:
    ks = [r for r in sel if r["Queue_Id"] == q]
    busy = sum(r["e"] - r["s"] for r in ks) / 1e3
    g = [(ks[i + 1]["s"] - ks[i]["e"]) / 1e3 for i in range(len(ks) - 1)]
    print("queue %-3s busy %7.1f kernels %3d short gaps %7.1f long gaps %7.1f" % (q, busy, len(ks), sum(x for x in g if x < 30), sum(x for x in g if x >= 30)))
# concurrency: time-weighted number of kernels executing
ev = []
for r in sel: ev += [(r["s"], 1), (r["e"], -1)]
ev.sort()
cur, lastt, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[cur] += t - lastt; lastt = t; cur += d
tot = sum(hist.values())
print("# kernels executing at once, share of time:", " ".join("%d: %.0f%%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
PY
done
