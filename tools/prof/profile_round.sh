#!/bin/bash
# Regenerates the measurement records of a round on an MI355X box:  bash tools/prof/profile_round.sh r02
# Everything lands in gpurun_out/<tag>/ (scratch); the files worth judging are then copied to profiles/.
# Counter passes use the SEQUENTIAL form of the bench (--no-overlap, SVOSLAM_GRAPHS=0, launch-chain tracker): counter
# collection serialises kernels, which would deadlock the multi-stream pipeline and the one-launch tracker's spins;
# per-kernel traffic does not depend on the overlap.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
SCR=/tmp/svoslam_prof; mkdir -p $SCR
cd /tmp; export TMPDIR=/tmp
line() { grep '^{"metric"' | tail -1; }
P="$OUT/$TAG"

echo "== bench lines"
python $R/bench.py --steps 100 --warmup 5 2>/dev/null | line > ${P}_bench_cfg3.json
python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --stages 2>/dev/null | line > ${P}_bench_cfg3_with_stages.json
python $R/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --stages 2>/dev/null | line > ${P}_bench_cfg4_with_stages.json
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_driver_args_20frames.json
python $R/bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_300frames.json
python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --include-h2d 2>/dev/null | line > ${P}_bench_cfg3_include_h2d.json
python $R/bench.py --workload cfg4 --steps 40 --warmup 5 2>/dev/null | line > ${P}_bench_cfg4.json
SVOSLAM_FORCE_DIST=1 python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --exchange none 2>/dev/null | line > ${P}_bench_cfg3_forced_dist_none.json
SVOSLAM_FORCE_DIST=1 python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --exchange deltas 2>/dev/null | line > ${P}_bench_cfg3_forced_dist_deltas.json
for e in 0/2 1/2 0/4 2/4 0/8 3/8 7/8; do
  python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>/dev/null | line > ${P}_bench_cfg3_emulated_rank_$(echo $e | sed "s#/#_of_#").json
done
for e in 0/2 0/8 3/8; do
  python $R/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>/dev/null | line > ${P}_bench_cfg4_emulated_rank_$(echo $e | sed "s#/#_of_#").json
done
SVOSLAM_FORCE_DIST=1 python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --exchange allreduce 2>/dev/null | line > ${P}_bench_cfg3_forced_dist_allreduce.json
for f in ${P}_bench_*.json; do python3 - "$f" <<'PY'
import json, sys, os
try:
    d = json.load(open(sys.argv[1]))
    print('%-52s %8.1f fps  march %.3f ms  frac %.3f' % (os.path.basename(sys.argv[1]), d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "unreadable:", e)
PY
done

echo "== kernel stats (rocprofv3 --kernel-trace --stats) of the bench"
for W in cfg3 cfg4; do
  S=100; [ $W = cfg4 ] && S=40
  D=$SCR/ks_$W; mkdir -p $D
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python $R/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline > $SCR/ks_$W.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | sort | tail -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline"
    echo "# bench line of this profiled run: $(grep '^{"metric"' $SCR/ks_$W.log | tail -1)"
    head -1 $f
    grep -v -E "at::native|rocclr|^\"Name" $f; } > ${P}_bench_${W}_kernel_stats.csv
  grep -E "cone_trace|track_persistent|fill_mip|build_accel" ${P}_bench_${W}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-170
done

echo "== per-stage algorithmic bytes"
python $R/tools/prof/stage_bytes.py ${P}_bench_cfg3_kernel_stats.csv ${P}_stage_bytes_cfg3.txt 2>/dev/null | tail -8

echo "== PMC FETCH_SIZE / WRITE_SIZE per kernel (separate passes)"
echo "# rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --workload W --steps S --warmup 5 --no-cpu-baseline --no-overlap (SVOSLAM_GRAPHS=0 SVOSLAM_TRACK_CHAIN=1); one counter per pass; bytes per launch" > ${P}_pmc_fetch_write_per_kernel.txt
for W in cfg3 cfg4; do
  S=60; [ $W = cfg4 ] && S=20
  for c in FETCH_SIZE WRITE_SIZE; do
    D=$SCR/pm_${W}_$c; mkdir -p $D
    SVOSLAM_GRAPHS=0 SVOSLAM_TRACK_CHAIN=1 timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o p -- python $R/bench.py --workload $W --steps $S --warmup 5 --no-cpu-baseline --no-overlap > $SCR/pm.log 2>&1 || { echo "pass $W $c failed/timeout"; tail -3 $SCR/pm.log; }
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    [ -n "$f" ] && python3 - "$f" "$c" "$W" <<'PY' >> ${P}_pmc_fetch_write_per_kernel.txt
import csv, sys, collections
f, cname, w = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == cname:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "at::native" in k or "rocclr" in k: continue
        acc[k].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%s,%s,%s,calls=%d,mean=%.1f,total=%.1f" % (w, cname, k, len(v), sum(v) / len(v), sum(v)))
PY
  done
done
grep cone_trace ${P}_pmc_fetch_write_per_kernel.txt
python3 - ${P}_pmc_fetch_write_per_kernel.txt "$OUT/pmc_traffic.json" $TAG <<'PY'
import sys, json, re
src, dst, tag = sys.argv[1:4]
out = {"_comment": "HBM traffic of the dominant kernel from rocprofv3 PMC passes (one counter per pass); KB per launch, mean over the launches of the pass. bench.py reports traffic = (2*fetch_kb + write_kb)*1024 bytes: FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated there on wide coalesced reads; this kernel issues 8-byte gathers, so the factor is an upper bound here)."}
for line in open(src):
    m = re.match(r"(cfg\d),(FETCH_SIZE|WRITE_SIZE),([^,]*cone_trace_kernel.*),calls=\d+,mean=([\d.]+)", line)
    if m:
        d = out.setdefault(m.group(1), {}).setdefault("cone_trace_kernel", {"source": "profiles/%s_pmc_fetch_write_per_kernel.txt" % tag})
        d["fetch_kb" if m.group(2) == "FETCH_SIZE" else "write_kb"] = float(m.group(4))
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_comment"}))
PY

echo "== cache counters of the march alone (standalone renders of the 100-frame map)"
CC=${P}_cone_trace_cache_counters.txt
echo "# rocprofv3 --pmc <counters> --kernel-trace -- python tools/prof/render_only.py 100 ; cone_trace_kernel launches only; mean per launch" > $CC
i=0
for c in "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum" "TCC_HIT_sum" "TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"; do   # (the last pass sometimes runs into its 100 s limit on this pool: "pass ... failed" in the file then)
  i=$((i+1)); D=$SCR/cc_$i; mkdir -p $D
  timeout 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o p -- python $R/tools/prof/render_only.py 100 > $SCR/cc.log 2>&1 || { echo "pass $c failed" >> $CC; tail -2 $SCR/cc.log; }
  f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $CC
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "cone_trace_kernel" in k: acc[(k[-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%s,%s,calls=%d,mean=%.1f" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done
tail -30 $CC

echo "== march anatomy, scheduler timeline, tracker hand-off profile"
python $R/tools/prof/ray_anatomy.py 105 2>&1 | grep -v amdgpu.ids > ${P}_ray_anatomy_cfg3_105frames.txt; tail -3 ${P}_ray_anatomy_cfg3_105frames.txt
python $R/tools/prof/runner_timeline.py 2>&1 | grep -v amdgpu.ids > ${P}_runner_timeline_cfg3.txt; tail -4 ${P}_runner_timeline_cfg3.txt
python $R/tools/prof/runner_timeline.py 3/8 2>&1 | grep -v -E "amdgpu.ids|RCCL|HIP version|ROCm version|Hostname|Librccl" > ${P}_runner_timeline_cfg3_emulated_rank_3_of_8.txt; tail -4 ${P}_runner_timeline_cfg3_emulated_rank_3_of_8.txt
SVOSLAM_RUNNER_REPLICAS=2 python $R/tools/prof/runner_timeline.py 2>&1 | grep -v amdgpu.ids > ${P}_runner_timeline_cfg3_two_replicas.txt; tail -4 ${P}_runner_timeline_cfg3_two_replicas.txt
ls -la "$OUT"
