#!/bin/bash
# Regenerates the measurement records of a round on an MI355X box:  bash tools/prof/profile_round.sh r05 [quick]
# Everything lands in gpurun_out/<tag>/ (scratch); the files worth judging are then copied to profiles/.
# Counter passes use the SEQUENTIAL form of the bench (--no-overlap, launch-chain tracker): counter collection serialises
# dispatches, which would deadlock the multi-stream pipeline; per-kernel traffic does not depend on the overlap.  Only
# the library's kernels are counted (--kernel-include-regex svoslam: the torch kernels that generate the synthetic
# stream made the round-2 passes run into their timeouts).  A failed pass FAILS the script (no `|| echo`).
set -u
TAG=${1:-r05}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
SCR=/tmp/svoslam_prof; mkdir -p $SCR
cd /tmp; export TMPDIR=/tmp
export SVOSLAM_BENCH_FULL_LINE=1   # the files under profiles/ hold the FULL record (bench.py prints a <= 6000-byte line by default)
line() { grep '^{"metric"' | tail -1; }
P="$OUT/$TAG"
FAILED=0
fail() { echo "FAILED: $*"; FAILED=1; }

echo "== PMC FETCH_SIZE / WRITE_SIZE per kernel (separate passes)"
PM=${P}_pmc_fetch_write_per_kernel.txt
echo "# rocprofv3 --pmc <counter> --kernel-include-regex svoslam --kernel-trace -- python bench.py --workload W --no-overlap ... (SVOSLAM_CONFIG=track_mode=1); one counter per pass; KB per dispatch, all dispatches" > $PM
declare -A FR TI ARGS
FR[cfg3]=300; TI[cfg3]=10; ARGS[cfg3]="--steps 10 --warmup 2 --map-frames 300 --repeats 1"
FR[cfg4]=20;  TI[cfg4]=10; ARGS[cfg4]="--steps 10 --warmup 10 --map-frames 0 --repeats 1"
SPECS=""
for W in cfg3 cfg4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    D=$SCR/pm_${W}_$c; rm -rf $D; mkdir -p $D
    SVOSLAM_CONFIG=track_mode=1 timeout 900 rocprofv3 --pmc $c --kernel-include-regex svoslam --kernel-trace --output-format csv -d $D -o p -- \
      python $R/bench.py --workload $W ${ARGS[$W]} --no-overlap --no-cpu-baseline --allow-missing-traffic > $SCR/pm_${W}_$c.log 2>&1 || fail "pmc pass $W $c (tail: $(tail -2 $SCR/pm_${W}_$c.log))"
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    [ -n "$f" ] || fail "pmc pass $W $c left no counter_collection.csv"
    cp "$f" $SCR/${W}_$c.csv
    python3 - "$f" "$c" "$W" <<'PY' >> $PM
import csv, sys, collections
f, cname, w = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == cname:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%s,%s,%s,calls=%d,mean=%.1f,total=%.1f" % (w, cname, k, len(v), sum(v) / len(v), sum(v)))
PY
  done
  SPEC="$W:${FR[$W]}:${TI[$W]}:$SCR/${W}_FETCH_SIZE.csv:$SCR/${W}_WRITE_SIZE.csv"
  # the one-launch tracker (cfg4: its streaming form), on its own (the bench pass above runs the launch chain)
  NT=12; [ $W = cfg4 ] && NT=6
  for c in FETCH_SIZE WRITE_SIZE; do
    D=$SCR/pt_${W}_$c; rm -rf $D; mkdir -p $D
    timeout 300 rocprofv3 --pmc $c --kernel-include-regex track_persistent --kernel-trace --output-format csv -d $D -o p -- \
      python $R/tools/prof/track_only.py $NT $W > $SCR/pt_${W}_$c.log 2>&1 || fail "pmc pass one-launch tracker $W $c"
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    [ -n "$f" ] || fail "tracker pass $W $c left no csv"
    cp "$f" $SCR/trk_${W}_$c.csv
    python3 -c "
import csv,sys
v=[float(r['Counter_Value']) for r in csv.DictReader(open('$f')) if r['Counter_Name']=='$c']
print('$W,$c,svoslam::track_persistent_kernel (tools/prof/track_only.py $NT $W),calls=%d,mean=%.1f,total=%.1f'%(len(v),sum(v)/max(1,len(v)),sum(v)))" >> $PM
  done
  SPEC="$SPEC:$SCR/trk_${W}_FETCH_SIZE.csv:$SCR/trk_${W}_WRITE_SIZE.csv:$((NT-1))"
  SPECS="$SPECS $SPEC"
done
python3 $R/tools/prof/pmc_to_json.py "$OUT/pmc_traffic.json" $TAG $SPECS || fail "pmc_to_json"
# the bench lines below read the traffic from profiles/pmc_traffic.json: install the fresh one in this box's copy
[ $FAILED = 0 ] && cp "$OUT/pmc_traffic.json" $R/profiles/pmc_traffic.json

echo "== bench lines"
python $R/bench.py 2>/dev/null | line > ${P}_bench_cfg3.json
python $R/bench.py --steps 20 --warmup 5 2>/dev/null | line > ${P}_bench_cfg3_driver_args_20frames.json
python $R/bench.py --workload cfg4 --steps 40 --warmup 5 2>/dev/null | line > ${P}_bench_cfg4.json
python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --stages 2>/dev/null | line > ${P}_bench_cfg3_with_stages.json
python $R/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --stages 2>/dev/null | line > ${P}_bench_cfg4_with_stages.json
if [ -z "$QUICK" ]; then
  python $R/bench.py --steps 100 --warmup 5 --map-frames 0 --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_young_map_105frames.json
  python $R/bench.py --steps 300 --warmup 0 --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_all_300_frames_timed.json
  python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --include-h2d 2>/dev/null | line > ${P}_bench_cfg3_include_h2d.json
  for e in 0/2 0/4 0/8 3/8; do
    python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>/dev/null | line > ${P}_bench_cfg3_emulated_rank_$(echo $e | sed "s#/#_of_#").json
  done
  for e in 0/2 3/8; do
    python $R/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline --emulate-rank $e 2>/dev/null | line > ${P}_bench_cfg4_emulated_rank_$(echo $e | sed "s#/#_of_#").json
  done
fi
for f in ${P}_bench_*.json; do python3 - "$f" <<'PY'
import json, sys, os
try:
    d = json.load(open(sys.argv[1]))
    r = d['roofline']
    print('%-56s %8.1f fps  dominant %s %.3f ms frac %.3f traffic %s' % (os.path.basename(sys.argv[1]), d['value'], r['kernel'].split(' ')[0], r['kernel_ms'], r['frac'],
          ('%.1f MB' % (r['traffic'] / 1e6)) if r.get('traffic') else r.get('traffic')))
    for s in d.get('roofline_stages', []):
        print('      %-8s %.3f ms  alg %.1f MB  frac %.3f  traffic %s' % (s['stage'], s['kernel_ms'], s['alg_bytes_per_launch'] / 1e6, s['frac'],
              ('%.1f MB' % (s['traffic'] / 1e6)) if s.get('traffic') else s.get('traffic')))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "unreadable:", e); sys.exit(0)
PY
done
[ -s ${P}_bench_cfg3.json ] || fail "default bench line empty"

echo "== mesh configurations (BASELINE configs 2 and 5): stage times + rooflines, kernel statistics, PMC traffic per kernel"
for C in cfg2 cfg5; do
  python $R/tools/mesh_bench.py --config $C 2>/dev/null | grep '^{"config"' | tail -1 > ${P}_mesh_bench_$C.json
  [ -s ${P}_mesh_bench_$C.json ] || fail "mesh bench $C"
  D=$SCR/ks_mesh_$C; rm -rf $D; mkdir -p $D
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python $R/tools/mesh_bench.py --config $C --reps 2 > $SCR/ks_mesh_$C.log 2>&1 || fail "kernel stats mesh $C"
  f=$(find $D -name "*kernel_stats.csv" | sort | tail -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python tools/mesh_bench.py --config $C --reps 2   (2 x (voxelize + svo_from_voxel_grid) + 3 views x 2 modes x 6 renders)"
    head -1 $f
    grep -v -E "at::native|rocclr|^\"Name" $f; } > ${P}_mesh_${C}_kernel_stats.csv
  MP=${P}_mesh_pmc_fetch_write_per_kernel.txt
  [ $C = cfg2 ] && echo "# rocprofv3 --pmc <counter> --kernel-include-regex svoslam --kernel-trace -- python tools/mesh_bench.py --config C --reps 1; one counter per pass; KB per dispatch" > $MP
  for c in FETCH_SIZE WRITE_SIZE; do
    D=$SCR/pm_mesh_${C}_$c; rm -rf $D; mkdir -p $D
    timeout 900 rocprofv3 --pmc $c --kernel-include-regex svoslam --kernel-trace --output-format csv -d $D -o p -- python $R/tools/mesh_bench.py --config $C --reps 1 > $SCR/pm_mesh_${C}_$c.log 2>&1 || fail "pmc pass mesh $C $c"
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    [ -n "$f" ] && python3 - "$f" "$c" "$C" <<'PY' >> $MP
import csv, sys, collections
f, cname, w = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == cname:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%s,%s,%s,calls=%d,mean=%.1f,total=%.1f" % (w, cname, k, len(v), sum(v) / len(v), sum(v)))
PY
  done
done
python3 - ${P}_mesh_bench_cfg2.json ${P}_mesh_bench_cfg5.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        m = json.load(open(f))
        print(m["config"], "voxels", m["voxels"], "fragments", m["fragments"], "Mrays/s", m["mrays_per_s_min_max"])
        for k, v in m["stages"].items(): print("    %-22s %9.3f ms  frac %.4f" % (k, v["ms"], v["frac"] or 0))
    except Exception as e:
        print(f, "unreadable", e)
PY

echo "== kernel stats (rocprofv3 --kernel-trace --stats) of the bench"
for W in cfg3 cfg4; do
  A="--steps 20 --warmup 5 --repeats 1 --no-corrected-line"; [ $W = cfg4 ] && A="--workload cfg4 --steps 40 --warmup 5 --repeats 1 --no-corrected-line"
  D=$SCR/ks_$W; rm -rf $D; mkdir -p $D
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python $R/bench.py $A --no-cpu-baseline > $SCR/ks_$W.log 2>&1 || fail "kernel stats $W"
  f=$(find $D -name "*kernel_stats.csv" | sort | tail -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $A --no-cpu-baseline   (all dispatches of the run: map history + warm-up + timed + the sequential stage pass)"
    echo "# bench line of this profiled run: $(grep '^{"metric"' $SCR/ks_$W.log | tail -1)"
    head -1 $f
    grep -v -E "at::native|rocclr|^\"Name" $f; } > ${P}_bench_${W}_kernel_stats.csv
  # the timed frames alone: mean duration of each kernel's LAST dispatches, from the kernel trace
  t=$(find $D -name "*kernel_trace.csv" | sort | tail -1)
  [ -n "$t" ] && python3 - "$t" > ${P}_bench_${W}_kernel_trace_timed_frames.txt <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "svoslam" not in n: continue
    rows[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("# per kernel: dispatches, mean duration over ALL dispatches, mean over the LAST 20 % (old map: the timed frames), us")
for n, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    d = [x for _, x in v]
    tail = d[-max(1, len(d) // 5):]
    print("%-70s %7d %9.2f %9.2f" % (n[:70], len(d), sum(d) / len(d) / 1e3, sum(tail) / len(tail) / 1e3))
PY
  grep -E "cone_trace|track_persistent|fill_mip|mip_straddle|icp_accumulate" ${P}_bench_${W}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done

if [ -z "$QUICK" ]; then
echo "== A/B records of the round's switches (same box): occupancy bricks, deferred commits, tile order of the brick march"
SVOSLAM_CONFIG=march_bricks=0 python $R/bench.py --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_ab_bricks_off.json
SVOSLAM_CONFIG=runner_deferred=0 python $R/bench.py --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_ab_inplace_commits.json
SVOSLAM_CONFIG=march_bricks=0,runner_deferred=0 python $R/bench.py --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_ab_round2_march_and_schedule.json
SVOSLAM_CONFIG=march_bricks=0 python $R/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg4_ab_bricks_off.json
SVOSLAM_CONFIG=track_stream=0 python $R/bench.py --workload cfg4 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg4_ab_chain_tracker.json
python $R/bench.py --tracker corrected --no-cpu-baseline 2>/dev/null | line > ${P}_bench_cfg3_corrected_tracker.json
for f in ${P}_bench_cfg3_ab_*.json; do python3 -c "import json,sys,os; d=json.load(open('$f')); print('%-60s %8.1f fps  march %.3f ms' % (os.path.basename('$f'), d['value'], d['roofline_stages'][0]['kernel_ms']))"; done
echo "== brick march anatomy (diag variant of the library: -DSVO_BRICK_DIAG, built by tools/prof/build_diag_variant.sh)"
if [ -f $R/octree-slam_amd/_variants/libsvoslam_hip_diag.so ]; then
  cp $R/octree-slam_amd/libsvoslam_hip.so /tmp/base.so
  cp $R/octree-slam_amd/_variants/libsvoslam_hip_diag.so $R/octree-slam_amd/libsvoslam_hip.so
  { echo "# tools/prof/band_diag.py 300 with the -DSVO_BRICK_DIAG library: cycles (clock64) per wavefront-step of cone_trace_brick_kernel on the 300-frame cfg3 map,"
    echo "# before the entries are needed / waiting for them (forced s_waitcnt) / after; 'full' = whole image (the waits of a full chip are other wavefronts'"
    echo "# issue slots, and the per-lane diagnostic atomics), 'band N' = rows N..N+15 alone (a lone wavefront per SIMD: the tail's regime)"
    python $R/tools/prof/band_diag.py 300 2>&1 | grep -v amdgpu.ids; } > ${P}_brick_march_anatomy.txt
  cp /tmp/base.so $R/octree-slam_amd/libsvoslam_hip.so
  tail -9 ${P}_brick_march_anatomy.txt
fi
echo "== the schedule's two steady states under the kernel trace (host 1 / 4 commits ahead)"
bash $R/tools/prof/state_trace.sh > /dev/null 2>&1
for l in 1 4; do [ -s $R/gpurun_out/state/frames_lead$l.txt ] && cp $R/gpurun_out/state/frames_lead$l.txt ${P}_steady_states_kernel_trace_lead$l.txt; done
cd /tmp
echo "== march anatomy, scheduler timeline"
python $R/tools/prof/ray_anatomy.py 300 2>&1 | grep -v amdgpu.ids > ${P}_ray_anatomy_cfg3_300frames.txt; tail -3 ${P}_ray_anatomy_cfg3_300frames.txt
python $R/tools/prof/runner_timeline.py 2>&1 | grep -v amdgpu.ids > ${P}_runner_timeline_cfg3.txt; tail -4 ${P}_runner_timeline_cfg3.txt
TIMELINE_WORKLOAD=cfg4 TIMELINE_FRAMES=60 python $R/tools/prof/runner_timeline.py 2>&1 | grep -v amdgpu.ids > ${P}_runner_timeline_cfg4.txt; tail -4 ${P}_runner_timeline_cfg4.txt
python $R/tools/prof/ramp_timeline.py 2>&1 | grep -v amdgpu.ids > ${P}_ramp_timeline_cfg3_driver_20frames.txt; head -1 ${P}_ramp_timeline_cfg3_driver_20frames.txt
echo "== one-launch tracker: in-kernel stamps (-DSVO_TRK_PROF variant) and SQ counters of the 1080p streaming form"
if [ -f $R/octree-slam_amd/_variants/libsvoslam_hip_trkprof.so ]; then
  cp $R/octree-slam_amd/libsvoslam_hip.so /tmp/base.so
  cp $R/octree-slam_amd/_variants/libsvoslam_hip_trkprof.so $R/octree-slam_amd/libsvoslam_hip.so
  { echo "# tools/prof/tracker_profile.py with the -DSVO_TRK_PROF library: clock64 stamps of the solver workgroup (s.*) and of worker 0 (w.*) per epoch (19 ICP iterations), 640x480"
    python $R/tools/prof/tracker_profile.py 2>&1 | grep -v amdgpu.ids
    echo "# the same at 1920x1080 (streaming form)"
    python $R/tools/prof/tracker_profile.py 1920 1080 2>&1 | grep -v amdgpu.ids; } > ${P}_tracker_stamps.txt
  cp /tmp/base.so $R/octree-slam_amd/libsvoslam_hip.so
fi
{ echo "# rocprofv3 --pmc <4 SQ counters per pass> --kernel-include-regex track_persistent --kernel-trace -- python tools/prof/track_only.py 6 cfg4: the streaming one-launch tracker at 1920x1080 alone on the GPU; per launch, mean of 5; WAVE_CYCLES / WAIT_* / ACTIVE_* in quad-cycles"
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
    D=$SCR/sq_trk; rm -rf $D; mkdir -p $D
    timeout 300 rocprofv3 --pmc $set --kernel-include-regex track_persistent --kernel-trace --output-format csv -d $D -o p -- python $R/tools/prof/track_only.py 6 cfg4 > $SCR/sq_trk.log 2>&1 || { echo "FAILED $set"; continue; }
    f=$(find $D -name "*counter_collection.csv" | sort | tail -1)
    python3 -c "
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open('$f')): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print('%-24s calls=%d mean=%.0f' % (k, len(v), sum(v) / len(v)))"
  done; } > ${P}_tracker_sq_counters_cfg4_column_major_rows.txt
fi
ls -la "$OUT"
echo "profile_round: FAILED=$FAILED"
exit $FAILED
