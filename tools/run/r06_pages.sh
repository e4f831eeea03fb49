#!/bin/bash
# the paged brick field: parity tests, page counts, A/B against the dense-field library (variant "dense" = the commit before)
O=gpurun_out/r06k; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
timeout 1500 python -m pytest tests/test_gpu_bricks.py tests/test_gpu_render.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
python - <<'PY'
import json, subprocess, sys
for a in (["--steps","20","--warmup","5"], ["--workload","cfg4","--steps","40","--warmup","5","--repeats","3"]):
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-other-configs", "--lean"] + a, capture_output=True, text=True)
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print(a[1] if a[0]=="--workload" else "cfg3", round(d["value"],1), json.dumps(d["config"]["device_memory_GiB"]))
    except Exception as e:
        print("FAILED", a, r.stderr[-800:])
PY
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ss = d.get('stages_sequential', {})
print('%.1f (%.1f..%.1f) march %.4f trk %.4f | alone: march %.3f' % (d['value'], d['value_min'], d['value_max'], d['roofline_stages'][0]['kernel_ms'], d['roofline_stages'][1]['kernel_ms'], ss.get('march_ms', 0)))"; }
# dense = the tree of the commit before (a copy under _ab_head/, git-ignored): its own bench.py, binding and library
{
for rep in 1 2 3; do
  for v in dense paged; do
    B=bench.py; [ $v = dense ] && B=_ab_head/bench.py
    echo -n "$v rep $rep  20: "; python $B --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "$v rep $rep 100: "; python $B --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
    echo -n "$v rep $rep cfg4: "; python $B --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  done
done
} 2>&1 | tee $O/ab.txt
echo "== dense: render_only 300"; python _ab_head/tools/prof/render_only.py 300 2>&1 | grep "mode 0"
echo "== paged: render_only 300"; python tools/prof/render_only.py 300 2>&1 | grep "mode 0"
