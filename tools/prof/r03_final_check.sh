#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_final_check; mkdir -p $OUT; cd $R
( time python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -20 ) > $OUT/pytest_gpu.log 2>&1; tail -16 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py --steps 20 --warmup 5 ) > $OUT/bench_driver.log 2>&1; grep '^{"metric"' $OUT/bench_driver.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"; tail -4 $OUT/bench_driver.log | grep real
