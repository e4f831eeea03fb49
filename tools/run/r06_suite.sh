#!/bin/bash
# full GPU suite + the driver's command + mesh benches, with the library's defaults
O=gpurun_out/${1:-r06g}; mkdir -p $O
SVOSLAM_DEBUG_LDS=1 python -c "
import torch, numpy as np, svoslam_pkg
pkg = svoslam_pkg.load()
from oracle import oracle as ora
import __graft_entry__ as g
g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_tests.log 2>&1; tail -14 $O/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.out 2> $O/bench20.err; cp bench_details.json $O/bench20_details.json; wc -c $O/bench20.out; cut -c1-400 $O/bench20.out
