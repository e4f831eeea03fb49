#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_ninth; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_mailbox.py -m gpu -x -q --timeout 120 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -8 > $OUT/pytest_mailbox.log; cat $OUT/pytest_mailbox.log
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q --timeout 300 --durations=6 2>&1 | grep -vE "RCCL|HIP version|ROCm version|Hostname|Librccl|amdgpu.ids" | tail -16 > $OUT/pytest_dist.log; cat $OUT/pytest_dist.log
