#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r03_dbg; mkdir -p $OUT; cd $R
timeout 900 python -X faulthandler -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | cut -c1-230 | tail -40 > $OUT/suite.log
tail -12 $OUT/suite.log
