#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_configs.py tests/test_gpu_fusion.py tests/test_gpu_compat.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for c in cfg2 cfg5; do timeout 300 python tools/mesh_bench.py --config $c > $O/mesh_$c.json 2> $O/mesh_$c.err; done
python - <<PY
import json
for c in ("cfg2","cfg5"):
    try:
        m=json.load(open("$O/mesh_%s.json"%c))
        print(c, m["voxels"], m["fragments"], m["call_ms"]["voxelize_call_ms"], m["call_ms"]["svo_from_voxel_grid_call_ms"])
        for k,v in m["stages"].items(): print("   ", k, round(v["ms"],3), round(v["frac"],4), v.get("parts_ms"))
    except Exception as e: print(c, "FAILED", e)
PY
