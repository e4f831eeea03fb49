#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_bricks.py tests/test_gpu_mesh.py tests/test_gpu_configs.py tests/test_gpu_render.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 300 python tools/mesh_bench.py --config cfg5 --reps 2 2>/dev/null > $O/mesh_cfg5.json
python - <<PY
import json
m=json.load(open("$O/mesh_cfg5.json"))
for k,v in m["stages"].items(): print("   ", k, round(v["ms"],3), round(v["frac"],4), v.get("parts_ms"))
for r in m["renders"]: print("   ", r["view"], r["mode"], r["trace_kernel_ms"], r["Mrays_per_s"], round(r["frac"],3), r.get("bands_equal_full"))
print(m["render_kernel"])
PY
