#!/bin/bash
# packed sorts in the blocking insert and the mesh voxelizer, straddler prefetch, row scan: the whole GPU suite, then the mesh
# configurations' stage times and the two bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for c in cfg2 cfg5; do timeout 300 python tools/mesh_bench.py --config $c > $O/mesh_$c.json 2> $O/mesh_$c.err; done
timeout 300 python bench.py --workload cfg4 --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --lean 2>$O/cfg4.err | grep '^{"metric"' | tail -1 > $O/cfg4.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --lean 2>$O/cfg3.err | grep '^{"metric"' | tail -1 > $O/cfg3.json
python - <<PY
import json
for c in ("cfg2","cfg5"):
    try:
        m=json.load(open("$O/mesh_%s.json"%c))
        print(c, m["voxels"], m["fragments"], m["call_ms"]["voxelize_call_ms"], m["call_ms"]["svo_from_voxel_grid_call_ms"])
        for k,v in m["stages"].items(): print("   ", k, round(v["ms"],3), round(v["frac"],4), v.get("parts_ms"))
    except Exception as e: print(c, "FAILED", e)
for w in ("cfg3","cfg4"):
    d=json.load(open("$O/%s.json"%w)); print(w, d["value"], d["runs"], d["stages_sequential"])
PY
