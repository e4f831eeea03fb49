#!/bin/bash
# end-of-round check on one box: the whole GPU suite, smoke(), the driver's command, a two-rank run of bench.py on one device (gloo)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/bench_driver.time
grep real $O/bench_driver.time
python - <<PY
import json
d=json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
print("driver line:", round(d["value"],1), [round(x) for x in d["runs"]], "roofline", d["roofline"]["kernel"].split()[0], round(d["roofline"]["frac"],3), "wall", {k:round(v,1) for k,v in d["wall"].items() if k!="note"})
oc=d["other_configs"]; print("  cfg4", round(oc["cfg4"]["value"],1), "cfg2 voxelize", round(oc["cfg2"]["stages"]["voxelize"]["ms"],3), "cfg5 voxelize", round(oc["cfg5"]["stages"]["voxelize"]["ms"],2), "svo", round(oc["cfg5"]["stages"]["svo_from_voxel_grid"]["ms"],2))
PY
SVOSLAM_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline 2>$O/two_ranks.err | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('two ranks on one device (gloo):', round(d['value'],1), d['n_gpus'], d['config']['parallelism'][:60], 'other:', (d.get('other_partition') or {}).get('value'))"
