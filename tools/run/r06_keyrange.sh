#!/bin/bash
# key-range sharded fusion: tests, then emulated ranks (one rank of N on one GPU) against the frame-sharded "deltas" scheme and the single GPU
O=gpurun_out/r06k; mkdir -p $O
export SVOSLAM_BENCH_FULL_LINE=1
[ -z "$KR_ONLY" ] && timeout 1200 python -m pytest tests/test_gpu_keyrange.py tests/test_gpu_sharded.py -x -q -s -m gpu 2>&1 | grep -E "passed|failed|error|world|Error|assert" | tail -20
line() { grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('keyrange') or {}
print('%.1f (%.1f..%.1f) frames/s; frames with shared records %s; delta bytes/frame all ranks mean %s max %s this rank %s' % (d['value'], d['value_min'], d['value_max'], k.get('frames_timed_with_records_above_the_splitter_level'), k.get('delta_bytes_all_ranks_per_frame_mean'), k.get('delta_bytes_all_ranks_per_frame_max'), k.get('delta_bytes_this_rank_per_frame_mean')))"; }
{
for wl in cfg3 cfg4; do
  st=40; [ $wl = cfg3 ] && st=100
  echo -n "$wl single GPU: "; python bench.py --workload $wl --steps $st --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean 2>/dev/null | line
  for ex in deltas keyrange; do
    for rk in 3/8 1/4 1/2; do [ -n "$KR_ONLY" ] && [ "$ex$rk" != "keyrange3/8" ] && [ "$ex$rk" != "keyrange1/2" ] && continue
      echo -n "$wl $ex rank $rk: "; timeout 900 python bench.py --workload $wl --steps $st --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs --lean --exchange $ex --emulate-rank $rk 2>$O/err_${wl}_${ex}_${rk/\//of}.txt | tee $O/bench_${wl}_${ex}_${rk/\//of}.json | line
    done
  done
done
} 2>&1 | tee $O/keyrange_emulated.txt
