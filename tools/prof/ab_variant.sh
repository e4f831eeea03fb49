
mkdir -p gpurun_out/ab
L=octree-slam_amd/libsvoslam_hip.so
cp $L /tmp/base.so
for rep in 1 2; do
  for v in base spec; do
    if [ $v = base ]; then cp /tmp/base.so $L; else cp octree-slam_amd/_variants/libsvoslam_hip_$v.so $L; fi
    python tools/prof/render_only.py 100 > gpurun_out/ab/render_${v}_$rep.txt 2>&1
    python bench.py > gpurun_out/ab/bench_${v}_$rep.json 2>/dev/null
    python bench.py --steps 20 --warmup 5 > gpurun_out/ab/bench20_${v}_$rep.json 2>/dev/null
  done
done
cp octree-slam_amd/_variants/libsvoslam_hip_spec.so $L
python -m pytest tests/test_gpu_render.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" > gpurun_out/ab/pytest_spec.txt
cp /tmp/base.so $L
grep -h "standalone" gpurun_out/ab/render_*.txt
for f in gpurun_out/ab/bench*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['roofline']['kernel_ms'])"; done
cat gpurun_out/ab/pytest_spec.txt
