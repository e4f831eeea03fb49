"""Per-ray step anatomy of the mesh configurations' renders (cfg2 view 1 is 6x slower per step than its
neighbours):  python tools/prof/mesh_ray_anatomy.py [cfg2|cfg5]
The step image comes from the march's diagnostic mode (mode | 0x100: the pixel holds its ray's step count)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import svoslam_pkg
pkg = svoslam_pkg.load()
import mesh_bench as mb

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
pool, center, size, depth, (W, H), views = mb.build_scene(cfg, pkg)
img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    ev0.record()
    for _ in range(n): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n
print("config %s: depth %d, %dx%d, root half-edge %.5f, centre %s, pool %d nodes" % (cfg, depth, W, H, size, np.asarray(center).tolist(), pool.size))
for vi, (name, view) in enumerate(views):
    for mode, mname in ((pkg.RENDER_REFERENCE, "reference"), (pkg.RENDER_CARRY, "carry")):
        pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, center, size, mode | 0x100)
        st = img.cpu().numpy().view(np.uint32).reshape(H, W).astype(np.int64)
        t = st.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
        ms = timeit(lambda: pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, center, size, mode))
        pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, center, size, mode)
        lit = int((img[..., :3].sum(-1) > 0).sum().item())
        print("view %d (%s) %s: %.4f ms; steps sum %d mean %.1f p50 %d p90 %d p99 %d p99.9 %d max %d; rays >= 200: %d, >= 500: %d, >= 1000: %d; "
              "wavefronts with a ray >= 500: %d of %d; wave-steps / ray-steps %.2f; lit pixels %d; %.3f us per step of the longest ray"
              % (vi, name, mname, ms, st.sum(), st.mean(), np.median(st), np.percentile(st, 90), np.percentile(st, 99), np.percentile(st, 99.9), st.max(),
                 int((st >= 200).sum()), int((st >= 500).sum()), int((st >= 1000).sum()), int((t >= 500).sum()), t.size, t.sum() * 64 / st.sum(), lit,
                 ms * 1e3 / st.max()))
