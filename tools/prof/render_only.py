import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svoslam_pkg
pkg = svoslam_pkg.load()
import importlib
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, edge = 640, 480, 12, 4.096
if os.environ.get("DIAG_CFG4"):
    W, H, D = 1920, 1080, 14
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
depth, rgb = synth.render_stream(K, W, H, device="cuda")
P = pl.SlamPipeline(W, H, D, (0, 1.5, 0), edge)
for k in range(K):
    P.track(depth[k], rgb[k], k); P.backproject(depth[k]); P.fuse(rgb[k])
img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
view = pl.ground_truth_view(K - 1, synth)
cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
for mode in (0, 1):
    for _ in range(3):
        pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, mode)
    torch.cuda.synchronize()
    pkg.cone_trace_timing(True)
    for _ in range(20):
        pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, mode)
    ms, n = pkg.cone_trace_timing_read()
    cnt.zero_()
    pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, mode, cnt)
    torch.cuda.synchronize()
    print("mode", mode, "standalone trace ms %.4f" % (ms / n), "steps/levels", cnt[:2].tolist(), "extra", cnt[2:].tolist(), "nodes", P.pool.size)
