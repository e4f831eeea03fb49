"""march kernel time right after a commit (cold) and repeated (hot), cfg3 / cfg4 (DIAG_CFG4=1)"""
import sys, numpy as np, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svoslam_pkg
pkg = svoslam_pkg.load()
import importlib
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, edge = 640, 480, 12, 4.096
if os.environ.get("DIAG_CFG4"):
    W, H, D, edge = 1920, 1080, 14, 8.192
K = int(sys.argv[1]) if len(sys.argv) > 1 else 45
depth, rgb = synth.render_stream(K + 4, W, H, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(K + 4)]
P = pl.SlamPipeline(W, H, D, (0, 1.5, 0), edge, pool_capacity_nodes=(1 << 30) - 8, count_steps=bool(os.environ.get("COUNT")))
if os.environ.get("LIKE_BENCH"):
    P.run_stream(depth[:6], rgb[:6], list(range(6)), views[:6]); torch.cuda.synchronize(); P.reset()
    P.run_stream(depth[:5], rgb[:5], list(range(5)), views[:5]); torch.cuda.synchronize()
    P.run_stream(depth[5:K], rgb[5:K], list(range(5, K)), views[5:K])
else:
    P.run_stream(depth[:K], rgb[:K], list(range(K)), views[:K])
torch.cuda.synchronize()
print("map built: nodes", P.pool.size, "lost", getattr(P.cam, "lost_levels", None))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for k in range(K, K + 2):
    P.track(depth[k], rgb[k], k)
    P.fuse_frame(depth[k], rgb[k])
    torch.cuda.synchronize()
    pkg.stage_timing([pkg.STAGE_MARCH])
    ev[0].record(); P.render(views[k]); ev[1].record(); torch.cuda.synchronize()
    cold = pkg.stage_timing_read(pkg.STAGE_MARCH)[0]
    pkg.stage_timing([pkg.STAGE_MARCH])
    ev[2].record(); P.render(views[k]); ev[3].record(); torch.cuda.synchronize()
    hot = pkg.stage_timing_read(pkg.STAGE_MARCH)[0]
    pkg.stage_timing([])
    print("frame %d: march kernel cold %.4f ms (render call %.4f)   hot %.4f ms (render call %.4f)  nodes %d" %
          (k, cold, ev[0].elapsed_time(ev[1]), hot, ev[2].elapsed_time(ev[3]), P.pool.size))
