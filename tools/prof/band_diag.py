"""Diag-variant library (-DSVO_BRICK_DIAG): cycles per wavefront-step of the brick march, split into
before-the-entries-are-needed / waiting-for-them / after, for the whole image and for single 16-row bands."""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
depth, rgb = synth.render_stream(K, W, H, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(K)]
P = pl.SlamPipeline(W, H, D, (0, 1.5, 0), edge, pool_capacity_nodes=(1 << 30) - 8)
P.run_stream(depth, rgb, list(range(K)), views)
torch.cuda.synchronize()
img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
view = views[K - 1]
cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
def show(name, fn):
    fn(None); torch.cuda.synchronize()
    cnt.zero_(); fn(cnt); torch.cuda.synchronize()
    c = cnt.tolist()
    ws = max(c[7], 1)
    print("%-10s steps %9d fast %9d rare %7d | wave-steps %7d | cycles/wave-step: pre %6.0f wait %6.0f post %6.0f" %
          (name, c[0], c[2], c[3], c[7], c[4] / ws, c[5] / ws, c[6] / ws))
show("full", lambda c: pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0, c))
cnt.zero_(); pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0x200, cnt); torch.cuda.synchronize()
c = cnt.tolist(); nw = W * H // 64
print("full: per wavefront (lane 0): LDS tables %.0f cycles, ray set-up %.0f, loop %.0f, epilogue (last sample's walk, pixel) %.0f; waves %d"
      % (c[2] / nw, c[3] / nw, c[4] / nw, c[5] / nw, nw))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, rows in (("full", H), ("half", H // 2), ("quarter", H // 4), ("eighth", H // 8)):
    f = lambda: pkg.cone_trace_svo_band(img, 0, rows, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0)
    f(); torch.cuda.synchronize(); ev0.record()
    for _ in range(10): f()
    ev1.record(); torch.cuda.synchronize()
    print("rows 0..%d: %.4f ms per render" % (rows, ev0.elapsed_time(ev1) / 10))
for first in ((0, 96, 240, 320, 384, 448) if H == 480 else (0, 256, 512, 768, 1024)):
    show("band %d" % first, lambda c: pkg.cone_trace_svo_band(img, first, 16, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0, c))
for first in ((384,) if H == 480 else ()):
    show("rows %d+8" % first, lambda c: pkg.cone_trace_svo_band(img, first, 8, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0, c))
