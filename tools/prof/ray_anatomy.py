import sys, numpy as np, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import svoslam_pkg
pkg = svoslam_pkg.load()
import importlib
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, edge = 640, 480, 12, 4.096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 105
depth, rgb = synth.render_stream(K, W, H, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(K)]
P = pl.SlamPipeline(W, H, D, (0, 1.5, 0), edge, pool_capacity_nodes=(1 << 30) - 8)
P.run_stream(depth, rgb, list(range(K)), views)
torch.cuda.synchronize()
img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
view = views[K - 1]
pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0x100)
st = img.cpu().numpy().view(np.uint32).reshape(H, W).astype(np.int64)
print("pool nodes", P.pool.size)
print("steps: sum %d mean %.1f median %d p90 %d p99 %d max %d" % (st.sum(), st.mean(), np.median(st), np.percentile(st, 90), np.percentile(st, 99), st.max()))
t = st.reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
print("per-wave max: mean %.1f p50 %d p90 %d max %d ; wave-steps/ideal: %.2f ; total wave-steps %d" % (t.mean(), np.median(t), np.percentile(t, 90), t.max(), t.sum() * 64 / st.sum(), t.sum()))
hist = np.bincount(np.minimum(st.reshape(-1) // 10, 20))
print("histogram of steps/10:", hist.tolist())
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev0.record()
    for _ in range(n): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n
print("full render ms %.4f" % timeit(lambda: pkg.cone_trace_svo(img, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0)))
xs, ys = [], []
for first in range(0, H, 16):
    ms = timeit(lambda: pkg.cone_trace_svo_band(img, first, 16, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0), 10)
    mx = st[first:first + 16].max(); tot = st[first:first + 16].sum()
    xs.append(mx); ys.append(ms)
    print("band %3d: %.4f ms  max steps %3d  total steps %7d" % (first, ms, mx, tot))
xs, ys = np.array(xs, float), np.array(ys)
A = np.stack([np.ones_like(xs), xs], 1)
coef = np.linalg.lstsq(A, ys, rcond=None)[0]
print("fit: band ms = %.4f + %.5f * max_steps  (=> %.3f us per step on the critical path)" % (coef[0], coef[1], coef[1] * 1e3))
# VERDICT r04 item 5 asked whether the render is "a ~0.1 ms bulk plus a single-wavefront tail": parts of the image alone
long_rows = np.where(st.max(axis=1) >= 200)[0]
lo, hi = int(long_rows.min()) // 16 * 16, (int(long_rows.max()) // 16 + 1) * 16
n_long_rays, n_long_waves = int((st >= 200).sum()), int((t >= 200).sum())
print("rays with >= 200 steps: %d (%.1f %%), in %d of %d wavefronts (8x8 pixels), rows %d..%d" % (n_long_rays, 100.0 * n_long_rays / st.size, n_long_waves, t.size, lo, hi))
for a, b, what in ((0, lo, "rows above the long rays"), (lo, hi, "the rows that hold the long rays"), (0, H, "whole image")):
    if b > a:
        ms = timeit(lambda: pkg.cone_trace_svo_band(img, a, b - a, 45.0, view, P.pool.data_ptr, P.center, P.edge, 0), 10)
        print("rows %3d..%3d alone: %.4f ms  (%s: %d wavefronts, %d of them with a ray of >= 200 steps, total steps %d)" % (a, b, ms, what, (b - a) // 8 * (W // 8), int((t[a // 8:b // 8] >= 200).sum()), int(st[a:b].sum())))
