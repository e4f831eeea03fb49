"""frame-to-frame against frame-to-model tracking on the synthetic stream: rotation angle of the estimated orientation against the
ground truth's yaw (0.1 degrees per frame), every 25 frames.   python tools/prof/model_drift.py [W H DEPTH FRAMES]"""
import sys, os, importlib, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, K = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (320, 240, 10, 300)
center, edge = (0.0, 1.5, 0.0), 4.096
depth, rgb = synth.render_stream(K, W, H, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(K)]


def angle(o):
    m = o.reshape(3, 3).astype(np.float64)
    return np.degrees(np.arccos(np.clip((np.trace(m) - 1.0) / 2.0, -1.0, 1.0)))


rows = {}
for mode in (False, True):
    P = pl.SlamPipeline(W, H, D, center, edge, pool_capacity_nodes=1 << 26, frame_to_model=mode, count_steps=mode)
    t0 = time.perf_counter()
    out = []
    for k in range(K):
        P.frame(depth[k], rgb[k], k, views[k])
        if k % 25 == 0 or k == K - 1:
            out.append((k, angle(P.cam.pose()[1]), int((P.model_depth != 0).sum().item()) / (W * H) if mode else 0.0))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rows[mode] = out
    print("%s: %d frames in %.2f s (%.1f frames/s, stage by stage on one stream); tracking lost %d levels%s" % (
        "frame-to-model" if mode else "frame-to-frame", K, el, K / el, P.cam.tracking_lost_count(),
        "; model accepted for %d frames; %.1f M model march steps" % (P.model_used, P.model_steps.item() / 1e6) if mode else ""))
print("frame   ground truth   frame-to-frame (error)   frame-to-model (error)   model coverage")
for (k, a, _), (_, b, cov) in zip(rows[False], rows[True]):
    gt = 0.1 * k
    print("%5d   %8.3f deg   %8.3f (%+.3f)        %8.3f (%+.3f)        %.2f" % (k, gt, a, a - gt, b, b - gt, cov))
