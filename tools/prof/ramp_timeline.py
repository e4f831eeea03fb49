"""the driver's command in slow motion: 275 frames fused untimed, 5 warm-up frames, a barrier, then 20 frames -- the scheduler's
HIP-event marks of those 20 (ms from the first mark), to see what the cold start of the timed region costs"""
import sys, os, numpy as np, torch, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SVOSLAM_CONFIG"] = "runner_timeline=1"
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, edge = 640, 480, 12, 4.096
K = 300
depth, rgb = synth.render_stream(K, W, H, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(K)]
P = pl.SlamPipeline(W, H, D, (0, 1.5, 0), edge, pool_capacity_nodes=(1 << 30) - 8, count_steps=True)
P.run_stream(depth[:6], rgb[:6], list(range(6)), views[:6]); torch.cuda.synchronize(); P.reset()
P.run_stream(depth[:275], rgb[:275], list(range(275)), views[:275]); torch.cuda.synchronize()
P.run_stream(depth[275:280], rgb[275:280], list(range(275, 280)), views[275:280]); torch.cuda.synchronize()
t0 = time.perf_counter()
P.run_stream(depth[280:], rgb[280:], list(range(280, 300)), views[280:]); torch.cuda.synchronize()
el = time.perf_counter() - t0
print("20 frames: %.3f ms wall = %.1f frames/s" % (el * 1e3, 20 / el))
tl = P._runner.timeline()
names = ["maps0", "maps1", "trk0", "pose", "prep0", "plan0", "plan1", "com0", "com1", "ray1"]
print("frame " + " ".join("%8s" % n for n in names))
base = tl[0][0]
for i in range(len(tl)):
    print("%5d " % i + " ".join("%8.3f" % (tl[i][k] - base) for k in range(10)))
d = np.diff(tl[:, 9]); print("march-end periods:", " ".join("%.3f" % x for x in d))
