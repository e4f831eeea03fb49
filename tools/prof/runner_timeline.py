import sys, os, numpy as np, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["SVOSLAM_CONFIG"] = "runner_timeline=1"
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
W, H, D, edge = 640, 480, 12, 4.096
K = int(os.environ.get('TIMELINE_FRAMES', '300'))   # the map's age matters: BASELINE config 3 is 300 frames
if os.environ.get("TIMELINE_WORKLOAD") == "cfg4":  # 1920x1080, depth 14, half edge 8.192 m (bench.py --workload cfg4)
    W, H, D, edge = 1920, 1080, 14, 8.192
depth, rgb = synth.render_stream(K, W, H, device="cuda")
views = [pl.ground_truth_view(k, synth) for k in range(K)]
emu = None
if len(sys.argv) > 1:    # "R/N": rank R of an N-rank frame-sharded session, emulated on this GPU (pipeline.EmulatedRank)
    er, en = (int(x) for x in sys.argv[1].split("/"))
    emu = pl.EmulatedRank(er, en)
    dcam = pkg.Camera(W, H, synth.focal_length(W), synth.focal_length(W))
    table = torch.zeros((K, pkg.DELTA_FLOATS), dtype=torch.float32, device="cuda")
    for k in range(1, K):
        dcam.pair_delta(depth[k - 1], rgb[k - 1], depth[k], rgb[k], table[k])
    torch.cuda.synchronize()
    print("emulated rank %d of %d (frame-sharded: maps* / trk0 / pose columns = the pose composition only)" % (er, en))
P = pl.SlamPipeline(W, H, D, (0, 1.5, 0), edge, pool_capacity_nodes=(1 << 30) - 8, dist=emu)
if emu: emu.expect(table[:6], 0, max(1, 16 // emu.world))
P.run_stream(depth[:6], rgb[:6], list(range(6)), views[:6]); torch.cuda.synchronize(); P.reset()
import time
if emu: emu.expect(table, 0, max(1, 16 // emu.world))
t0 = time.perf_counter()
P.run_stream(depth, rgb, list(range(K)), views); torch.cuda.synchronize()
print("ms/frame %.3f" % ((time.perf_counter() - t0) / K * 1e3))
tl = P._runner.timeline()
names = ["maps0", "maps1", "trk0", "pose", "prep0", "plan0", "plan1", "com0", "com1", "ray1"]
print("frame " + " ".join("%8s" % n for n in names))
A, B = K - 40, K - 2
for i in range(K - 20, K - 14):
    print("%5d " % i + " ".join("%8.3f" % (tl[i][k] - tl[K - 20][0]) for k in range(10)))
d = np.diff(tl[A:B, 9]); print("march-end period: mean %.3f ms" % d.mean())
print("durations (mean, the last 38 frames): maps %.3f track %.3f bp+sort %.3f plan %.3f commit %.3f commit->ray end %.3f" % (
    (tl[A:B,1]-tl[A:B,0]).mean(), (tl[A:B,3]-tl[A:B,2]).mean(), (tl[A:B,5]-tl[A:B,4]).mean(), (tl[A:B,6]-tl[A:B,5]).mean(),
    (tl[A:B,8]-tl[A:B,7]).mean(), (tl[A:B,9]-tl[A:B,8]).mean()))
print("waits: pose->prep0 %.3f  plan1->com0 %.3f  ray1(prev)->com0 %.3f" % ((tl[A:B,4]-tl[A:B,3]).mean(), (tl[A:B,7]-tl[A:B,6]).mean(), (tl[A + 1:B,7]-tl[A:B - 1,9]).mean()))
