import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svoslam_pkg
pkg = svoslam_pkg.load()
import importlib
synth = importlib.import_module("octree_slam_amd.synth")
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
K = 12
depth, rgb = synth.render_stream(K, W, H, device="cuda")
f = synth.focal_length(W)
cam = pkg.Camera(W, H, f, f)
for i in range(K): cam.update(depth[i], rgb[i], i)
torch.cuda.synchronize()
p = cam.track_profile().astype(np.int64)
names = ["s.start", "s.fanin", "s.rows", "s.publish", "w.start", "w.terms", "w.stored", "w.bcast"]
t0 = p[1][4]
print("clock ticks relative to worker0 epoch-1 start; columns:", names)
for e in range(1, 20):
    print("e%2d " % e + " ".join("%8d" % (p[e][k] - t0) for k in range(8)))
tot = p[19][3] - p[1][4]
print("total ticks", tot)
