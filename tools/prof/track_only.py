"""RGBDCamera::update over N frames of the cfg3 stream, nothing else (PMC passes of the tracker kernels):  python tools/prof/track_only.py [N] [cfg4]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h = (1920, 1080) if "cfg4" in sys.argv else (640, 480)
f = synth.focal_length(w)
depth, rgb = synth.render_stream(n, w, h, device="cuda")
cam = pkg.Camera(w, h, f, f)
for k in range(n):
    cam.update(depth[k], rgb[k], k)
torch.cuda.synchronize()
print("tracked", n, "frames; lost levels", cam.tracking_lost_count(), "pose", cam.pose()[0])
