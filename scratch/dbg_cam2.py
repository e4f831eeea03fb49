import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svoslam_pkg
pkg = svoslam_pkg.load()
import importlib
synth = importlib.import_module("octree_slam_amd.synth")
w, h = 160, 120
f = synth.focal_length(w)
for variant in ("update", "update_sync_each", "stepping"):
    cam = pkg.Camera(w, h, f, f)
    for k in range(3):
        d, c = synth.render_frame(3 * k, w, h)
        dd, cc = d.cuda(), c.cuda()
        if variant == "update":
            cam.update(dd, cc, k)
        else:
            cam.begin(dd, cc, k)
            for level in (2, 1, 0):
                for it in range(pkg.PYRAMID_ITERS[level]):
                    cam.icp_accumulate(level, it)
                    if variant == "update_sync_each": torch.cuda.synchronize()
                    cam.icp_solve(level, it)
            cam.end()
        p, o = cam.pose()
        A, b, x = cam.last_system()
        print(variant, k, "lost", cam.tracking_lost_count(), "o", o[:4], "x", x, "A00", A[0, 0])
