"""Regenerates tests/golden/oracle_digests.json: SHA-256 digests of CPU-oracle outputs on seeded inputs.
They pin the ORACLE (test infrastructure) against accidental change; they say nothing about the reference,
for which no vectors exist.  Run from the repository root:  python tests/golden/make_oracle_digests.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def compute():
    from oracle import oracle as ora
    from util import surface_cloud
    out = {}
    rng = np.random.default_rng(20240901)
    pts, col = surface_cloud(rng, 6000)
    pool = ora.Pool()
    for k in range(2):
        pool.insert_cloud(pts + np.float32(0.01 * k), col, 8, (0, 0, 0), 1.0)
    out["fusion_pool_depth8_2frames"] = {"size": int(pool.size), "sha256": digest(pool.words())}
    view = ora.look_at((0.1, 0.2, -2.6), (0, 0, 0), (0, 1, 0))
    for mode in (0, 1):
        img, steps, levels = ora.cone_trace(pool.words(), 64, 48, 45.0, view, (0, 0, 0), 1.0, mode)
        out["cone_trace_64x48_mode%d" % mode] = {"steps": int(steps), "levels": int(levels), "sha256": digest(img)}
    if hasattr(pool, "extract"):
        c, cc = pool.extract(8, (0, 0, 0), 1.0)
        out["extract_depth8"] = {"n": int(len(c)), "sha256": digest(c, cc)}
    h, w = 60, 80
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (1200 + 9 * xx + 4 * yy + rng.integers(0, 30, (h, w))).astype(np.uint16)
    depth[rng.random((h, w)) < 0.03] = 0
    out["bilateral_60x80"] = {"sha256": digest(ora.bilateral(depth))}
    f = 570.3 * w / 640.0
    v1 = ora.vertex_map(depth, f, f, w, h); n1 = ora.normal_map(v1)
    T = ora.icp_update_transform(np.array([0.004, -0.003, 0.002, 0.004, -0.002, 0.003], np.float32))
    v2 = ora.transform_vertex_map(v1, T); n2 = ora.transform_normal_map(n1, T)
    A, b = ora.icp_cost2(v1, n1, v2, n2)
    out["icp_cost2_60x80"] = {"sha256": digest(A, b)}
    A, b, m = ora.icp_cost(v1, n1, v2, n2)
    out["icp_cost_60x80"] = {"correspondences": int(m), "sha256": digest(A, b)}
    # photometric RGB-D term (own specification)
    i1 = (0.5 + 0.4 * np.sin(xx / 5.0) * np.cos(yy / 7.0)).astype(np.float32)
    i2 = (0.5 + 0.4 * np.sin((xx + 0.7) / 5.0) * np.cos(yy / 7.0)).astype(np.float32)
    g1 = ora.gradient(i1)
    out["gradient_60x80"] = {"sha256": digest(g1)}
    A, b = ora.rgbd_cost(i1, g1, v1, i2, v2, f, f, w, h)
    out["rgbd_cost_60x80"] = {"sha256": digest(A, b)}
    import importlib
    import svoslam_pkg
    svoslam_pkg.load()
    synth = importlib.import_module("octree_slam_amd.synth")
    cam = ora.Camera(160, 120, 142.575, 142.575)
    cam.set_rgbd(True)
    for k in range(4):
        d, c = synth.render_frame(k, 160, 120)
        cam.update(d.numpy().view(np.uint16), c.numpy(), k)
    out["camera_rgbd_160x120_4frames"] = {"sha256": digest(*cam.pose(), *cam.last_system())}
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "oracle_digests.json")
    json.dump(compute(), open(path, "w"), indent=1, sort_keys=True)
    print(open(path).read())
