"""Regenerates tests/golden/cfg2_bunny.json: digests of the CPU ORACLE's outputs for BASELINE config 2 on the
reference's own input data (objs/bunny_tex.obj + textures/texture1.bmp, committed as data fixtures under tests/data/):
voxel list of meshToVoxelGrid at 2^10 per axis, node pool of svoFromVoxelGrid at depth 10 (Scene::voxelizeMeshes
geometry: centre = bbox centre, size = bbox1.x, scene.cpp:64-85), and three 640x480 cone-traced views in both render
modes.  The reference itself cannot be run here (CUDA); these vectors pin the HIP path to the oracle on real data.
Also a CPU test (tests/test_mesh_cpu.py) recomputes part of it so the oracle cannot drift silently.
Run from the repository root:  python tests/golden/make_cfg2_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def compute(with_images=True):
    from oracle import oracle as ora
    from test_gpu_configs import cfg2_views
    obj = os.path.join(ROOT, "tests", "data", "bunny_tex.obj")
    bmp = os.path.join(ROOT, "tests", "data", "texture1.bmp")
    log_n = 10
    mesh, tex = ora.mesh_load_obj(obj), ora.load_bmp(bmp)
    ce, co, idx = ora.mesh_to_voxel_grid(mesh, tex, log_n)
    b0, b1 = mesh["bbox0"], mesh["bbox1"]
    center = (b1 + b0) / np.float32(2.0)
    size = float(b1[0])
    pool = ora.Pool()
    pool.insert_voxel_grid(ce, co, log_n, center, size)
    out = {"n_tris": int(mesh["vbo"].shape[0]),
           "n_voxels": int(len(idx)), "voxel_index_sha256": sha(idx.astype(np.int64)),
           "voxel_centers_sha256": sha(ce), "voxel_colors_sha256": sha(co),
           "bbox0": [float(v) for v in b0], "bbox1": [float(v) for v in b1],
           "center": [float(v) for v in center], "size": size,
           "num_nodes": int(pool.size), "pool_sha256": sha(pool.words())}
    if with_images:
        out["images_sha256"] = []
        out["image_coloured_pixels"] = []
        out["image_steps_levels"] = []
        for eye in cfg2_views(center):
            view = ora.look_at(eye, tuple(np.asarray(center, np.float64)), (0, 1, 0))
            row, cnt, sl = [], [], []
            for mode in (0, 1):
                img, steps, levels = ora.cone_trace(pool, 640, 480, 45.0, view, center, size, mode)
                sl.append([int(steps), int(levels)]); row.append(sha(img)); cnt.append(int((img[..., :3].max(-1) > 0).sum()))
            out["images_sha256"].append(row); out["image_coloured_pixels"].append(cnt); out["image_steps_levels"].append(sl)
    return out


if __name__ == "__main__":
    res = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg2_bunny.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))
