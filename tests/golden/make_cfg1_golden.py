"""Regenerates tests/golden/cfg1_cube.json: BASELINE config 1 on the reference's OWN objs/cube.obj (committed as a data
fixture, tests/data/cube.obj, like the bunny of config 2): voxelize at 2^5 per axis (no texture: the default green of
voxelization.cu:101-103), depth-5 SVO with Scene::voxelizeMeshes' geometry (centre = bbox centre, size = bbox1.x,
scene.cpp:64-85), one 256x256 render from lookAt((0, 0.1, -0.6), (0, 0.1, 0), (0, 1, 0)), fov 45 by the CPU
oracle's host walk of the node pool -- config 1 is the reference-style CPU-runnable case; the `-m gpu` test then requires
the HIP path to reproduce the same digests.  Run from the repository root:  python tests/golden/make_cfg1_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

EYE, AT, UP, FOV, W, H, LOG_N = (0.0, 0.1, -0.6), (0.0, 0.1, 0.0), (0.0, 1.0, 0.0), 45.0, 256, 256, 5


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def compute():
    from oracle import oracle as ora
    obj = os.path.join(ROOT, "tests", "data", "cube.obj")
    mesh = ora.mesh_load_obj(obj)
    ce, co, idx = ora.mesh_to_voxel_grid(mesh, None, LOG_N)
    b0, b1 = mesh["bbox0"], mesh["bbox1"]
    center = (b1 + b0) / np.float32(2.0)
    size = float(b1[0])
    pool = ora.Pool()
    pool.insert_voxel_grid(ce, co, LOG_N, center, size)
    words = pool.words()
    view = ora.look_at(EYE, AT, UP)
    out = {"n_tris": int(mesh["vbo"].shape[0]), "bbox0": [float(v) for v in b0], "bbox1": [float(v) for v in b1],
           "center": [float(v) for v in center], "size": size,
           "n_voxels": int(len(idx)), "voxel_index_sha256": sha(idx.astype(np.int64)), "voxel_centers_sha256": sha(ce),
           "voxel_colors_sha256": sha(co), "num_nodes": int(pool.size), "pool_sha256": sha(words),
           "alpha_values": sorted(set(int(w >> 24) for w in words[1::2])),
           "images_sha256": [], "image_steps_levels": [], "image_coloured_pixels": []}
    for mode in (0, 1):
        img, steps, levels = ora.cone_trace(pool, W, H, FOV, view, center, size, mode)
        out["images_sha256"].append(sha(img)); out["image_steps_levels"].append([int(steps), int(levels)])
        out["image_coloured_pixels"].append(int((img[..., :3].max(-1) > 0).sum()))
    return out


if __name__ == "__main__":
    res = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg1_cube.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))
