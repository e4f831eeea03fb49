"""Regenerates tests/golden/fullsize_digests.json: SHA-256 digests of the CPU oracle's state while it runs BASELINE
config 3 (640x480, depth 12, 300 frames) and config 4 (1920x1080, depth 14, 16 frames) at FULL size, at checkpoints.

Why digests: the oracle needs ~1 s (cfg3) / ~6 s (cfg4) per frame on one core, too slow to repeat on the GPU box inside
the parity suite, and the 300-frame pool is 2.3 GB.  This script runs it ONCE in the build container; the `-m gpu` test
tests/test_gpu_fullsize.py replays the same frames through the HIP path and compares pool words, pose, image and the
march counters with the digests.  The digests pin HIP == oracle at the sizes the bench runs; they
say nothing about the reference (parity unpinned by the reference).

The input frames come from octree_slam_amd.synth on the CPU (torch CPU generators are fixed by the seeds); a digest of
the inputs is stored with every checkpoint so that a differing generator shows up as such, not as a parity failure.

  python tests/golden/make_fullsize_digests.py [cfg3|cfg4] ...      (from the repository root; ~8 minutes)
"""
import hashlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")

CENTER = (0.0, 1.5, 0.0)
CONFIGS = {
    # name: width, height, depth, half-edge, checkpoints (frames fused when the state is digested), render mode, strict tracker
    "cfg3": (640, 480, 12, 4.096, (4, 24, 72, 150, 300), 0, True),
    "cfg4": (1920, 1080, 14, 8.192, (2, 8, 16), 0, True),
    # the same stream with this build's CORRECTED tracker (own specification: oracle ora_camera_set_strict_reference(c, 0)): the
    # pose follows the sensor, surfaces are re-observed, alpha saturates, rays retire, the reference-mode image has colour
    "cfg3_corrected": (640, 480, 12, 4.096, (4, 24, 72, 150, 300), 0, False),
}


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def sha_pool(pool):
    """digest of the oracle's pool words in place (2.3 GB at the end of cfg3: no copy)"""
    import ctypes as C
    n = 2 * pool.size
    buf = (C.c_uint32 * n).from_address(C.addressof(pool._p.data.contents))
    h = hashlib.sha256()
    mv = memoryview(buf).cast("B")
    step = 1 << 28
    for o in range(0, len(mv), step):
        h.update(mv[o:o + step])
    return h.hexdigest()


def run_stream_config(name):
    import svoslam_pkg
    svoslam_pkg.load()
    from oracle import oracle as ora
    synth = importlib.import_module("octree_slam_amd.synth")
    pl = importlib.import_module("octree_slam_amd.pipeline")
    w, h, depth, edge, checkpoints, mode, strict = CONFIGS[name]
    L = ora.lib(native=True)            # -O3 -march=native, -ffp-contract=off kept: same bits as the -O2 build
    f = synth.focal_length(w)
    cam, pool = ora.Camera(w, h, f, f, L=L), ora.Pool(L=L)
    if not strict:
        cam.set_strict_reference(False)
    out = {"width": w, "height": h, "depth": depth, "edge": edge, "center": list(CENTER), "render_mode": mode, "strict_reference": bool(strict),
           "checkpoints": {}}
    tot_steps = tot_levels = 0
    hin = hashlib.sha256()
    t0 = time.time()
    for k in range(max(checkpoints)):
        d, c = synth.render_frame(k, w, h)
        dn, cn = d.numpy().view(np.uint16), c.numpy()
        hin.update(dn.tobytes()); hin.update(cn.tobytes())
        view = pl.ground_truth_view(k, synth)
        cam.update(dn, cn, k)
        v = ora.vertex_map(dn, f, f, w, h)
        v = ora.transform_vertex_map(v, cam.fusion_transform())
        pool.insert_cloud(v.reshape(-1, 3), cn.reshape(-1, 3), depth, CENTER, edge)
        img, steps, levels = ora.cone_trace(pool, w, h, 45.0, view, CENTER, edge, mode, L=L)
        tot_steps += steps; tot_levels += levels
        if k + 1 in checkpoints:
            p, o = cam.pose()
            img1, _, _ = ora.cone_trace(pool, w, h, 45.0, view, CENTER, edge, 1, L=L)     # carry mode: a coloured image
            out["checkpoints"][str(k + 1)] = {
                "inputs_sha256": hin.copy().hexdigest(),
                "pool_nodes": int(pool.size), "pool_sha256": sha_pool(pool),
                "pose_sha256": sha(p.view(np.uint32), o.view(np.uint32)),
                "image_sha256": sha(img), "image_coloured_pixels": int((img[..., :3].max(-1) > 0).sum()),
                "image_carry_sha256": sha(img1), "image_carry_coloured_pixels": int((img1[..., :3].max(-1) > 0).sum()),
                "steps_total": int(tot_steps), "levels_total": int(tot_levels),
                "tracking_lost_levels": int(cam.tracking_lost_count()),
                "saturated_nodes": int(((pool.words()[1::2] >> 24) >= 254).sum()) if pool.size < (1 << 27) else None,
            }
            print(name, k + 1, "%.0f s" % (time.time() - t0), out["checkpoints"][str(k + 1)], flush=True)
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4"]
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in which:
        res[name] = run_stream_config(name)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)
