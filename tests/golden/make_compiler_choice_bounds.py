"""Writes profiles/r05_compiler_choice_bounds.json: the measurements of tests/test_oracle_compiler_choices.py at the sizes of the
BASELINE configurations -- cfg3's first 24 frames at 640x480 / depth 12 and cfg2 (bunny_tex.obj at 2^10 cells per
axis), parity oracle against its two sensitivity builds (oracle/Makefile `variants`: fmad = every product-sum a contraction candidate,
satu8 = the other reading of float -> uint8_t and of max(0, unsigned - 127)).  CPU only, ~3 minutes:

    python tests/golden/make_compiler_choice_bounds.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "octree-slam_amd"))

import synth  # noqa: E402
import test_oracle_compiler_choices as T  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    libs = {"base": oracle.lib(), "fmad": oracle.lib(variant="fmad"), "satu8": oracle.lib(variant="satu8")}
    w, h, n = 640, 480, 24
    f = synth.focal_length(w)
    frames = T.frames_of(synth, n, w, h)
    out = {"what": "parity oracle vs its sensitivity builds on identical inputs; see tests/test_oracle_compiler_choices.py",
           "sizes": {"cfg3": "%dx%d, %d frames, depth 12" % (w, h, n), "cfg2": "bunny_tex.obj, 2^10 cells per axis"}}
    r0, m0 = T.sensor_stage_deviation(oracle, libs, frames[0][0], w, h, f)
    r1, m1 = T.sensor_stage_deviation(oracle, libs, frames[1][0], w, h, f)
    out["sensor_frame0"], out["sensor_frame1"] = r0, r1
    out["icp_cost2_frames_0_1"] = T.icp_deviation(oracle, libs, m0, m1)
    out["tracker_24_frames"] = T.tracker_deviation(oracle, libs, frames, w, h, f)
    out["whole_frames_24"] = T.slam_structure_deviation(oracle, libs, frames, w, h, f, 12, (0.0, 1.5, 0.0), 4.096)
    rng = np.random.default_rng(11)
    pool = T.lit_pool(oracle, libs["base"], rng, n=40000, depth=9)
    view = oracle.look_at((0.1, 0.2, -2.2), (0, 0, 0), (0, 1, 0))
    for mode, name in ((oracle.RENDER_REFERENCE, "reference"), (oracle.RENDER_CARRY, "carry")):
        dev, _ = T.render_deviation(oracle, libs, pool.words(), 640, 480, view, mode)
        out["render_640x480_lit_pool_%s_mode" % name] = dev
    obj = os.path.join(ROOT, "tests", "data", "bunny_tex.obj")
    sets = {}
    for nm in ("base", "fmad"):
        with oracle.using(libs[nm]):
            mesh = oracle.mesh_load_obj(obj)
            _, _, idx = oracle.mesh_to_voxel_grid(mesh, None, 10)
        sets[nm] = set(int(x) for x in idx)
    inter, union = len(sets["base"] & sets["fmad"]), len(sets["base"] | sets["fmad"])
    out["cfg2_voxel_set"] = {"voxels_base": len(sets["base"]), "voxels_fmad": len(sets["fmad"]), "jaccard": inter / union}
    path = os.path.join(ROOT, "profiles", "r05_compiler_choice_bounds.json")
    with open(path, "w") as fp:
        json.dump(out, fp, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
