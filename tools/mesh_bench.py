"""Stage timings and rooflines of the mesh configurations of BASELINE.json (configs 2 and 5; SURVEY.md 8d inputs 2 and 5):
OBJ + BMP -> sparse voxelization -> SVO -> cone-traced renders, on one MI355X.

    python tools/mesh_bench.py --config cfg2   # the reference's own objs/bunny_tex.obj + textures/texture1.bmp (data fixtures under
                                               # tests/data), depth 10, 640x480, 3 views
    python tools/mesh_bench.py --config cfg5   # procedural colonnade (sponza.obj is NOT in the reference checkout), depth 16, 3840x2160,
                                               # whole image and 8 row bands of 270 rows

Prints ONE JSON object; bench.py embeds it in its line as other_configs.cfg2 / .cfg5.  Stage times are HIP-event
brackets on the launch stream (svoslam_stage_timing: mesh_raster / mesh_sort / mesh_emit inside svoslam_mesh_to_voxel_grid, fuse_sort /
fuse_plan / fuse_commit inside svoslam_svo_from_voxel_grid, march around each trace kernel); `call_ms` is the wall clock of the C call.

Algorithmic bytes (T triangles, F (cell, triangle) fragments, V voxels, K split nodes, D depth):
  mesh_raster   36 T (vertices) + 8 F (one packed word per fragment: cell index << tri_bits | triangle id)
  mesh_sort     16 F per pass (8 read + 8 written), ceil(3 D / b) passes of b-bit digits (b = 11 / 9 / 8 by size: radix_sort.hip)
  mesh_emit     12 F read (8-byte key + 4-byte triangle id of the unpacking pass) + 24 T (uv) + 32 V (centre + colour vec4) written
  voxelize call 60 T + 32 V: what meshToVoxelGrid's interface takes in and hands out (the fragments are this build's intermediate)
  svo_from_voxel_grid   SURVEY 8d's fusion formula with 32 bytes per voxel in: 32 V + 4 D V + 8 V + 72 K + 36 K
                        (a fresh pool: every touched inner node is split once, so sum_l U_l = K = (nodes - 8) / 8, U_D = V)
  render        4 (levels + steps) + 4 W H, levels and steps from the kernel's own counters
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """glm::lookAt as the column-major float[16] the C ABI takes"""
    eye, target, up = (np.asarray(v, np.float32) for v in (eye, target, up))
    f = target - eye
    f = f / np.float32(np.sqrt(np.float32(np.dot(f, f))))
    s = np.cross(f, up).astype(np.float32)
    s = s / np.float32(np.sqrt(np.float32(np.dot(s, s))))
    u = np.cross(s, f).astype(np.float32)
    m = np.zeros((4, 4), np.float32)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    m[3, 3] = 1
    return m.T.reshape(16).copy()          # column-major


def roof(alg_bytes, ms):
    gbs = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None
    return {"alg_bytes": alg_bytes, "ms": ms, "achieved_GBs": gbs, "frac": (gbs / HBM_PEAK_GBS) if gbs else None}


def scene_spec(cfg, tmp, meshgen):
    """(obj path, texture path, depth, (W, H), row bands, description) of a mesh configuration"""
    if cfg == "cfg2":
        return (os.path.join(ROOT, "tests", "data", "bunny_tex.obj"), os.path.join(ROOT, "tests", "data", "texture1.bmp"), 10, (640, 480), 1,
                "bunny_tex.obj + texture1.bmp (the reference's own data files, tests/data), depth-10 SVO, 640x480")
    tex_path = meshgen.write_bmp(os.path.join(tmp, "t.bmp"), 256, 256)
    # round 6: round 5's colonnade plus a 10 m x 1.2 m floor quad at its -x end, so that views from INSIDE the model
    # put leaf-level surfaces (cone LOD 15 / 16: within ~1.3 m of the eye at 3840x2160) over a quarter of the image.  (The voxel grid is
    # 2^16 cells along each axis of the mesh's own 40 x 5 x 6 m box: the quad is 1/20 of the box's floor = 215 M voxels of 0.6 x 0.09 mm.)
    obj = meshgen.write_colonnade_obj(os.path.join(tmp, "m.obj"), n_cols=16, length=40.0, col_radius=0.004, col_height=5.0,
                                      segs=16, z_off=3.0, beam=0.002, floor=(-20.0, -10.0, -0.6, 0.6, 80, 10, 0.37))
    return (obj, tex_path, 16, (3840, 2160), 8,
            "STAND-IN for crytek-sponza (sponza.obj is not in the reference checkout): procedural colonnade + a floor quad (tests/meshgen.py), "
            "2^16 cells per axis, depth-16 SVO, 3840x2160, whole image + 8 row bands")


def scene_views(cfg, center, size):
    """[(name, column-major view matrix)]: the fixed poses of a configuration's renders"""
    c = np.asarray(center, np.float64)
    if cfg == "cfg2":   # around the mesh; the last one close enough for the cone LOD to reach the leaves
        return [(n, look_at(c + np.array(o) * size, c)) for n, o in (("front, outside the root cube", (0.15, 0.3, -2.6)),
                                                                      ("side, outside the root cube", (2.2, 0.1, 0.4)),
                                                                      ("close, inside", (-0.3, 0.25, 0.45)))]
    # inside the colonnade (x along the nave; the floor is a platform at y = 0.37 for x in [-20, -10], |z| <= 0.6 -- NOT at y = 0: the
    # root cube's centre is at y = 2.501, so a floor at 0 lies 1 mm under the boundary of a 2.5 m level-4 cell, every ray above it takes
    # 1.25 m steps and jumps through it (the reference's step is half the edge of the EMPTY cell the sample is in); columns at z = +-3)
    return [("nave: 0.3 m over the floor, along +x", look_at((-18.0, 0.67, 0.02), (20.0, -0.03, 0.0))),
            ("back along -x, 0.25 m over the floor", look_at((-12.0, 0.62, 0.3), (-20.0, 0.2, -0.3))),
            ("grazing along the z = +3 row of columns", look_at((-19.6, 1.5, 2.9), (20.0, 1.6, 3.02)))]


def build_scene(cfg, pkg):
    """voxelize + build the SVO of a configuration once: (pool, centre, half-edge, depth, (W, H), views) -- tools/prof/mesh_ray_anatomy.py"""
    import meshgen
    obj, tex_path, depth, res, _, _ = scene_spec(cfg, tempfile.mkdtemp(), meshgen)
    mesh, tex = pkg.Mesh(obj), pkg.Texture(tex_path)
    b0, b1 = mesh.bbox()
    center, size = (b1 + b0) / np.float32(2.0), float(b1[0])
    ws, pool = pkg.Workspace(), pkg.Pool()
    ce, co, _, _ = pkg.mesh_to_voxel_grid(ws, mesh, tex, depth, want_indices=False)
    pkg.svo_from_voxel_grid(ws, ce, co, depth, pool, center, size)
    del ce, co
    build_scene.keep = (ws, mesh, tex)
    return pool, center, size, depth, res, scene_views(cfg, center, size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg5"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import meshgen
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    torch.cuda.set_device(0)
    tmp = tempfile.mkdtemp()
    obj, tex_path, depth, (W, H), bands, what = scene_spec(args.config, tmp, meshgen)
    mesh, tex = pkg.Mesh(obj), pkg.Texture(tex_path)
    b0, b1 = mesh.bbox()
    center, size = (b1 + b0) / np.float32(2.0), float(b1[0])     # Scene::voxelizeMeshes, scene.cpp:72-78
    ws = pkg.Workspace()
    T = int(mesh.n_tris)
    out = {"config": args.config, "workload": what, "triangles": T, "depth": depth, "render": [W, H]}
    mesh_stages = (("mesh_raster", pkg.STAGE_MESH_RASTER), ("mesh_sort", pkg.STAGE_MESH_SORT), ("mesh_emit", pkg.STAGE_MESH_EMIT))
    fuse_stages = (("fuse_sort", pkg.STAGE_FUSE_SORT), ("fuse_plan", pkg.STAGE_FUSE_PLAN), ("fuse_commit", pkg.STAGE_FUSE_COMMIT))
    best, best_stage = {}, {}
    pool = None
    for rep in range(args.reps):
        if pool is not None:
            del pool
        pool = pkg.Pool()
        torch.cuda.synchronize()
        pkg.stage_timing([s for _, s in mesh_stages + fuse_stages])
        t0 = time.perf_counter()
        ce, co, _, scale = pkg.mesh_to_voxel_grid(ws, mesh, tex, depth, want_indices=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pkg.svo_from_voxel_grid(ws, ce, co, depth, pool, center, size)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for nm, st in mesh_stages + fuse_stages:
            ms, n = pkg.stage_timing_read(st)
            if n:
                best_stage[nm] = min(best_stage.get(nm, 1e9), ms)
        pkg.stage_timing([])
        for k, v in (("voxelize_call_ms", t1 - t0), ("svo_from_voxel_grid_call_ms", t2 - t1)):
            best[k] = min(best.get(k, 1e9), v * 1e3)
        V, nodes = int(ce.shape[0]), int(pool.size)
        if rep + 1 < args.reps:
            del ce, co
    out["voxels"], out["nodes"] = V, nodes
    K = (nodes - 8) // 8
    F = pkg.mesh_last_fragments(ws)
    out["fragments"] = F
    digit = 11 if F <= (4 << 20) else (9 if F <= (64 << 20) else 8)     # radix_packed_digit_bits_for
    passes = (3 * depth + digit - 1) // digit
    vox_ms = sum(best_stage.get(nm, 0.0) for nm, _ in mesh_stages)
    fus_ms = sum(best_stage.get(nm, 0.0) for nm, _ in fuse_stages)
    st = {}
    if F is not None:
        st["mesh_raster"] = dict(kernel="tri_scanline_count_kernel + scanline_kernel<false> + scanline_kernel<true> + 2 scans (2 count readbacks inside)",
                                 **roof(36.0 * T + 8.0 * F, best_stage.get("mesh_raster", 0.0)))
        st["mesh_sort"] = dict(kernel="radix_sort_packed_ex: %d passes of <= %d bits x (packed_upsweep, packed_column_scan, packed_downsweep)" % (passes, digit),
                               **roof(16.0 * F * passes, best_stage.get("mesh_sort", 0.0)))
        st["mesh_emit"] = dict(kernel="voxel_flag_kernel + scan + voxel_emit_kernel (1 count readback inside)",
                               **roof(12.0 * F + 24.0 * T + 32.0 * V, best_stage.get("mesh_emit", 0.0)))
    st["voxelize"] = dict(kernel="svoslam_mesh_to_voxel_grid (sum of the three stages above)", **roof(60.0 * T + 32.0 * V, vox_ms))
    fuse_alg = 32.0 * V + 4.0 * depth * V + 8.0 * V + 72.0 * K + 36.0 * K
    st["svo_from_voxel_grid"] = dict(kernel="compute_keys + packed key-only sort (radix_sort_packed_ex) + plan (3) + split_pass per level + fill_kernel + mip_level per level",
                                     parts_ms={nm: best_stage.get(nm) for nm, _ in fuse_stages}, **roof(fuse_alg, fus_ms))
    out["stages"] = st
    out["call_ms"] = {k: round(v, 3) for k, v in best.items()}
    out["call_ms"]["note"] = ("wall clock of the Python call: includes the binding's device-to-device copy of the voxel grid into torch tensors "
                              "(32 B per voxel) and allocation; the stage times above are HIP-event brackets inside the C call")
    views = scene_views(args.config, center, size)
    img = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    renders = []
    for vi, (view_name, view) in enumerate(views):
        for mode, name in ((pkg.RENDER_REFERENCE, "reference"), (pkg.RENDER_CARRY, "carry")):
            cnt.zero_()
            pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, center, size, mode, counters=cnt)      # warm (accel tables) + counters
            torch.cuda.synchronize()
            steps, levels = (int(x) for x in cnt.cpu().tolist())
            pkg.cone_trace_timing(True)
            for _ in range(5):
                pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, center, size, mode)
            ms, n = pkg.cone_trace_timing_read()
            kms = ms / n
            alg = 4.0 * (levels + steps) + 4.0 * W * H
            rec = {"view": vi, "pose": view_name, "mode": name, "trace_kernel_ms": round(kms, 4), "Mrays_per_s": round(W * H / kms / 1e3, 1),
                   "steps": steps, "levels": levels, "alg_bytes": alg, "achieved_GBs": alg / (kms * 1e-3) / 1e9,
                   "frac": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "lit_pixels": int((img[..., :3].sum(-1) > 0).sum().item())}
            if bands > 1:
                full = img.clone()
                img.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for b in range(bands):
                    pkg.cone_trace_svo_band(img, b * (H // bands), H // bands, 45.0, view, pool.data_ptr, center, size, mode)
                torch.cuda.synchronize()
                rec["bands"] = bands
                rec["bands_ms_total_one_gpu"] = round((time.perf_counter() - t0) * 1e3, 4)
                rec["bands_equal_full"] = bool(torch.equal(img, full))
            renders.append(rec)
    pkg.cone_trace_timing(False)
    out["renders"] = renders
    out["render_kernel"] = ("cone_trace_brick_kernel (reference mode over occupancy bricks)" if depth <= 14 else
                            "cone_trace_kernel (tree march: no brick shape for pools deeper than 14)") + " / cone_trace_kernel<CARRY> (carry mode)"
    out["mrays_per_s_min_max"] = [min(r["Mrays_per_s"] for r in renders), max(r["Mrays_per_s"] for r in renders)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
