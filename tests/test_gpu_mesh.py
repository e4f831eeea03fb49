"""GPU parity of the mesh path: meshToVoxelGrid and Scene::voxelizeMeshes / the config-1 and config-2 style
pipelines (mesh -> voxel grid -> SVO -> extract -> cone-traced render) through the C ABI vs the CPU oracle."""
import numpy as np
import pytest

import meshgen
from util import describe_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    return svoslam_pkg.load(), torch


def f32(x):
    return np.float32(x)


@pytest.mark.parametrize("shape,log_n", [("cube", 5), ("cube", 8), ("sphere", 7), ("sphere", 10), ("soup", 8), ("soup", 9),
                                         ("soup3", 11), ("soup5", 12)])
def test_mesh_to_voxel_grid_matches_oracle(env, oracle, tmp_path, shape, log_n):
    """(soup3 / soup5, round 5: random triangles at 2^11 / 2^12 cells per axis -- 2 M / 6.8 M voxels: scan lines of thousands of cells,
    chunks cut into several pieces of 32 768 candidate cells, 9-bit digits in the fragment sort)"""
    pkg, torch = env
    writers = {"cube": meshgen.write_cube_obj, "sphere": meshgen.write_sphere_obj, "soup": meshgen.write_soup_obj,
               "soup3": lambda q: meshgen.write_soup_obj(q, n=120, seed=3), "soup5": lambda q: meshgen.write_soup_obj(q, n=60, seed=5)}
    path = writers[shape](tmp_path / (shape + ".obj"))
    tex_path = meshgen.write_bmp(tmp_path / "t.bmp") if shape == "sphere" else None
    mesh, tex = pkg.Mesh(path), (pkg.Texture(tex_path) if tex_path else None)
    omesh, otex = oracle.mesh_load_obj(str(path)), (oracle.load_bmp(str(tex_path)) if tex_path else None)
    ws = pkg.Workspace()
    ce, co, idx, scale = pkg.mesh_to_voxel_grid(ws, mesh, tex, log_n)
    rce, rco, ridx = oracle.mesh_to_voxel_grid(omesh, otex, log_n)
    assert len(ridx) > 100
    assert np.array_equal(idx.astype(np.int64), ridx), describe_mismatch(idx.astype(np.int64), ridx)
    assert np.array_equal(ce.cpu().numpy().view(np.uint32), rce.view(np.uint32))
    assert np.array_equal(co.cpu().numpy().view(np.uint32), rco.view(np.uint32))
    assert scale == float((omesh["bbox1"][0] - omesh["bbox0"][0]) / f32(1 << log_n) / f32(2.0))


def oracle_scene(oracle, omesh, otex, log_n):
    """Scene::voxelizeMeshes(true) composed from oracle calls (scene.cpp:64-85, octree.cpp:293-337)"""
    ce, co, _ = oracle.mesh_to_voxel_grid(omesh, otex, log_n)
    b0, b1 = omesh["bbox0"], omesh["bbox1"]
    scale = b1[0] / f32(1 << log_n)
    center = (b1 + b0) / f32(2.0)
    size = b1[0]
    pool = oracle.Pool()
    pool.insert_voxel_grid(ce, co, log_n, center, float(size))   # depth = ceil(log2(size/scale)) = log_n exactly
    ece, eco = pool.extract(log_n, center, float(size))
    return pool, center, float(size), ece, eco, float(scale)


@pytest.mark.parametrize("shape,log_n,res", [("cube", 5, (256, 256)), ("sphere", 8, (160, 120)), ("sphere", 10, (640, 480))])
def test_scene_voxelize_and_render(env, oracle, tmp_path, shape, log_n, res):
    """config 1 (cube, depth 5, one 256x256 raycast from lookAt((0,0.1,-0.6),(0,0.1,0),(0,1,0))) and a config-2
    style textured mesh at depth 8 / 10, 640x480, 3 views"""
    pkg, torch = env
    path = (meshgen.write_cube_obj if shape == "cube" else meshgen.write_sphere_obj)(tmp_path / "m.obj")
    tex_path = None if shape == "cube" else meshgen.write_bmp(tmp_path / "t.bmp")
    scene = pkg.Scene()
    scene.load_obj(path)
    if tex_path:
        scene.load_bmp(tex_path)
    scene.voxelize_meshes(octree=True, log_n=log_n)
    omesh, otex = oracle.mesh_load_obj(str(path)), (oracle.load_bmp(str(tex_path)) if tex_path else None)
    opool, center, size, ece, eco, scale = oracle_scene(oracle, omesh, otex, log_n)
    svo = scene.svo()
    assert svo["max_depth"] == log_n and svo["num_nodes"] == opool.size
    assert np.array_equal(svo["center"], center) and svo["size"] == size
    gw, cw = scene.pool_words(), opool.words()
    assert np.array_equal(gw, cw), describe_mismatch(gw, cw)
    gce, gco, gscale = scene.voxel_grid()
    assert gscale == scale and gce.shape == ece.shape and gce.shape[0] > 0
    assert np.array_equal(gce.view(np.uint32), ece.view(np.uint32)) and np.array_equal(gco.view(np.uint32), eco.view(np.uint32))
    w, h = res
    if shape == "cube":
        views = [oracle.look_at((0, 0.1, -0.6), (0, 0.1, 0), (0, 1, 0))]
    else:
        c = center.astype(np.float64)
        views = [oracle.look_at(tuple(c + np.array(o)), tuple(c), (0, 1, 0)) for o in ((0.2, 0.4, -3.5), (3.0, 0.1, 0.5), (-1.5, 1.5, 2.0))]
    for view in views:
        for mode in (0, 1):
            img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            pkg.cone_trace_svo(img, 45.0, view, svo["data_ptr"], center, size, mode)
            ref, _, _ = oracle.cone_trace(opool, w, h, 45.0, view, center, size, mode)
            got = img.cpu().numpy()
            assert np.array_equal(got, ref), describe_mismatch(got, ref)
    # a once-voxelized mesh never saturates (Q10): carry-mode image is non-trivial, reference-mode alpha is 255 everywhere
    assert (ref[..., 3] == 255).all()


def test_scene_point_cloud_path(env, oracle):
    """Scene::addPointCloudToOctree: tree created from the first cloud's bbox (resolution 0.01, size = bbox1.x)"""
    pkg, torch = env
    rng = np.random.default_rng(2)
    scene = pkg.Scene()
    opool = None
    for f in range(3):
        pts = (rng.random((20000, 3)) * np.array([2.4, 1.6, 2.0]) + np.array([-1.2, -0.8, 0.4])).astype(np.float32) * f32(0.9 if f else 1.0)
        col = rng.integers(0, 256, (20000, 3), dtype=np.uint8)
        b0, b1 = oracle.point_cloud_bbox(pts)
        if f == 0:
            center, size = (b1 + b0) / f32(2.0), float(b1[0])
            q = f32(size) / f32(0.01)
            depth = int(np.ceil(np.log2(np.float64(q))))  # exact ceil-log2 of the rounded quotient (not a power of two here)
            opool = oracle.Pool()
        scene.add_point_cloud_to_octree((0, 0, 0), torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), b0, b1)
        opool.insert_cloud(pts, col, depth, center, size)
        svo = scene.svo()
        assert svo["max_depth"] == depth and svo["num_nodes"] == opool.size
        assert np.array_equal(scene.pool_words(), opool.words())
    scene.extract_voxel_grid_from_octree()
    gce, gco, gscale = scene.voxel_grid()
    ed = int(np.ceil(np.log2(np.float64(f32(size) / f32(0.01)))))
    rce, rco = opool.extract(ed, center, size)
    assert np.array_equal(gce.view(np.uint32), rce.view(np.uint32)) and np.array_equal(gco.view(np.uint32), rco.view(np.uint32))


def test_config5_style_depth16_4k_bands(env, oracle, tmp_path):
    """config 5's shape (sponza.obj is not in the reference checkout: a procedural colonnade stands in): sparse
    voxelization at 2^16 per axis -> depth-16 SVO -> 3840x2160 cone trace, whole and in 8 bands of 270 rows,
    pool and image against the oracle.  The mesh is a needle-thin variant so that the oracle's 7.8 M voxels /
    8.3 M rays stay within seconds; tools/mesh_bench.py times the 318 M-voxel variant."""
    pkg, torch = env
    log_n, (w, h) = 16, (3840, 2160)
    # (round 6: with a patch of floor platform, as tools/mesh_bench.py's stand-in has it, and a pose over it)
    path = meshgen.write_colonnade_obj(tmp_path / "col.obj", n_cols=1, length=8.0, col_radius=0.0003, col_height=2.0, segs=6,
                                       z_off=1.0, beam=0.0002, floor=(-4.0, -3.9, -0.05, 0.05, 4, 2, 0.148))
    tex_path = meshgen.write_bmp(tmp_path / "t.bmp", 256, 256)
    scene = pkg.Scene()
    scene.load_obj(path)
    scene.load_bmp(tex_path)
    scene.voxelize_meshes(octree=True, log_n=log_n)
    omesh, otex = oracle.mesh_load_obj(str(path)), oracle.load_bmp(str(tex_path))
    opool, center, size, ece, eco, scale = oracle_scene(oracle, omesh, otex, log_n)
    svo = scene.svo()
    assert svo["max_depth"] == log_n and svo["num_nodes"] == opool.size > 1000000
    gw, cw = scene.pool_words(), opool.words()
    assert np.array_equal(gw, cw), describe_mismatch(gw, cw)
    gce, gco, gscale = scene.voxel_grid()
    assert gscale == scale and np.array_equal(gce.view(np.uint32), ece.view(np.uint32)) and np.array_equal(gco.view(np.uint32), eco.view(np.uint32))
    # close to a column foot, so that the cone LOD reaches depth 16 (voxel 0.12 mm, pixel footprint 0.36 mrad)
    target = ece[len(ece) // 3, :3].astype(np.float64)
    views = [oracle.look_at(tuple(target + np.array((0.004, 0.003, -0.012))), tuple(target), (0, 1, 0)),
             # tools/mesh_bench.py's "nave" pose at this mesh's scale: low over the floor platform, along +x, slightly down
             oracle.look_at((-3.99, 0.148 + 0.03, 0.002), (4.0, 0.148 - 0.04, 0.0), (0, 1, 0))]
    lit = []
    for view, mode in ((views[0], 0), (views[0], 1), (views[1], 1)):
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
        pkg.cone_trace_svo(img, 45.0, view, svo["data_ptr"], center, size, mode, counters=cnt)
        ref, steps, levels = oracle.cone_trace(opool, w, h, 45.0, view, center, size, mode)
        got = img.cpu().numpy()
        assert np.array_equal(got, ref), describe_mismatch(got, ref)
        assert [int(x) for x in cnt.cpu()] == [steps, levels]
        band = torch.zeros_like(img)
        for b in range(8):
            pkg.cone_trace_svo_band(band, b * 270, 270, 45.0, view, svo["data_ptr"], center, size, mode)
        assert torch.equal(band, img)
        lit.append(int((ref[..., :3].sum(-1) > 0).sum()))
    assert lit[1] > 1000            # the column is in view (carry mode)
    assert lit[2] > 0.04 * w * h    # the floor platform lights a good part of the image from the pose over it (4.9 % here)


def test_voxel_grid_to_mesh(env, oracle, tmp_path):
    """voxelization::voxelGridToMesh: a cube mesh instanced per voxel of a voxelized mesh (vbo/ibo/nbo/cbo bit-exact)"""
    pkg, torch = env
    path = meshgen.write_sphere_obj(tmp_path / "s.obj")
    tex_path = meshgen.write_bmp(tmp_path / "t.bmp")
    mesh, tex = pkg.Mesh(path), pkg.Texture(tex_path)
    ws = pkg.Workspace()
    ce, co, _, scale = pkg.mesh_to_voxel_grid(ws, mesh, tex, 7, want_indices=False)
    assert ce.shape[0] > 1000
    cube = oracle.mesh_load_obj(str(meshgen.write_cube_obj(tmp_path / "cube.obj", h=1.0)))
    cube_vbo = cube["vbo"].reshape(-1)
    rng = np.random.default_rng(3)
    cube_nbo = rng.standard_normal(cube_vbo.size).astype(np.float32)   # any per-vertex normals: they are copied through
    cube_ibo = np.arange(cube_vbo.size // 3, dtype=np.int32)
    factor = float(np.float32(scale) / np.float32(0.1))                  # computeScale / CUBE_MESH_SCALE
    vbo, ibo, nbo, cbo = pkg.voxel_grid_to_mesh(ws, ce, co, factor, cube_vbo, cube_ibo, cube_nbo)
    rv, ri, rn, rc = oracle.voxel_grid_to_mesh(ce.cpu().numpy(), co.cpu().numpy(), factor, cube_vbo, cube_ibo, cube_nbo)
    assert np.array_equal(vbo.cpu().numpy().view(np.uint32), rv.view(np.uint32))
    assert np.array_equal(ibo.cpu().numpy(), ri)
    assert np.array_equal(nbo.cpu().numpy().view(np.uint32), rn.view(np.uint32))
    assert np.array_equal(cbo.cpu().numpy().view(np.uint32), rc.view(np.uint32))
    # an empty grid is accepted
    e = torch.zeros((0, 4), dtype=torch.float32, device="cuda")
    v0, i0, n0, c0 = pkg.voxel_grid_to_mesh(ws, e, e, factor, cube_vbo, cube_ibo, cube_nbo)
    assert v0.numel() == 0 and i0.numel() == 0
