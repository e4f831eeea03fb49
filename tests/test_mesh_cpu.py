"""CPU tests of the mesh path: the product's host-side OBJ/BMP loaders (no GPU involved) against the
oracle's restatement, and oracle sanity against SURVEY.md Appendix D.8."""
import os

import numpy as np
import pytest

import meshgen

REF_OBJS = "/root/reference/objs"


def load_pkg():
    import svoslam_pkg
    return svoslam_pkg.load()


def check_same_mesh(pkg, oracle, path):
    m, r = pkg.Mesh(path), oracle.mesh_load_obj(str(path))
    assert m.n_tris == r["vbo"].shape[0] > 0
    assert np.array_equal(m.vbo().view(np.uint32), r["vbo"].view(np.uint32))
    if r["tbo"] is None:
        assert m.tbo() is None
    else:
        assert np.array_equal(m.tbo().view(np.uint32), r["tbo"].view(np.uint32))
    b0, b1 = m.bbox()
    assert np.array_equal(b0, r["bbox0"]) and np.array_equal(b1, r["bbox1"])
    return m, r


def test_obj_loaders_agree(tmp_path, oracle):
    pkg = load_pkg()
    m, r = check_same_mesh(pkg, oracle, meshgen.write_cube_obj(tmp_path / "cube.obj"))
    assert m.n_tris == 12
    # obj::recenter: x/z centred, y min = 0  (bbox0=(-0.1,0,-0.1), bbox1=(0.1,0.2,0.1))
    np.testing.assert_allclose(r["bbox0"], [-0.1, 0.0, -0.1], atol=1e-7)
    np.testing.assert_allclose(r["bbox1"], [0.1, 0.2, 0.1], atol=1e-7)
    m, r = check_same_mesh(pkg, oracle, meshgen.write_sphere_obj(tmp_path / "sphere.obj"))
    assert r["tbo"].shape == r["vbo"].shape[:2] + (2,)
    check_same_mesh(pkg, oracle, meshgen.write_soup_obj(tmp_path / "soup.obj"))


@pytest.mark.skipif(not os.path.isdir(REF_OBJS), reason="reference data files only exist in the build container")
def test_obj_loaders_on_reference_data(oracle):
    pkg = load_pkg()
    # (objs/pyramid.obj is excluded: its double-space separated, 0-based faces make the reference's loader
    #  index points[-1]: undefined behaviour, nothing to agree on)
    for name in ("cube.obj", "bunny_tex.obj", "suzanne.obj"):
        p = os.path.join(REF_OBJS, name)
        if os.path.exists(p):
            check_same_mesh(pkg, oracle, p)


def test_bmp_loaders_agree(tmp_path, oracle):
    pkg = load_pkg()
    p = meshgen.write_bmp(tmp_path / "t.bmp", 64, 32)
    t = pkg.Texture(p).data()
    r = oracle.load_bmp(str(p))
    assert t.shape == r.shape == (32, 64, 3) and np.array_equal(t, r)
    assert 0.0 <= r.min() and r.max() <= 1.0


def test_oracle_cube_voxelization_appendix_d8(tmp_path, oracle):
    """geometry on the max faces of the mesh's own AABB falls on w = N and is dropped; min faces are kept"""
    m = oracle.mesh_load_obj(str(meshgen.write_cube_obj(tmp_path / "cube.obj")))
    ce, co, idx = oracle.mesh_to_voxel_grid(m, None, 5)
    T, M = 8, 4
    tile, pix = idx // 512, idx % 512
    x = (tile % M) * 8 + pix % 8; y = (tile // M % M) * 8 + pix // 8 % 8; z = (tile // (M * M)) * 8 + pix // 64
    for c in (x, y, z):
        assert (c == 0).sum() == 32 * 32          # the three min faces are complete
        assert (c == 31).sum() < 100              # max faces: only the conservative fringe
    assert np.array_equal(np.unique(co, axis=0), [[0.0, 1.0, 0.0, 0.0]])   # no texture -> green (voxelization.cu:101-103)
    assert np.array_equal(idx, np.sort(idx)) and np.unique(idx).size == idx.size


def test_cfg1_cube_obj_oracle_matches_committed_digests():
    """BASELINE config 1 (the CPU-runnable case) on the reference's own objs/cube.obj (tests/data/cube.obj): 12 triangles,
    recentred box y in [0, 0.2], 2^5 voxelization, depth-5 SVO, one 256x256 render by the oracle's host walk -- against
    tests/golden/cfg1_cube.json (the GPU side: tests/test_gpu_configs.py::test_cfg1_cube_obj)."""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_cfg1_golden", os.path.join(here, "golden", "make_cfg1_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "cfg1_cube.json")))
    got = mod.compute()
    assert got == want
    assert got["n_tris"] == 12 and got["bbox0"][1] == 0.0 and abs(got["bbox1"][1] - 0.2) < 1e-6
    assert got["size"] == got["bbox1"][0] and 129 in got["alpha_values"]                              # A = 127 + 2 after one insert
