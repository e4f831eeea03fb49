"""The CPU oracle on seeded inputs against the committed digests (tests/golden/oracle_digests.json, written by
tests/golden/make_oracle_digests.py): pins the oracle itself -- the checker of every GPU parity test -- against
accidental change.  Not a statement about the reference (parity unpinned by the reference)."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_outputs_match_committed_digests():
    spec = importlib.util.spec_from_file_location("make_oracle_digests", os.path.join(HERE, "golden", "make_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "oracle_digests.json")))
    got = mod.compute()
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k


def test_oracle_cfg2_bunny_matches_committed_digests():
    """BASELINE config 2 on the reference's own bunny_tex.obj + texture1.bmp (tests/data/): voxel list and node pool of
    the oracle against tests/golden/cfg2_bunny.json (the images are checked on the GPU side, tests/test_gpu_configs.py);
    the product's host OBJ / BMP loaders agree with the oracle's on the same files"""
    import numpy as np
    spec = importlib.util.spec_from_file_location("make_cfg2_golden", os.path.join(HERE, "golden", "make_cfg2_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "cfg2_bunny.json")))
    got = mod.compute(with_images=False)
    for k in got:
        assert got[k] == want[k], k
    import svoslam_pkg
    from oracle import oracle as ora
    pkg = svoslam_pkg.load()
    obj, bmp = os.path.join(HERE, "data", "bunny_tex.obj"), os.path.join(HERE, "data", "texture1.bmp")
    m, r = pkg.Mesh(obj), ora.mesh_load_obj(obj)
    assert m.n_tris == 4968 and np.array_equal(m.vbo().view(np.uint32), r["vbo"].view(np.uint32))
    assert np.array_equal(m.tbo().view(np.uint32), r["tbo"].view(np.uint32))
    t, rt = pkg.Texture(bmp).data(), ora.load_bmp(bmp)
    rt_arr = rt["data"] if isinstance(rt, dict) else rt
    assert np.array_equal(np.asarray(t, np.float32).reshape(-1), np.asarray(rt_arr, np.float32).reshape(-1))
