"""Pins the OBJ loading half of the mesh path (SURVEY.md 8a row a12's input side) against the REFERENCE's
own loader: tests/golden/ref_obj_loader.json holds digests of what the reference's objUtil code (compiled
from /root/reference by `make -C oracle ref`, run by tests/golden/make_ref_obj_golden.py) produces; the
oracle's restatement (ora_mesh_load_obj) and the product's host loader (svoslam_mesh_load_obj) must
reproduce them bit for bit."""
import json
import os
import sys

import numpy as np
import pytest

import meshgen

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_ref_obj_golden import digest  # noqa: E402

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_obj_loader.json")))
REF_OBJS = "/root/reference/objs"


def product_mesh(path):
    import svoslam_pkg
    m = svoslam_pkg.load().Mesh(str(path))
    b0, b1 = m.bbox()
    return {"vbo": m.vbo(), "tbo": m.tbo(), "bbox0": b0, "bbox1": b1}


def strip(rec):
    return {k: v for k, v in rec.items() if k != "input_sha256"}


def test_generated_objs_match_reference_loader(tmp_path, oracle):
    paths = meshgen.write_generated_objs(str(tmp_path))
    assert sorted(paths) == sorted(GOLDEN["generated"])
    for name, path in sorted(paths.items()):
        want = strip(GOLDEN["generated"][name])
        assert digest(oracle.mesh_load_obj(path)) == want, "oracle loader differs from the reference on " + name
        assert digest(product_mesh(path)) == want, "product loader differs from the reference on " + name


@pytest.mark.skipif(not os.path.isdir(REF_OBJS), reason="reference assets only exist in the build container")
def test_reference_objs_match_reference_loader(oracle):
    for name, rec in sorted(GOLDEN["reference_objs"].items()):
        path = os.path.join(REF_OBJS, name)
        assert digest(oracle.mesh_load_obj(path)) == strip(rec), name
        assert digest(product_mesh(path)) == strip(rec), name


@pytest.mark.skipif(not os.path.isdir("/root/reference/external/src/objUtil"), reason="needs the reference sources")
def test_live_reference_loader(tmp_path, oracle):
    """the compiled reference loader itself, array against array (not only digests)"""
    oracle.build_reference_obj_loader()
    paths = meshgen.write_generated_objs(str(tmp_path))
    for name, path in sorted(paths.items()):
        ref, mine = oracle.reference_obj_load(path), oracle.mesh_load_obj(path)
        assert ref["vbo"].shape == mine["vbo"].shape, name
        assert np.array_equal(ref["vbo"].view(np.uint32), mine["vbo"].view(np.uint32)), name
        assert (ref["tbo"] is None) == (mine["tbo"] is None), name
        if ref["tbo"] is not None:
            assert np.array_equal(ref["tbo"].view(np.uint32), mine["tbo"].view(np.uint32)), name
        assert np.array_equal(ref["bbox0"], mine["bbox0"]) and np.array_equal(ref["bbox1"], mine["bbox1"]), name
