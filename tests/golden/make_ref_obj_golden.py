"""Generates tests/golden/ref_obj_loader.json from the REFERENCE's own OBJ loader.

Run in the build container only (needs /root/reference): `make -C oracle ref` compiles the reference's
external/src/objUtil sources where they lie into oracle/_ref/libobjref.so; this script runs that loader
(Scene::loadObjFile + objToMesh, reference src/world/scene.cpp:26-33,115-133) over
  * the OBJ files tests/meshgen.py generates (inputs reproducible anywhere, so the digests pin the
    oracle's and the product's loaders on every machine), and
  * the reference's objs/*.obj (inputs only present in the build container; pyramid.obj is skipped:
    its faces use index 0, which reads points[-1] in the reference = undefined behaviour),
and stores sha256 digests of the VBO / TBO float bits plus the bounding box bits.

    python tests/golden/make_ref_obj_golden.py
"""
import glob
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def digest(mesh):
    """the record compared by tests/test_ref_obj_loader.py"""
    vbo = np.ascontiguousarray(mesh["vbo"], np.float32)
    rec = {"n_tris": int(vbo.shape[0]), "vbo_sha256": hashlib.sha256(vbo.tobytes()).hexdigest(),
           "tbo_floats": 0 if mesh["tbo"] is None else int(np.asarray(mesh["tbo"]).size),
           "tbo_sha256": None if mesh["tbo"] is None else hashlib.sha256(np.ascontiguousarray(mesh["tbo"], np.float32).tobytes()).hexdigest(),
           "bbox0_bits": [int(x) for x in np.asarray(mesh["bbox0"], np.float32).view(np.uint32)],
           "bbox1_bits": [int(x) for x in np.asarray(mesh["bbox1"], np.float32).view(np.uint32)]}
    return rec


def main():
    import meshgen
    from oracle import oracle as ora
    ora.build_reference_obj_loader()
    out = {"generated": {}, "reference_objs": {}}
    with tempfile.TemporaryDirectory() as d:
        for name, path in sorted(meshgen.write_generated_objs(d).items()):
            rec = digest(ora.reference_obj_load(path))
            rec["input_sha256"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
            out["generated"][name] = rec
    for path in sorted(glob.glob("/root/reference/objs/*.obj")):
        name = os.path.basename(path)
        if name == "pyramid.obj":
            continue
        rec = digest(ora.reference_obj_load(path))
        rec["input_sha256"] = hashlib.sha256(open(path, "rb").read()).hexdigest()
        out["reference_objs"][name] = rec
    with open(os.path.join(HERE, "ref_obj_loader.json"), "w") as fp:
        json.dump(out, fp, indent=1, sort_keys=True)
        fp.write("\n")
    print("wrote %d generated + %d reference records" % (len(out["generated"]), len(out["reference_objs"])))


if __name__ == "__main__":
    main()
