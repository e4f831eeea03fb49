"""Regenerates tests/golden/ref_glm.json: inputs and outputs of the REFERENCE's own vendored glm 0.9.5.4 for the host
matrix functions of the hot path (oracle/ref_glm_shim.cpp compiled by `make -C oracle ref` from /root/reference/
external/include/glm): inverse (createRays), rotate / translate / operator* / vec4 x mat4 (RGBDCamera::update), lookAt,
and this_trans of rgbd_camera.cpp:154-158.  Floats are stored as their uint32 bit patterns.  Needs /root/reference
(build container only).  Run from the repository root:  python tests/golden/make_ref_glm_golden.py"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ref_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], stderr=subprocess.DEVNULL)
    return C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libglmref.so"))


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32).reshape(-1).tolist()


def cases(rng):
    """deterministic inputs: view-like rigid matrices, general matrices, small ICP updates, axes, angles"""
    out = {"inverse": [], "mul": [], "translate": [], "rotate": [], "look_at": [], "vec4_mul_mat4": [], "mat4_mul_vec4": [], "icp_update": []}
    def rigid():
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        m = np.eye(4); m[:3, :3] = R; m[:3, 3] = rng.uniform(-3, 3, 3)
        return np.ascontiguousarray(m.T, np.float32).reshape(16)        # column-major
    for _ in range(40):
        out["inverse"].append([rigid()])
        out["inverse"].append([(rng.uniform(-2, 2, 16)).astype(np.float32)])
        out["mul"].append([rigid(), rigid()])
        out["translate"].append([rigid(), rng.uniform(-2, 2, 3).astype(np.float32)])
        ax = rng.normal(size=3).astype(np.float32)
        out["rotate"].append([rigid(), np.float32(rng.uniform(-180, 180)), ax])
        out["rotate"].append([np.eye(4, dtype=np.float32).reshape(16), np.float32(rng.uniform(-2, 2)), np.eye(3, dtype=np.float32)[rng.integers(0, 3)]])
        eye = rng.uniform(-3, 3, 3).astype(np.float32)
        out["look_at"].append([eye, (eye + rng.normal(size=3)).astype(np.float32), np.array([0, 1, 0], np.float32)])
        out["vec4_mul_mat4"].append([np.append(rng.uniform(-3, 3, 3), 1.0).astype(np.float32), rigid()])
        out["mat4_mul_vec4"].append([rigid(), np.append(rng.uniform(-3, 3, 3), rng.integers(0, 2)).astype(np.float32)])
        out["icp_update"].append([(rng.normal(size=6) * np.array([0.01, 0.01, 0.01, 0.02, 0.02, 0.02])).astype(np.float32)])
    return out


def run_ref(L, name, args):
    o16, o4 = np.zeros(16, np.float32), np.zeros(4, np.float32)
    if name == "inverse": L.ref_glm_inverse(fp(args[0]), fp(o16)); return o16
    if name == "mul": L.ref_glm_mul(fp(args[0]), fp(args[1]), fp(o16)); return o16
    if name == "translate": L.ref_glm_translate(fp(args[0]), fp(args[1]), fp(o16)); return o16
    if name == "rotate": L.ref_glm_rotate_deg(fp(args[0]), C.c_float(float(args[1])), fp(np.ascontiguousarray(args[2], np.float32)), fp(o16)); return o16
    if name == "look_at": L.ref_glm_look_at(fp(args[0]), fp(args[1]), fp(args[2]), fp(o16)); return o16
    if name == "vec4_mul_mat4": L.ref_glm_vec4_mul_mat4(fp(args[0]), fp(args[1]), fp(o4)); return o4
    if name == "mat4_mul_vec4": L.ref_glm_mat4_mul_vec4(fp(args[0]), fp(args[1]), fp(o4)); return o4
    if name == "icp_update": L.ref_icp_update_transform(fp(args[0]), fp(o16)); return o16
    raise KeyError(name)


def main():
    L = ref_lib()
    rng = np.random.default_rng(20240929)
    golden = {}
    for name, lst in cases(rng).items():
        golden[name] = [{"in": [bits(np.atleast_1d(a)) for a in args], "out": bits(run_ref(L, name, args))} for args in lst]
    path = os.path.join(ROOT, "tests", "golden", "ref_glm.json")
    json.dump(golden, open(path, "w"), separators=(",", ":"))
    print("wrote", path, {k: len(v) for k, v in golden.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
