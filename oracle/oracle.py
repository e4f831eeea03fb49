"""ctypes/numpy front end of the CPU oracle (oracle/svoslam_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (octree-slam_amd/) never imports it.
Parity status: unpinned by the reference (no reference tests exist); pinned by
the SURVEY.md Appendix C known-answer vectors in tests/golden/.  The OBJ loader alone is
pinned by the reference's own code (oracle/_ref/libobjref.so, reference_obj_load below).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_CHILDREN = 0x40000000
CHILD_MASK = 0x3FFFFFFF
RENDER_REFERENCE = 0
RENDER_CARRY = 1


def build(native=False):
    """Compile the oracle with gcc (seconds).  native=True adds -O3 -march=native."""
    env = dict(os.environ)
    if native:
        env["ORACLE_NATIVE"] = "1"
    subprocess.check_call(["make", "-C", _HERE, "-s"], env=env)
    name = "libsvoslam_oracle_native.so" if native else "libsvoslam_oracle.so"
    return os.path.join(_HERE, name)


class _Pool(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint32)), ("size", C.c_int)]


VARIANTS = ("fmad", "satu8")   # sensitivity variants (oracle/Makefile `variants`): NEVER the parity oracle


def lib(native=False, variant=None):
    """the oracle library.  variant: one of VARIANTS -- a build under the other reading of a compiler choice the reference's
    source leaves open (tests/test_oracle_compiler_choices.py); use it through `with using(L):` or the L= arguments"""
    global _LIB
    if _LIB is not None and not native and variant is None:
        return _LIB
    if variant is not None:
        assert variant in VARIANTS and not native
        name = "libsvoslam_oracle_%s.so" % variant
    else:
        name = "libsvoslam_oracle_native.so" if native else "libsvoslam_oracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "svoslam_oracle.c")
    src2 = os.path.join(_HERE, "svoslam_oracle_mesh.c")
    if (not os.path.exists(path)) or os.path.getmtime(path) < max(os.path.getmtime(src), os.path.getmtime(src2)):
        if variant is not None:
            subprocess.check_call(["make", "-C", _HERE, "-s", name])
        else:
            build(native)
    L = C.CDLL(path)
    f32p, u8p, u16p, u32p, i64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint16),
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_int64))
    L.ora_compute_key.restype = C.c_int64
    L.ora_compute_key.argtypes = [f32p, f32p, C.c_int, C.c_float]
    L.ora_depth_from_key.restype = C.c_int
    L.ora_depth_from_key.argtypes = [C.c_int64]
    L.ora_get_first_value_and_shift_down.restype = C.c_int
    L.ora_get_first_value_and_shift_down.argtypes = [i64p]
    L.ora_compute_keys.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, C.c_float, i64p]
    L.ora_prepass.restype = C.c_int
    L.ora_prepass.argtypes = [i64p, C.c_int, C.c_int, u32p, C.POINTER(C.c_int), C.POINTER(i64p)]
    L.ora_svo_from_point_cloud.argtypes = [f32p, u8p, C.c_int, C.c_int, C.POINTER(_Pool), f32p, C.c_float]
    L.ora_svo_from_voxel_grid.argtypes = [f32p, f32p, C.c_int, C.c_int, C.POINTER(_Pool), f32p, C.c_float]
    L.ora_extract_voxel_grid.restype = C.c_int
    L.ora_extract_voxel_grid.argtypes = [C.POINTER(_Pool), C.c_int, f32p, C.c_float, C.POINTER(f32p), C.POINTER(f32p)]
    L.ora_pool_free.argtypes = [C.POINTER(_Pool)]
    L.ora_pool_load_words.restype = C.c_int
    L.ora_pool_load_words.argtypes = [C.POINTER(_Pool), u32p, C.c_int]
    L.ora_cone_trace_svo.restype = C.c_int64
    L.ora_cone_trace_svo.argtypes = [u8p, C.c_int, C.c_int, C.c_float, f32p, u32p, f32p, C.c_float, C.c_int, i64p]
    L.ora_generate_vertex_map.argtypes = [u16p, f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]
    L.ora_generate_normal_map.argtypes = [f32p, f32p, C.c_int, C.c_int]
    L.ora_bilateral_filter.argtypes = [u16p, u16p, C.c_int, C.c_int]
    L.ora_subsample_depth_u16.argtypes = [u16p, C.c_int, C.c_int]
    L.ora_subsample_depth_f32.argtypes = [f32p, C.c_int, C.c_int]
    L.ora_subsample_f32.argtypes = [f32p, C.c_int, C.c_int]
    L.ora_subsample_rgb8.argtypes = [u8p, C.c_int, C.c_int]
    L.ora_color_to_intensity.argtypes = [u8p, f32p, C.c_int]
    L.ora_transform_vertex_map.argtypes = [f32p, f32p, C.c_int]
    L.ora_transform_normal_map.argtypes = [f32p, f32p, C.c_int]
    L.ora_point_cloud_bbox.argtypes = [f32p, C.c_int, f32p, f32p]
    L.ora_icp_cost2.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p]
    L.ora_gradient.argtypes = [f32p, f32p, C.c_int, C.c_int]
    L.ora_difference.argtypes = [f32p, f32p, f32p, C.c_int]
    L.ora_rgbd_cost.argtypes = [f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, f32p, f32p]
    L.ora_camera_set_rgbd.argtypes = [C.c_void_p, C.c_int]
    L.ora_raycast_model_depth.restype = C.c_int64
    L.ora_raycast_model_depth.argtypes = [u16p, C.c_int, C.c_int, C.c_float, C.c_float, f32p, u32p, f32p, C.c_float]
    L.ora_camera_set_model_depth.restype = C.c_int
    L.ora_camera_set_model_depth.argtypes = [C.c_void_p, u16p]
    L.ora_camera_set_frame_to_model.restype = C.c_int
    L.ora_camera_set_frame_to_model.argtypes = [C.c_void_p, C.c_int]
    L.ora_icp_cost.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p]
    L.ora_icp_cost.restype = C.c_int
    L.ora_icp_cost2_raw.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, i64p]
    L.ora_icp_finish.argtypes = [i64p, f32p, f32p]
    L.ora_solve_cholesky.argtypes = [C.c_int, f32p, f32p, f32p]
    for nm in ("ora_mat4_identity",):
        getattr(L, nm).argtypes = [f32p]
    L.ora_mat4_mul.argtypes = [f32p, f32p, f32p]
    L.ora_mat4_inverse.argtypes = [f32p, f32p]
    L.ora_mat4_translate.argtypes = [f32p, f32p, f32p]
    L.ora_mat4_rotate_deg.argtypes = [f32p, C.c_float, f32p, f32p]
    L.ora_mat4_look_at.argtypes = [f32p, f32p, f32p, f32p]
    L.ora_sincos.argtypes = [C.c_float, f32p, f32p]
    L.ora_icp_update_transform.argtypes = [f32p, f32p]
    L.ora_camera_create.restype = C.c_void_p
    L.ora_camera_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
    L.ora_camera_destroy.argtypes = [C.c_void_p]
    L.ora_camera_update.restype = C.c_int
    L.ora_camera_update.argtypes = [C.c_void_p, u16p, u8p, C.c_longlong]
    L.ora_camera_pose.argtypes = [C.c_void_p, f32p, f32p]
    L.ora_camera_set_pose.argtypes = [C.c_void_p, f32p, f32p]
    L.ora_camera_set_strict_reference.argtypes = [C.c_void_p, C.c_int]
    L.ora_camera_last_update.argtypes = [C.c_void_p, f32p]
    L.ora_camera_apply_delta.restype = C.c_int
    L.ora_camera_apply_delta.argtypes = [C.c_void_p, f32p, C.c_int, C.c_longlong]
    L.ora_camera_fusion_transform.argtypes = [C.c_void_p, f32p]
    L.ora_camera_last_system.argtypes = [C.c_void_p, f32p, f32p, f32p]
    L.ora_camera_tracking_lost_count.restype = C.c_int
    L.ora_camera_tracking_lost_count.argtypes = [C.c_void_p]
    L.ora_mesh_load_obj.restype = C.c_int
    L.ora_mesh_load_obj.argtypes = [C.c_char_p, C.POINTER(f32p), C.POINTER(f32p), C.POINTER(C.c_int), f32p, f32p]
    L.ora_load_bmp.restype = C.c_int
    L.ora_load_bmp.argtypes = [C.c_char_p, C.POINTER(f32p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ora_mesh_to_voxel_grid.restype = C.c_int
    L.ora_mesh_to_voxel_grid.argtypes = [f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int,
                                         C.POINTER(f32p), C.POINTER(f32p), C.POINTER(i64p)]
    if not native and variant is None:
        _LIB = L
    return L


import contextlib


@contextlib.contextmanager
def using(L):
    """every oracle call inside the block goes to library L (a sensitivity variant); objects created inside keep it"""
    global _LIB
    lib()   # (the default library is loaded, so that leaving the block restores it)
    old, _LIB = _LIB, L
    try:
        yield L
    finally:
        _LIB = old


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ keys
def compute_key(p, center, depth, edge):
    p = _f32(p); c = _f32(center)
    return int(lib().ora_compute_key(_p(p, C.c_float), _p(c, C.c_float), depth, edge))


def compute_keys(points, depth, center, edge):
    pts = _f32(points)
    n, stride = pts.shape
    keys = np.empty(n, dtype=np.int64)
    c = _f32(center)
    lib().ora_compute_keys(_p(pts, C.c_float), stride, n, depth, _p(c, C.c_float), edge, _p(keys, C.c_int64))
    return keys


def depth_from_key(k):
    return int(lib().ora_depth_from_key(int(k)))


def get_first_value_and_shift_down(k):
    v = C.c_int64(int(k))
    val = lib().ora_get_first_value_and_shift_down(C.byref(v))
    return int(val), int(v.value)


# ------------------------------------------------------------------ pool
class Pool:
    """Host node pool: 2 uint32 words per node (common_types.h:75-79)."""

    def __init__(self, L=None):
        self._L = L or lib()
        self._p = _Pool(None, 0)

    @property
    def size(self):
        return int(self._p.size)

    def words(self):
        if self._p.size == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(self._p.data, shape=(2 * self._p.size,)).copy()

    def set_words(self, words):
        """Replace pool content (used by KAT C6 to hand-set a node)."""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        assert words.size == 2 * self._p.size
        C.memmove(self._p.data, words.ctypes.data, words.nbytes)

    def load_words(self, words):
        """the pool becomes a copy of these nodes (any size: a map fused elsewhere, continued here)"""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        assert words.size % 2 == 0 and words.size >= 16
        assert self._L.ora_pool_load_words(C.byref(self._p), _p(words, C.c_uint32), words.size // 2) == 0

    def insert_cloud(self, points, colors, depth, center, edge):
        pts = _f32(points).reshape(-1, 3)
        col = np.ascontiguousarray(colors, dtype=np.uint8).reshape(-1, 3)
        c = _f32(center)
        self._L.ora_svo_from_point_cloud(_p(pts, C.c_float), _p(col, C.c_uint8), pts.shape[0], depth,
                                         C.byref(self._p), _p(c, C.c_float), edge)

    def insert_voxel_grid(self, centers, colors, depth, center, edge):
        ce = _f32(centers).reshape(-1, 4)
        co = _f32(colors).reshape(-1, 4)
        c = _f32(center)
        self._L.ora_svo_from_voxel_grid(_p(ce, C.c_float), _p(co, C.c_float), ce.shape[0], depth,
                                        C.byref(self._p), _p(c, C.c_float), edge)

    def extract(self, depth, center, edge):
        c = _f32(center)
        pc = C.POINTER(C.c_float)()
        pk = C.POINTER(C.c_float)()
        n = self._L.ora_extract_voxel_grid(C.byref(self._p), depth, _p(c, C.c_float), edge, C.byref(pc), C.byref(pk))
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        if n > 0:
            ce = np.ctypeslib.as_array(pc, shape=(n, 4)).copy()
            co = np.ctypeslib.as_array(pk, shape=(n, 4)).copy()
        else:
            ce = np.zeros((0, 4), np.float32); co = np.zeros((0, 4), np.float32)
        libc.free(pc); libc.free(pk)
        return ce, co

    def prepass(self, keys, depth):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        sizes = (C.c_int * depth)()
        codes = C.POINTER(C.c_int64)()
        w = self.words()
        total = self._L.ora_prepass(_p(keys, C.c_int64), keys.size, depth, _p(w, C.c_uint32), sizes, C.byref(codes))
        out = np.ctypeslib.as_array(codes, shape=(max(total, 1),))[:total].copy()
        libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]; libc.free(codes)
        return total, list(sizes), out

    def free(self):
        self._L.ora_pool_free(C.byref(self._p))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------ render
def cone_trace(words, w, h, fov, view, center, size, mode=RENDER_REFERENCE, L=None):
    L = L or lib()
    if isinstance(words, Pool):          # walk the oracle's own pool in place (no copy)
        wptr = words._p.data
    else:
        words = np.ascontiguousarray(words, dtype=np.uint32)
        wptr = _p(words, C.c_uint32)
    view = _f32(view).reshape(16)
    c = _f32(center)
    pos = np.zeros((h, w, 4), dtype=np.uint8)
    lv = C.c_int64(0)
    steps = L.ora_cone_trace_svo(_p(pos, C.c_uint8), w, h, fov, _p(view, C.c_float), wptr,
                                 _p(c, C.c_float), size, mode, C.byref(lv))
    return pos, int(steps), int(lv.value)


def raycast_model_depth(words, w, h, fx, fy, cam_to_world, center, size, L=None):
    """the map ray-cast into a depth image in the sensor's pixel grid and unit (own specification: svoslam_oracle.c,
    ora_raycast_model_depth) -> (uint16 [h, w], march steps)"""
    L = L or lib()
    if isinstance(words, Pool):
        wptr = words._p.data
    else:
        words = np.ascontiguousarray(words, dtype=np.uint32)
        wptr = _p(words, C.c_uint32)
    m = _f32(cam_to_world).reshape(16)
    c = _f32(center)
    out = np.zeros((h, w), dtype=np.uint16)
    steps = L.ora_raycast_model_depth(_p(out, C.c_uint16), w, h, fx, fy, _p(m, C.c_float), wptr, _p(c, C.c_float), size)
    return out, int(steps)


# ------------------------------------------------------------------ sensor
def vertex_map(depth, fx, fy, img_w, img_h):
    d = np.ascontiguousarray(depth, dtype=np.uint16)
    h, w = d.shape
    out = np.empty((h, w, 3), dtype=np.float32)
    lib().ora_generate_vertex_map(_p(d, C.c_uint16), _p(out, C.c_float), w, h, fx, fy, img_w, img_h)
    return out


def normal_map(vmap):
    v = _f32(vmap)
    h, w, _ = v.shape
    out = np.empty_like(v)
    lib().ora_generate_normal_map(_p(v, C.c_float), _p(out, C.c_float), w, h)
    return out


def bilateral(depth, L=None):
    d = np.ascontiguousarray(depth, dtype=np.uint16)
    h, w = d.shape
    out = np.empty_like(d)
    (L or lib()).ora_bilateral_filter(_p(d, C.c_uint16), _p(out, C.c_uint16), w, h)
    return out


def subsample_depth(img):
    a = np.ascontiguousarray(img).copy()
    h, w = a.shape
    if a.dtype == np.uint16:
        lib().ora_subsample_depth_u16(_p(a, C.c_uint16), w, h)
    else:
        a = a.astype(np.float32)
        lib().ora_subsample_depth_f32(_p(a, C.c_float), w, h)
    return a.reshape(-1)[: (w // 2) * (h // 2)].reshape(h // 2, w // 2).copy()


def subsample(img):
    a = np.ascontiguousarray(img).copy()
    if a.dtype == np.uint8:
        h, w, _ = a.shape
        lib().ora_subsample_rgb8(_p(a, C.c_uint8), w, h)
        return a.reshape(-1)[: 3 * (w // 2) * (h // 2)].reshape(h // 2, w // 2, 3).copy()
    a = a.astype(np.float32)
    h, w = a.shape
    lib().ora_subsample_f32(_p(a, C.c_float), w, h)
    return a.reshape(-1)[: (w // 2) * (h // 2)].reshape(h // 2, w // 2).copy()


def color_to_intensity(rgb):
    r = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(-1, 3)
    out = np.empty(r.shape[0], dtype=np.float32)
    lib().ora_color_to_intensity(_p(r, C.c_uint8), _p(out, C.c_float), r.shape[0])
    return out


def transform_vertex_map(v, m):
    a = _f32(v).copy(); mm = _f32(m).reshape(16)
    lib().ora_transform_vertex_map(_p(a, C.c_float), _p(mm, C.c_float), a.size // 3)
    return a


def transform_normal_map(v, m):
    a = _f32(v).copy(); mm = _f32(m).reshape(16)
    lib().ora_transform_normal_map(_p(a, C.c_float), _p(mm, C.c_float), a.size // 3)
    return a


def point_cloud_bbox(points, bbox0=(0, 0, 0), bbox1=(0, 0, 0)):
    p = _f32(points).reshape(-1, 3)
    b0 = _f32(bbox0).copy(); b1 = _f32(bbox1).copy()
    lib().ora_point_cloud_bbox(_p(p, C.c_float), p.shape[0], _p(b0, C.c_float), _p(b1, C.c_float))
    return b0, b1


def gradient(img):
    """Sobel / 8 (own specification of the reference's declared-only gradient()) -> [h, w, 2]"""
    a = _f32(img)
    h, w = a.shape
    out = np.zeros((h, w, 2), np.float32)
    lib().ora_gradient(_p(a, C.c_float), _p(out, C.c_float), w, h)
    return out


def difference(a, b):
    a, b = _f32(a), _f32(b)
    out = np.zeros_like(a)
    lib().ora_difference(_p(a, C.c_float), _p(b, C.c_float), _p(out, C.c_float), a.size)
    return out


def rgbd_cost(last_i, last_g, last_v, cur_i, cur_v, fx, fy, img_w, img_h):
    li, lg, lv, ci, cv = (_f32(x) for x in (last_i, last_g, last_v, cur_i, cur_v))
    h, w = li.shape
    A, b = np.zeros(36, np.float32), np.zeros(6, np.float32)
    lib().ora_rgbd_cost(_p(li, C.c_float), _p(lg, C.c_float), _p(lv, C.c_float), _p(ci, C.c_float), _p(cv, C.c_float), w, h,
                        C.c_float(fx), C.c_float(fy), img_w, img_h, _p(A, C.c_float), _p(b, C.c_float))
    return A.reshape(6, 6), b


def icp_cost2(last_v, last_n, cur_v, cur_n, L=None):
    lv, ln, cv, cn = _f32(last_v), _f32(last_n), _f32(cur_v), _f32(cur_n)
    h, w, _ = lv.shape
    A = np.empty(36, np.float32); b = np.empty(6, np.float32)
    (L or lib()).ora_icp_cost2(_p(lv, C.c_float), _p(ln, C.c_float), _p(cv, C.c_float), _p(cn, C.c_float), w, h,
                               _p(A, C.c_float), _p(b, C.c_float))
    return A.reshape(6, 6), b


def icp_cost(last_v, last_n, cur_v, cur_n, A0=None, b0=None):
    """computeICPCost (correspondence variant).  Returns (A, b, num_correspondences); A, b keep A0, b0 when there is none."""
    lv, ln, cv, cn = (np.ascontiguousarray(a, np.float32) for a in (last_v, last_n, cur_v, cur_n))
    h, w = lv.shape[0], lv.shape[1]
    A = np.zeros(36, np.float32) if A0 is None else np.ascontiguousarray(A0, np.float32).reshape(36).copy()
    b = np.zeros(6, np.float32) if b0 is None else np.ascontiguousarray(b0, np.float32).copy()
    m = lib().ora_icp_cost(_p(lv, C.c_float), _p(ln, C.c_float), _p(cv, C.c_float), _p(cn, C.c_float), w, h, _p(A, C.c_float),
                           _p(b, C.c_float))
    return A.reshape(6, 6), b, int(m)


def icp_cost2_raw(last_v, last_n, cur_v, cur_n, first_pixel=0, num_pixels=None):
    lv, ln, cv, cn = _f32(last_v), _f32(last_n), _f32(cur_v), _f32(cur_n)
    h, w, _ = lv.shape
    if num_pixels is None:
        num_pixels = w * h - first_pixel
    acc = np.zeros(27, np.int64)
    lib().ora_icp_cost2_raw(_p(lv, C.c_float), _p(ln, C.c_float), _p(cv, C.c_float), _p(cn, C.c_float),
                            first_pixel, num_pixels, w, h, _p(acc, C.c_int64))
    return acc


def icp_finish(acc):
    acc = np.ascontiguousarray(acc, dtype=np.int64)
    A = np.empty(36, np.float32); b = np.empty(6, np.float32)
    lib().ora_icp_finish(_p(acc, C.c_int64), _p(A, C.c_float), _p(b, C.c_float))
    return A.reshape(6, 6), b


def solve_cholesky(A, b):
    A = _f32(A).reshape(36); b = _f32(b).reshape(6)
    x = np.zeros(6, np.float32)
    lib().ora_solve_cholesky(6, _p(A, C.c_float), _p(b, C.c_float), _p(x, C.c_float))
    return x


# ------------------------------------------------------------------ mat4 (column-major flat[16])
def mat4_identity():
    m = np.empty(16, np.float32); lib().ora_mat4_identity(_p(m, C.c_float)); return m


def mat4_mul(a, b):
    a = _f32(a).reshape(16); b = _f32(b).reshape(16); o = np.empty(16, np.float32)
    lib().ora_mat4_mul(_p(a, C.c_float), _p(b, C.c_float), _p(o, C.c_float)); return o


def mat4_inverse(a):
    a = _f32(a).reshape(16); o = np.empty(16, np.float32)
    lib().ora_mat4_inverse(_p(a, C.c_float), _p(o, C.c_float)); return o


def mat4_translate(m, v):
    m = _f32(m).reshape(16); v = _f32(v); o = np.empty(16, np.float32)
    lib().ora_mat4_translate(_p(m, C.c_float), _p(v, C.c_float), _p(o, C.c_float)); return o


def mat4_rotate_deg(m, angle, axis):
    m = _f32(m).reshape(16); ax = _f32(axis); o = np.empty(16, np.float32)
    lib().ora_mat4_rotate_deg(_p(m, C.c_float), angle, _p(ax, C.c_float), _p(o, C.c_float)); return o


def look_at(eye, center, up):
    e, c, u = _f32(eye), _f32(center), _f32(up); o = np.empty(16, np.float32)
    lib().ora_mat4_look_at(_p(e, C.c_float), _p(c, C.c_float), _p(u, C.c_float), _p(o, C.c_float)); return o


def sincos(a):
    s = C.c_float(); c = C.c_float()
    lib().ora_sincos(a, C.byref(s), C.byref(c)); return s.value, c.value


def icp_update_transform(x):
    x = _f32(x); o = np.empty(16, np.float32)
    lib().ora_icp_update_transform(_p(x, C.c_float), _p(o, C.c_float)); return o


# ------------------------------------------------------------------ tracker
class Camera:
    """CPU restatement of sensor::RGBDCamera (src/sensor/rgbd_camera.cpp)."""

    def __init__(self, w, h, fx, fy, L=None):
        self._L = L or lib()
        self._c = self._L.ora_camera_create(w, h, fx, fy)
        self.w, self.h = w, h

    def update(self, depth, rgb, timestamp):
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        r = np.ascontiguousarray(rgb, dtype=np.uint8)
        return self._L.ora_camera_update(self._c, _p(d, C.c_uint16), _p(r, C.c_uint8), timestamp)

    def pose(self):
        p = np.empty(3, np.float32); o = np.empty(9, np.float32)
        self._L.ora_camera_pose(self._c, _p(p, C.c_float), _p(o, C.c_float))
        return p, o

    def set_pose(self, position, orientation):
        p, o = _f32(position).reshape(3), _f32(orientation).reshape(9)
        self._L.ora_camera_set_pose(self._c, _p(p, C.c_float), _p(o, C.c_float))

    def last_update(self):
        """update_trans of the frame most recently tracked (identity for a first frame)"""
        m = np.empty(16, np.float32)
        self._L.ora_camera_last_update(self._c, _p(m, C.c_float))
        return m

    def apply_delta(self, update_trans, levels_lost, timestamp):
        """the pose step of update() for a frame tracked elsewhere (update_trans None: pose unchanged)"""
        if update_trans is None:
            return self._L.ora_camera_apply_delta(self._c, None, 0, timestamp)
        m = np.ascontiguousarray(update_trans, dtype=np.float32)
        return self._L.ora_camera_apply_delta(self._c, _p(m, C.c_float), int(levels_lost), timestamp)

    def set_rgbd(self, enable=True):
        self._L.ora_camera_set_rgbd(self._c, 1 if enable else 0)

    def set_strict_reference(self, strict=True):
        """False: this build's corrected tracker (svoslam_oracle.c ora_camera_set_strict_reference); before the first frame"""
        self._L.ora_camera_set_strict_reference(self._c, 1 if strict else 0)

    def set_model_depth(self, depth):
        if depth is None:       # no model: frame to frame until the next one
            return self._L.ora_camera_set_model_depth(self._c, None)
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        assert d.shape == (self.h, self.w)
        return self._L.ora_camera_set_model_depth(self._c, _p(d, C.c_uint16))

    def set_frame_to_model(self, enable=True):
        return self._L.ora_camera_set_frame_to_model(self._c, 1 if enable else 0)

    def tracking_lost_count(self):
        return int(self._L.ora_camera_tracking_lost_count(self._c))

    def fusion_transform(self):
        m = np.empty(16, np.float32)
        self._L.ora_camera_fusion_transform(self._c, _p(m, C.c_float))
        return m

    def last_system(self):
        A = np.empty(36, np.float32); b = np.empty(6, np.float32); x = np.empty(6, np.float32)
        self._L.ora_camera_last_system(self._c, _p(A, C.c_float), _p(b, C.c_float), _p(x, C.c_float))
        return A.reshape(6, 6), b, x

    def __del__(self):
        try:
            self._L.ora_camera_destroy(self._c)
        except Exception:
            pass


# ------------------------------------------------------------------ mesh path
def _take(ptr, shape, dtype):
    n = int(np.prod(shape))
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    if n > 0:
        out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).reshape(shape).copy()
    else:
        out = np.zeros(shape, dtype)
    libc.free(ptr)
    return out


def mesh_load_obj(path):
    """-> dict(vbo [T,3,3], tbo [T,3,2] or None, bbox0, bbox1) as Scene::loadObjFile builds the Mesh"""
    vb, tb = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    tsz = C.c_int(0)
    b0, b1 = np.zeros(3, np.float32), np.zeros(3, np.float32)
    n = lib().ora_mesh_load_obj(path.encode(), C.byref(vb), C.byref(tb), C.byref(tsz), _p(b0, C.c_float), _p(b1, C.c_float))
    if n < 0:
        raise IOError(path)
    vbo = _take(vb, (n, 3, 3), np.float32)
    tbo = _take(tb, (tsz.value // 6, 3, 2), np.float32) if tsz.value > 0 else None
    return {"vbo": vbo, "tbo": tbo, "bbox0": b0, "bbox1": b1}


def voxel_grid_to_mesh(centers, colors, scale_factor, cube_vbo, cube_ibo, cube_nbo):
    """voxelGridToMesh -> (vbo, ibo, nbo, cbo) flat arrays"""
    ce, co = _f32(centers).reshape(-1, 4), _f32(colors).reshape(-1, 4)
    cv, cn = _f32(cube_vbo).reshape(-1), _f32(cube_nbo).reshape(-1)
    ci = np.ascontiguousarray(cube_ibo, dtype=np.int32).reshape(-1)
    n = ce.shape[0]
    vbo, nbo, cbo = (np.empty(n * cv.size, np.float32) for _ in range(3))
    ibo = np.empty(n * ci.size, np.int32)
    lib().ora_voxel_grid_to_mesh(_p(ce, C.c_float), _p(co, C.c_float), n, C.c_float(scale_factor), _p(cv, C.c_float), cv.size,
                                 _p(ci, C.c_int), ci.size, _p(cn, C.c_float), _p(vbo, C.c_float), _p(ibo, C.c_int),
                                 _p(nbo, C.c_float), _p(cbo, C.c_float))
    return vbo, ibo, nbo, cbo


_REF_OBJ = None


def build_reference_obj_loader():
    """make -C oracle ref: the reference's own objUtil sources + ref_obj_shim.cpp -> oracle/_ref/libobjref.so"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return os.path.join(_HERE, "_ref", "libobjref.so")


def reference_obj_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libobjref.so"))


def reference_obj_load(path):
    """The REFERENCE's loader (compiled from /root/reference, see ref_obj_shim.cpp), same dict as mesh_load_obj.
    Malformed files are undefined behaviour there (face index 0, empty tokens): feed it well-formed input."""
    global _REF_OBJ
    if _REF_OBJ is None:
        _REF_OBJ = C.CDLL(os.path.join(_HERE, "_ref", "libobjref.so"))
        _REF_OBJ.ref_obj_load.restype = C.c_int
        _REF_OBJ.ref_obj_free.argtypes = [C.c_void_p]
    vb, tb = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    tsz = C.c_int(0)
    b0, b1 = np.zeros(3, np.float32), np.zeros(3, np.float32)
    n = _REF_OBJ.ref_obj_load(str(path).encode(), C.byref(vb), C.byref(tb), C.byref(tsz), _p(b0, C.c_float), _p(b1, C.c_float))
    if n < 0:
        raise IOError(path)
    vbo = np.ctypeslib.as_array(vb, (max(n, 1) * 9,))[:n * 9].copy().reshape(n, 3, 3)
    tbo = np.ctypeslib.as_array(tb, (tsz.value,)).copy().reshape(-1, 3, 2) if tsz.value > 0 else None
    _REF_OBJ.ref_obj_free(vb)
    if tsz.value > 0:
        _REF_OBJ.ref_obj_free(tb)
    return {"vbo": vbo, "tbo": tbo, "bbox0": b0, "bbox1": b1}


def load_bmp(path):
    d = C.POINTER(C.c_float)()
    w, h = C.c_int(0), C.c_int(0)
    if lib().ora_load_bmp(path.encode(), C.byref(d), C.byref(w), C.byref(h)) != 0:
        raise IOError(path)
    return _take(d, (h.value, w.value, 3), np.float32)


def mesh_to_voxel_grid(mesh, tex, log_n, log_t=3):
    vbo = _f32(mesh["vbo"]).reshape(-1)
    tbo = _f32(mesh["tbo"]).reshape(-1) if mesh.get("tbo") is not None else np.zeros(0, np.float32)
    texa = _f32(tex).reshape(-1) if tex is not None else np.zeros(0, np.float32)
    tw, th = (tex.shape[1], tex.shape[0]) if tex is not None else (0, 0)
    b0, b1 = _f32(mesh["bbox0"]), _f32(mesh["bbox1"])
    pc, pk, pi = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_int64)()
    n = lib().ora_mesh_to_voxel_grid(_p(vbo, C.c_float), vbo.size // 9, _p(tbo, C.c_float), tbo.size, _p(texa, C.c_float), tw, th,
                                     _p(b0, C.c_float), _p(b1, C.c_float), log_n, log_t, C.byref(pc), C.byref(pk), C.byref(pi))
    return _take(pc, (n, 4), np.float32), _take(pk, (n, 4), np.float32), _take(pi, (n,), np.int64)
