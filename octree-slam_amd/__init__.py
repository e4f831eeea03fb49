"""octree_slam_amd -- Python front end of libsvoslam_hip.so (include/svoslam.h).

The product is the C-ABI shared library built from csrc/*.hip for gfx950; this
package only binds it with ctypes and moves pointers of torch CUDA tensors
across the boundary (PyTorch = device memory, streams, torch.distributed).
There is NO CPU fallback: every compute entry point raises SvoslamError when
the library or a gfx950 device is missing.

The directory is called ``octree-slam_amd`` (not importable by name); load it
with ``svoslam_pkg.load()`` from the repo root, which registers it as the module
``octree_slam_amd``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvoslam_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "svoslam.h")

MAX_DEPTH = 16
FLAG_CHILDREN = 0x40000000
CHILD_MASK = 0x3FFFFFFF
RENDER_REFERENCE = 0
RENDER_CARRY = 1
PYRAMID_ITERS = (10, 5, 4)  # rgbd_camera.cpp:19, index = pyramid level
DELTA_FLOATS = 20  # SVOSLAM_DELTA_FLOATS: update_trans[16], levels lost (int32 bits), 3 x padding


class SvoslamError(RuntimeError):
    pass


def build(verbose=False, force=False):
    """Compile csrc/*.hip for gfx950 into libsvoslam_hip.so (hipcc, in tree)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_svoslam_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=verbose, force=force)


class _PoolStruct(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("size", C.c_int32), ("capacity", C.c_int32), ("d_size", C.c_void_p),
                ("pending", C.c_int32), ("pending_bound", C.c_int64), ("tracker", C.c_void_p)]


class MeshStruct(C.Structure):
    _fields_ = [("vbo", C.POINTER(C.c_float)), ("tbo", C.POINTER(C.c_float)), ("n_tris", C.c_int32), ("tbosize", C.c_int32),
                ("bbox0", C.c_float * 3), ("bbox1", C.c_float * 3)]


class TextureStruct(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32)]


class FuseStats(C.Structure):
    _fields_ = [("num_points", C.c_int32), ("num_split", C.c_int32), ("pass_sizes", C.c_int32 * (MAX_DEPTH + 1)),
                ("pool_size_before", C.c_int32), ("pool_size_after", C.c_int32)]


_lib = None
_vp, _i32, _f32 = C.c_void_p, C.c_int32, C.c_float
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); this table is also what tests/test_abi_symbols.py checks against include/svoslam.h
SIGNATURES = {
    "svoslam_abi_version": (C.c_int, []),
    "svoslam_config_get": (C.c_int, [C.c_void_p]),
    "svoslam_config_set": (C.c_int, [C.c_void_p]),
    "svoslam_status_string": (C.c_char_p, [C.c_int]),
    "svoslam_last_error": (C.c_char_p, []),
    "svoslam_device_arch": (C.c_char_p, []),
    "svoslam_pool_init": (C.c_int, [C.POINTER(_PoolStruct), _i32, _vp]),
    "svoslam_pool_reserve": (C.c_int, [C.POINTER(_PoolStruct), _i32, _vp]),
    "svoslam_pool_free": (C.c_int, [C.POINTER(_PoolStruct)]),
    "svoslam_pool_sync": (C.c_int, [C.POINTER(_PoolStruct), _vp]),
    "svoslam_pool_reset": (C.c_int, [C.POINTER(_PoolStruct), _vp]),
    "svoslam_pool_expand": (C.c_int, [C.POINTER(_PoolStruct), _fp, C.POINTER(_f32), _fp, _vp]),
    "svoslam_camera_reset": (C.c_int, [_vp]),
    "svoslam_pool_save": (C.c_int, [C.POINTER(_PoolStruct), C.c_char_p, _fp, _f32, _i32, _vp]),
    "svoslam_pool_touch": (C.c_int, [C.POINTER(_PoolStruct)]),
    "svoslam_pool_march_accel": (C.c_int, [C.POINTER(_PoolStruct), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "svoslam_pool_set_nodes": (C.c_int, [C.POINTER(_PoolStruct), C.POINTER(C.c_uint32), _i32, _vp]),
    "svoslam_pool_evict_subtree": (C.c_int, [C.POINTER(_PoolStruct), C.POINTER(C.c_uint8), _i32, C.c_char_p, _vp]),
    "svoslam_pool_restore_subtree": (C.c_int, [C.POINTER(_PoolStruct), C.c_char_p, _vp]),
    "svoslam_subtree_file_nodes": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(_i32)]),
    "svoslam_pool_copy": (C.c_int, [C.POINTER(_PoolStruct), C.POINTER(_PoolStruct), _vp]),
    "svoslam_pool_load": (C.c_int, [C.POINTER(_PoolStruct), C.c_char_p, _fp, C.POINTER(_f32), C.POINTER(_i32), _vp]),
    "svoslam_svo_from_point_cloud_async": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _fp, _f32, _vp]),
    "svoslam_svo_fuse_sort": (C.c_int, [_vp, _vp, _i32, _i32, _fp, _f32, _vp]),
    "svoslam_svo_fuse_sort_frame": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _i32, _fp, _f32, _vp, _vp]),
    "svoslam_svo_fuse_plan": (C.c_int, [_vp, _i32, _i32, C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_commit": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_split_early": (C.c_int, [_vp, _i32, _i32, C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_plan_structure": (C.c_int, [_vp, _i32, _i32, C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_adopt_sorted": (C.c_int, [_vp, _vp, _vp, _i32, _i32]),
    "svoslam_svo_fuse_sort_frame_band": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _i32, _fp, _f32, _i32, _i32, _vp]),
    "svoslam_svo_fuse_merge_sorted": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _i32, _vp, _vp, _vp]),
    "svoslam_svo_fuse_export_sorted": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "svoslam_pool_structure_begin": (C.c_int, [C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_commit_to": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _i32, _i32, _vp]),
    "svoslam_svo_fuse_commit_deferred": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_apply": (C.c_int, [_vp, C.POINTER(_PoolStruct), _vp]),
    "svoslam_svo_fuse_keyrange_commit": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _i32, _i32, _vp, C.c_int64, _vp]),
    "svoslam_svo_fuse_keyrange_apply": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), C.POINTER(C.c_void_p), _i32, _vp]),
    "svoslam_svo_fuse_keyrange_status": (C.c_int, [_vp, C.POINTER(C.c_int32), _vp]),
    "svoslam_svo_fuse_keyrange_discard": (C.c_int, [_vp, C.POINTER(_PoolStruct)]),
    "svoslam_frame_reader_open": (C.c_int, [C.POINTER(_vp), C.c_char_p, _f32]),
    "svoslam_frame_reader_close": (C.c_int, [_vp]),
    "svoslam_frame_reader_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "svoslam_frame_reader_rewind": (C.c_int, [_vp]),
    "svoslam_frame_reader_next_host": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_longlong), C.POINTER(_i32)]),
    "svoslam_frame_reader_next": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_longlong), C.POINTER(_i32), _vp]),
    "svoslam_focal_from_fov": (C.c_int, [_i32, _i32, _f32, _f32, _fp, _fp]),
    "svoslam_image_load": (C.c_int, [C.c_char_p, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "svoslam_workspace_create": (C.c_int, [C.POINTER(_vp)]),
    "svoslam_workspace_destroy": (C.c_int, [_vp]),
    "svoslam_svo_from_point_cloud": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _fp, _f32,
                                                C.POINTER(FuseStats), _vp]),
    "svoslam_svo_from_voxel_grid": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.POINTER(_PoolStruct), _fp, _f32,
                                               C.POINTER(FuseStats), _vp]),
    "svoslam_extract_voxel_grid": (C.c_int, [_vp, C.POINTER(_PoolStruct), _i32, _fp, _f32, C.POINTER(_vp),
                                              C.POINTER(_vp), C.POINTER(_i32), _vp]),
    "svoslam_mesh_load_obj": (C.c_int, [C.c_char_p, C.POINTER(MeshStruct)]),
    "svoslam_mesh_free": (C.c_int, [C.POINTER(MeshStruct)]),
    "svoslam_texture_load_bmp": (C.c_int, [C.c_char_p, C.POINTER(TextureStruct)]),
    "svoslam_texture_free": (C.c_int, [C.POINTER(TextureStruct)]),
    "svoslam_mesh_to_voxel_grid": (C.c_int, [_vp, C.POINTER(MeshStruct), C.POINTER(TextureStruct), _i32, _i32, C.POINTER(_vp),
                                             C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _fp, _vp]),
    "svoslam_mesh_last_fragments": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "svoslam_voxel_grid_to_mesh": (C.c_int, [_vp, _vp, _vp, _i32, _f32, _fp, _i32, C.POINTER(C.c_int32), _i32, _fp, _vp, _vp, _vp, _vp, _vp]),
    "svoslam_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "svoslam_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "svoslam_runner_run_model": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_longlong), _fp, _i32, _vp, _i32, _i32, _vp,
                                           _f32, C.POINTER(_i32), _vp]),
    "svoslam_runner_create": (C.c_int, [C.POINTER(_vp), _vp, C.POINTER(_PoolStruct), _i32, _i32, _i32, _fp, _f32, _f32, _f32, _i32]),
    "svoslam_runner_destroy": (C.c_int, [_vp]),
    "svoslam_runner_timeline": (C.c_int, [_vp, _fp, _i32, C.POINTER(_i32)]),
    "svoslam_runner_run": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_longlong), _fp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "svoslam_runner_bbox": (C.c_int, [_vp, _fp]),
    "svoslam_runner_run_sharded": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_longlong), _fp, _i32, C.POINTER(_vp),
                                             C.POINTER(_vp), C.POINTER(C.c_uint8), C.POINTER(_vp), _i32, _i32, _vp, _vp]),
    "svoslam_runner_run_sharded_presorted": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_longlong), _fp, _i32, C.POINTER(_vp),
                                                       C.POINTER(_vp), C.POINTER(C.c_uint8), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                                       C.POINTER(_vp), _i32, _i32, _vp, _vp]),
    "svoslam_mailbox_create": (C.c_int, [C.POINTER(_vp), _i32, _i32]),
    "svoslam_mailbox_destroy": (C.c_int, [_vp]),
    "svoslam_mailbox_handle": (C.c_int, [_vp, _vp]),
    "svoslam_mailbox_connect": (C.c_int, [_vp, _vp]),
    "svoslam_mailbox_connect_local": (C.c_int, [_vp, C.POINTER(_vp)]),
    "svoslam_mailbox_all_gather": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "svoslam_mailbox_all_reduce_f64": (C.c_int, [_vp, _vp, _i32, _vp]),
    "svoslam_mailbox_failed": (C.c_int, [_vp, C.POINTER(_i32)]),
    "svoslam_mailbox_set_wait_limit": (C.c_int, [_vp, C.c_uint32]),
    "svoslam_mailbox_post": (C.c_int, [_vp, _vp, _i32, _vp]),
    "svoslam_mailbox_collect": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_scene_create": (C.c_int, [C.POINTER(_vp)]),
    "svoslam_scene_destroy": (C.c_int, [_vp]),
    "svoslam_scene_load_obj": (C.c_int, [_vp, C.c_char_p]),
    "svoslam_scene_load_bmp": (C.c_int, [_vp, C.c_char_p]),
    "svoslam_scene_set_octree": (C.c_int, [_vp, _f32, _fp, _f32, _i32]),
    "svoslam_scene_voxelize_meshes": (C.c_int, [_vp, _i32, _i32, _vp]),
    "svoslam_scene_extract_voxel_grid": (C.c_int, [_vp, _vp]),
    "svoslam_scene_add_point_cloud": (C.c_int, [_vp, _fp, _vp, _vp, _i32, _fp, _fp, _vp]),
    "svoslam_scene_voxel_grid": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i32), _fp]),
    "svoslam_scene_svo": (C.c_int, [_vp, C.POINTER(_vp), _fp, _fp, C.POINTER(_i32), C.POINTER(_i32)]),
    "svoslam_free": (C.c_int, [_vp]),
    "svoslam_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "svoslam_cone_trace_svo": (C.c_int, [_vp, _i32, _i32, _f32, _fp, _vp, _fp, _f32, _i32, _vp, _vp]),
    "svoslam_cone_trace_svo_band": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f32, _fp, _vp, _fp, _f32, _i32, _vp, _vp]),
    "svoslam_cone_trace_release": (C.c_int, [_vp, _i32]),
    "svoslam_cone_trace_timing": (C.c_int, [_i32]),
    "svoslam_cone_trace_timing_read": (C.c_int, [_fp, C.POINTER(_i32)]),
    "svoslam_stage_timing": (C.c_int, [C.c_uint32]),
    "svoslam_stage_timing_read": (C.c_int, [_i32, _fp, C.POINTER(_i32)]),
    "svoslam_generate_vertex_map": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _f32, _i32, _i32, _vp]),
    "svoslam_generate_vertex_map_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _vp]),
    "svoslam_generate_normal_map": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_bilateral_filter": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_subsample_depth_u16": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_subsample_depth_f32": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_subsample_f32": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_subsample_rgb8": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_color_to_intensity": (C.c_int, [_vp, _vp, _i32, _vp]),
    "svoslam_transform_vertex_map": (C.c_int, [_vp, _fp, _i32, _vp]),
    "svoslam_transform_normal_map": (C.c_int, [_vp, _fp, _i32, _vp]),
    "svoslam_transform_vertex_map_dmat": (C.c_int, [_vp, _vp, _i32, _vp]),
    "svoslam_point_cloud_bbox": (C.c_int, [_vp, _i32, _fp, _fp, _vp]),
    "svoslam_point_cloud_bbox_device": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "svoslam_gradient": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "svoslam_difference": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "svoslam_rgbd_cost": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _i32, _i32, _fp, _fp, _vp]),
    "svoslam_camera_set_rgbd": (C.c_int, [_vp, _i32]),
    "svoslam_camera_set_strict_reference": (C.c_int, [_vp, _i32]),
    "svoslam_raycast_model_depth": (C.c_int, [_vp, _i32, _i32, _f32, _f32, _fp, _vp, _vp, _fp, _f32, _vp, _vp]),
    "svoslam_camera_set_model_depth": (C.c_int, [_vp, _vp, _vp]),
    "svoslam_camera_set_frame_to_model": (C.c_int, [_vp, _i32]),
    "svoslam_icp_cost2": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _fp, _fp, _vp]),
    "svoslam_icp_cost": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _fp, _fp, C.POINTER(_i32), _vp]),
    "svoslam_icp_accumulate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "svoslam_camera_create": (C.c_int, [C.POINTER(_vp), _i32, _i32, _f32, _f32]),
    "svoslam_camera_destroy": (C.c_int, [_vp]),
    "svoslam_camera_set_band": (C.c_int, [_vp, _i32, _i32]),
    "svoslam_camera_set_acc": (C.c_int, [_vp, _vp]),
    "svoslam_camera_update": (C.c_int, [_vp, _vp, _vp, C.c_longlong, C.POINTER(_i32), _vp]),
    "svoslam_camera_begin": (C.c_int, [_vp, _vp, _vp, C.c_longlong, C.POINTER(_i32), _vp]),
    "svoslam_camera_prepare": (C.c_int, [_vp, _vp, _vp, C.c_longlong, C.POINTER(_i32), _vp]),
    "svoslam_camera_track": (C.c_int, [_vp, _vp]),
    "svoslam_camera_pair_delta": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "svoslam_camera_apply_delta": (C.c_int, [_vp, _vp, C.c_longlong, C.POINTER(_i32), _vp]),
    "svoslam_camera_icp_iters": (C.c_int, [_i32]),
    "svoslam_camera_icp_accumulate": (C.c_int, [_vp, _i32, _i32, _vp]),
    "svoslam_camera_acc": (_vp, [_vp]),
    "svoslam_camera_icp_solve": (C.c_int, [_vp, _i32, _i32, _vp]),
    "svoslam_camera_end": (C.c_int, [_vp, _vp]),
    "svoslam_camera_pose": (C.c_int, [_vp, _fp, _fp, _vp]),
    "svoslam_camera_fusion_transform_device": (_vp, [_vp]),
    "svoslam_camera_last_system": (C.c_int, [_vp, _fp, _fp, _fp, _vp]),
    "svoslam_camera_last_vertex": (_vp, [_vp, _i32]),
    "svoslam_camera_last_normal": (_vp, [_vp, _i32]),
    "svoslam_camera_tracking_lost_count": (C.c_int, [_vp, C.POINTER(_i32), _vp]),
    "svoslam_camera_latest_timestamp": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(C.c_longlong)]),
    "svoslam_camera_track_profile": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), _vp]),
    "svoslam_timer_start": (C.c_int, [_vp]),
    "svoslam_timer_stop": (C.c_int, [_vp, _fp]),
}


def lib():
    """Load libsvoslam_hip.so; raises SvoslamError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvoslamError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                           "There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime: whichever of the two is mapped first serves the whole process, and torch finds no
    # device when this library's (/opt/rocm) came first -- e.g. __graft_entry__.build() followed by smoke() in one process.
    # torch is the device-memory / stream provider of every caller of this binding, so it goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status):
    if status != 0:
        L = lib()
        raise SvoslamError("svoslam status %d (%s): %s" % (status, L.svoslam_status_string(status).decode(),
                                                          L.svoslam_last_error().decode()))


_raw_stream = None


def _stream():
    """torch's current HIP stream as a raw handle (the C call, not the Stream object: this runs ~20x per frame)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or False
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _fa(values, n):
    a = (C.c_float * n)(*[float(v) for v in np.asarray(values, dtype=np.float32).reshape(-1)])
    return a


class ConfigStruct(C.Structure):
    """include/svoslam.h svoslam_config"""
    _fields_ = [(n, C.c_int32) for n in ("march_bricks", "track_mode", "track_workers", "track_stream", "runner_deferred", "runner_lead",
                                         "runner_prio", "runner_replicas", "runner_timeline", "sort_pairs", "graphs", "march_ahead")] + [("reserved", C.c_int32 * 4)]


def get_config():
    """the library's settings as a dict (include/svoslam.h svoslam_config)"""
    c = ConfigStruct()
    check(lib().svoslam_config_get(C.byref(c)))
    return {n: int(getattr(c, n)) for n, _ in ConfigStruct._fields_ if n != "reserved"}


def configure(**settings):
    """svoslam_config_set: e.g. configure(track_mode=1, graphs=1); takes effect for cameras / runners created afterwards and
    for the next render / fusion call.  Returns the settings as they were."""
    c = ConfigStruct()
    check(lib().svoslam_config_get(C.byref(c)))
    before = {n: int(getattr(c, n)) for n, _ in ConfigStruct._fields_ if n != "reserved"}
    for k, v in settings.items():
        if k not in before:
            raise KeyError("svoslam_config has no field %r" % (k,))
        setattr(c, k, int(v))
    check(lib().svoslam_config_set(C.byref(c)))
    return before


def config_env(**settings):
    """{"SVOSLAM_CONFIG": "name=value,..."}: the same settings for a child process (read once when its library starts)"""
    return {"SVOSLAM_CONFIG": ",".join("%s=%d" % (k, int(v)) for k, v in settings.items())}


def device_arch():
    a = lib().svoslam_device_arch()
    return a.decode() if a else None


# ----------------------------------------------------------------------------- pool / fusion
class Workspace:
    def __init__(self):
        self._h = C.c_void_p()
        check(lib().svoslam_workspace_create(C.byref(self._h)))

    def close(self):
        if self._h:
            lib().svoslam_workspace_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """Device node pool (svoslam_pool).  words() copies it to the host as uint32."""

    def __init__(self, capacity_nodes=8):
        self._p = _PoolStruct(None, 0, 0, None, 0, 0)
        check(lib().svoslam_pool_init(C.byref(self._p), capacity_nodes, _stream()))

    @property
    def size(self):
        if self._p.pending > 0:   # asynchronous fusion calls in flight: fetch the exact size (blocking)
            check(lib().svoslam_pool_sync(C.byref(self._p), _stream()))
        return int(self._p.size)

    @property
    def capacity(self):
        return int(self._p.capacity)

    @property
    def data_ptr(self):
        return int(self._p.d_data)

    def reserve(self, capacity_nodes):
        check(lib().svoslam_pool_reserve(C.byref(self._p), capacity_nodes, _stream()))

    def words(self):
        import torch
        torch.cuda.synchronize()
        n = 2 * self.size
        out = np.empty(n, dtype=np.uint32)
        hip = _hip()
        r = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(self.data_ptr), C.c_size_t(n * 4), 2)
        if r != 0:
            raise SvoslamError("hipMemcpy D2H failed: %d" % r)
        return out

    def set_words(self, words):
        """replace the pool's contents (svoslam_pool_set_nodes: validates child pointers, resets the device-resident size
        and the reservations of asynchronous fusions, so fusing into the loaded tree allocates after it)"""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        check(lib().svoslam_pool_set_nodes(C.byref(self._p), words.ctypes.data_as(C.POINTER(C.c_uint32)), words.size // 2, _stream()))

    def evict_subtree(self, path, file):
        """page the sub-tree below the node reached by the octant `path` out to `file` (svoslam_pool_evict_subtree)"""
        p = (C.c_uint8 * len(path))(*[int(o) for o in path])
        check(lib().svoslam_pool_evict_subtree(C.byref(self._p), p, len(path), str(file).encode(), _stream()))

    def restore_subtree(self, file):
        check(lib().svoslam_pool_restore_subtree(C.byref(self._p), str(file).encode(), _stream()))

    def copy_from(self, other):
        """become a byte-identical replica of `other` (blocking)"""
        check(lib().svoslam_pool_copy(C.byref(self._p), C.byref(other._p), _stream()))

    def reset(self):
        """empty map (8 zeroed root children), same allocation"""
        check(lib().svoslam_pool_reset(C.byref(self._p), _stream()))

    def march_accel(self):
        """what the ray march of this pool runs on: {grid: bool, bricks: 1 in use / 0 not built / -1 allocation failed (tree march),
        brick_shift}"""
        g, b, sh = C.c_int32(0), C.c_int32(0), C.c_int32(-1)
        check(lib().svoslam_pool_march_accel(C.byref(self._p), C.byref(g), C.byref(b), C.byref(sh)))
        return {"grid": bool(g.value), "bricks": int(b.value), "brick_shift": int(sh.value)}

    def expand(self, center, edge_length, toward):
        """doubles the root cube towards `toward` (re-rooting); returns the new (center, edge_length)"""
        c = _fa(center, 3)
        e = C.c_float(float(edge_length))
        check(lib().svoslam_pool_expand(C.byref(self._p), c, C.byref(e), _fa(toward, 3), _stream()))
        return tuple(float(v) for v in c), float(e.value)

    def save(self, path, center, edge_length, max_depth):
        """checkpoint: linear tree + root parameters (svoslam_pool_save)"""
        check(lib().svoslam_pool_save(C.byref(self._p), str(path).encode(), _fa(center, 3), float(edge_length), int(max_depth),
                                      _stream()))

    def load(self, path):
        """resume from a checkpoint; returns (center, edge_length, max_depth)"""
        c, e, d = (C.c_float * 3)(), C.c_float(0), C.c_int32(0)
        check(lib().svoslam_pool_load(C.byref(self._p), str(path).encode(), c, C.byref(e), C.byref(d), _stream()))
        return tuple(float(v) for v in c), float(e.value), int(d.value)

    def close(self):
        if self._p.d_data:
            lib().svoslam_pool_free(C.byref(self._p))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_hiplib = None


def _hip():
    global _hiplib
    if _hiplib is None:
        _hiplib = C.CDLL("libamdhip64.so")
        _hiplib.hipMemcpy.restype = C.c_int
        _hiplib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hiplib


def svo_from_point_cloud(ws, points, colors, max_depth, pool, center, edge_length):
    """svo::svoFromPointCloud.  points: cuda float32 [n,3]; colors: cuda uint8 [n,3]."""
    n = int(points.shape[0]) if points is not None else 0
    stats = FuseStats()
    check(lib().svoslam_svo_from_point_cloud(ws._h, _ptr(points), _ptr(colors), n, max_depth, C.byref(pool._p),
                                             _fa(center, 3), float(edge_length), C.byref(stats), _stream()))
    return stats


def svo_from_point_cloud_async(ws, points, colors, max_depth, pool, center, edge_length):
    """svoFromPointCloud without any host round trip (size stays on the device until pool.size is read)."""
    n = int(points.shape[0]) if points is not None else 0
    check(lib().svoslam_svo_from_point_cloud_async(ws._h, _ptr(points), _ptr(colors), n, max_depth, C.byref(pool._p),
                                                   _fa(center, 3), float(edge_length), _stream()))


def svo_fuse_sort(ws, points, max_depth, center, edge_length):
    """phase 1 of the asynchronous fusion: keys + sort (workspace only)."""
    n = int(points.shape[0]) if points is not None else 0
    check(lib().svoslam_svo_fuse_sort(ws._h, _ptr(points), n, max_depth, _fa(center, 3), float(edge_length), _stream()))


def svo_fuse_sort_frame(ws, depth_image, pose_ptr, fx, fy, max_depth, center, edge_length, bbox7=None):
    """sort phase straight from a depth frame and a device-resident pose (vertex map + transform + bbox + keys in one launch)"""
    h, w = depth_image.shape[-2], depth_image.shape[-1]
    check(lib().svoslam_svo_fuse_sort_frame(ws._h, _ptr(depth_image), C.c_void_p(int(pose_ptr)), w, h, float(fx), float(fy),
                                            int(max_depth), _fa(center, 3), float(edge_length), _ptr(bbox7), _stream()))


def svo_fuse_sort_frame_band(ws, depth_image, pose_ptr, fx, fy, max_depth, center, edge_length, first_row, rows):
    """svo_fuse_sort_frame for one row band of the frame; the sorted point indices are whole-image indices"""
    h, w = depth_image.shape[-2], depth_image.shape[-1]
    check(lib().svoslam_svo_fuse_sort_frame_band(ws._h, _ptr(depth_image), C.c_void_p(int(pose_ptr)), w, h, float(fx), float(fy),
                                                 int(max_depth), _fa(center, 3), float(edge_length), int(first_row), int(rows), _stream()))


def svo_fuse_merge_sorted(keys_lists, idx_lists, keys_out, idx_out):
    """merge sorted (key, index) lists with ascending, disjoint index ranges (row bands in order) -> keys_out, idx_out"""
    n = len(keys_lists)
    kp = (C.c_void_p * n)(*[k.data_ptr() for k in keys_lists])
    ip = (C.c_void_p * n)(*[k.data_ptr() for k in idx_lists])
    cn = (C.c_int32 * n)(*[int(k.shape[0]) for k in keys_lists])
    check(lib().svoslam_svo_fuse_merge_sorted(kp, ip, cn, n, _ptr(keys_out), _ptr(idx_out), _stream()))


def svo_fuse_export_sorted(ws, n, keys_out, idx_out):
    """the outcome of the workspace's sort phase -> keys_out (int64 cuda tensor [n]), idx_out (int32 cuda tensor [n])"""
    check(lib().svoslam_svo_fuse_export_sorted(ws._h, int(n), _ptr(keys_out), _ptr(idx_out), _stream()))


def svo_fuse_adopt_sorted(ws, keys, idx, max_depth):
    """sorted keys / point indices from elsewhere become the outcome of the workspace's sort phase (they must stay alive
    until the commit has run)"""
    check(lib().svoslam_svo_fuse_adopt_sorted(ws._h, _ptr(keys), _ptr(idx), int(keys.shape[0]), int(max_depth)))


def svo_fuse_plan(ws, n, max_depth, pool):
    """phase 2: split planning against the pool's current tree (reads the pool)."""
    check(lib().svoslam_svo_fuse_plan(ws._h, int(n), max_depth, C.byref(pool._p), _stream()))


def pool_structure_begin(pool):
    """before a sequence of svo_fuse_plan_structure calls: the structure-side size := the pool's size"""
    check(lib().svoslam_pool_structure_begin(C.byref(pool._p), _stream()))


def svo_fuse_plan_structure(ws, n, max_depth, pool):
    """plan + every split with its links, independent of the previous frame's commit (colour words)"""
    check(lib().svoslam_svo_fuse_plan_structure(ws._h, int(n), int(max_depth), C.byref(pool._p), _stream()))


def svo_fuse_split_early(ws, n, max_depth, pool):
    """between plan and commit: the planned splits' child tiles, written where no ray march can see them yet"""
    check(lib().svoslam_svo_fuse_split_early(ws._h, int(n), int(max_depth), C.byref(pool._p), _stream()))


def svo_fuse_commit(ws, colors, max_depth, pool):
    """phase 3: splits, leaf blend, mip levels (writes the pool)."""
    n = int(colors.shape[0]) if colors is not None else 0
    check(lib().svoslam_svo_fuse_commit(ws._h, _ptr(colors), n, max_depth, C.byref(pool._p), _stream()))


def svo_fuse_commit_to(ws, colors, max_depth, pool, slot, keep_plan):
    """phase 3 applied to one of several identical replicas of the map (see svoslam.h)"""
    n = int(colors.shape[0]) if colors is not None else 0
    check(lib().svoslam_svo_fuse_commit_to(ws._h, _ptr(colors), n, max_depth, C.byref(pool._p), int(slot), 1 if keep_plan else 0,
                                           _stream()))


def svo_fuse_commit_deferred(ws, colors, max_depth, pool):
    """phase 3 without a store a concurrent render of the pool's present state can see; svo_fuse_apply publishes it"""
    n = int(colors.shape[0]) if colors is not None else 0
    check(lib().svoslam_svo_fuse_commit_deferred(ws._h, _ptr(colors), n, max_depth, C.byref(pool._p), _stream()))


def svo_fuse_apply(ws, pool):
    check(lib().svoslam_svo_fuse_apply(ws._h, C.byref(pool._p), _stream()))


KEYRANGE_OVERFLOW, KEYRANGE_MISMATCH, KEYRANGE_USED_WORD = 2, 4, 10
KEYRANGE_FIXED_WORDS = 11264    # header + fixed-size lists of a delta; its tiles, links and words follow


def svo_fuse_keyrange_commit(ws, sorted_keys, sorted_idx, colors, max_depth, pool, rank, world, delta):
    """key-range sharded fusion, first half: plan + (invisible) commit of rank's slice of the frame's sorted keys, its delta into `delta`
    (a uint32 / int32 device tensor with room: 64 bytes per key of the slice is ample)"""
    n = int(sorted_keys.shape[0])
    check(lib().svoslam_svo_fuse_keyrange_commit(ws._h, _ptr(sorted_keys), _ptr(sorted_idx), _ptr(colors), n, int(max_depth), C.byref(pool._p),
                                                 int(rank), int(world), _ptr(delta), int(delta.numel() * delta.element_size()), _stream()))


def svo_fuse_keyrange_apply(ws, sorted_keys, max_depth, pool, deltas):
    """second half, after the all-gather: every rank's delta (rank order, the own one included) into this replica"""
    n = int(sorted_keys.shape[0])
    arr = (C.c_void_p * len(deltas))(*[int(d.data_ptr()) for d in deltas])
    check(lib().svoslam_svo_fuse_keyrange_apply(ws._h, _ptr(sorted_keys), n, int(max_depth), C.byref(pool._p), arr, len(deltas), _stream()))


def svo_fuse_keyrange_discard(ws, pool):
    """the delta of the last keyrange_commit was all that was wanted: no apply follows on this pool"""
    check(lib().svoslam_svo_fuse_keyrange_discard(ws._h, C.byref(pool._p)))


def svo_fuse_keyrange_status(ws):
    """flags of the last apply on this workspace (blocking): 0 = applied"""
    f = C.c_int32(0)
    check(lib().svoslam_svo_fuse_keyrange_status(ws._h, C.byref(f), _stream()))
    return int(f.value)


def svo_from_voxel_grid(ws, centers, colors, max_depth, pool, center, edge_length):
    """svo::svoFromVoxelGrid.  centers, colors: cuda float32 [n,4]."""
    n = int(centers.shape[0]) if centers is not None else 0
    stats = FuseStats()
    check(lib().svoslam_svo_from_voxel_grid(ws._h, _ptr(centers), _ptr(colors), n, max_depth, C.byref(pool._p),
                                            _fa(center, 3), float(edge_length), C.byref(stats), _stream()))
    return stats


def extract_voxel_grid(ws, pool, max_depth, center, edge_length):
    """svo::extractVoxelGridFromSVO -> (centers[n,4], colors[n,4]) numpy float32."""
    import torch
    pc, pk, n = C.c_void_p(), C.c_void_p(), C.c_int32(0)
    check(lib().svoslam_extract_voxel_grid(ws._h, C.byref(pool._p), max_depth, _fa(center, 3), float(edge_length),
                                           C.byref(pc), C.byref(pk), C.byref(n), _stream()))
    torch.cuda.synchronize()
    ce = np.zeros((n.value, 4), np.float32)
    co = np.zeros((n.value, 4), np.float32)
    if n.value > 0:
        _hip().hipMemcpy(C.c_void_p(ce.ctypes.data), pc, C.c_size_t(ce.nbytes), 2)
        _hip().hipMemcpy(C.c_void_p(co.ctypes.data), pk, C.c_size_t(co.nbytes), 2)
        lib().svoslam_free(pc)
        lib().svoslam_free(pk)
    return ce, co


# ----------------------------------------------------------------------------- mesh path
class Mesh:
    """Host mesh as Scene::loadObjFile builds it (recentred, non-indexed)."""

    def __init__(self, path=None):
        self._m = MeshStruct()
        if path is not None:
            check(lib().svoslam_mesh_load_obj(str(path).encode(), C.byref(self._m)))

    @property
    def n_tris(self):
        return int(self._m.n_tris)

    def vbo(self):
        return np.ctypeslib.as_array(self._m.vbo, shape=(self.n_tris, 3, 3)).copy() if self.n_tris else np.zeros((0, 3, 3), np.float32)

    def tbo(self):
        n = int(self._m.tbosize)
        return np.ctypeslib.as_array(self._m.tbo, shape=(n // 6, 3, 2)).copy() if n else None

    def bbox(self):
        return np.array(list(self._m.bbox0), np.float32), np.array(list(self._m.bbox1), np.float32)

    def __del__(self):
        try:
            lib().svoslam_mesh_free(C.byref(self._m))
        except Exception:
            pass


class Texture:
    def __init__(self, path=None):
        self._t = TextureStruct()
        if path is not None:
            check(lib().svoslam_texture_load_bmp(str(path).encode(), C.byref(self._t)))

    def data(self):
        h, w = int(self._t.height), int(self._t.width)
        return np.ctypeslib.as_array(self._t.data, shape=(h, w, 3)).copy()

    def __del__(self):
        try:
            lib().svoslam_texture_free(C.byref(self._t))
        except Exception:
            pass


def mesh_to_voxel_grid(ws, mesh, tex, log_n, log_t=3, want_indices=True):
    """voxelization::meshToVoxelGrid -> (centers cuda [n,4], colors cuda [n,4], indices numpy or None, scale)."""
    import torch
    pc, pk, pi, n, scale = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int32(0), C.c_float(0)
    check(lib().svoslam_mesh_to_voxel_grid(ws._h, C.byref(mesh._m), C.byref(tex._t) if tex is not None else None, log_n, log_t,
                                           C.byref(pc), C.byref(pk), C.byref(pi) if want_indices else None, C.byref(n),
                                           C.byref(scale), _stream()))
    torch.cuda.synchronize()
    cnt = n.value
    ce = torch.empty((cnt, 4), dtype=torch.float32, device="cuda")
    co = torch.empty((cnt, 4), dtype=torch.float32, device="cuda")
    idx = None
    if cnt > 0:
        _hip().hipMemcpy(C.c_void_p(ce.data_ptr()), pc, C.c_size_t(cnt * 16), 3)
        _hip().hipMemcpy(C.c_void_p(co.data_ptr()), pk, C.c_size_t(cnt * 16), 3)
        lib().svoslam_free(pc); lib().svoslam_free(pk)
        if want_indices:
            idx = np.empty(cnt, np.uint64)
            _hip().hipMemcpy(C.c_void_p(idx.ctypes.data), pi, C.c_size_t(cnt * 8), 2)
            lib().svoslam_free(pi)
    return ce, co, idx, float(scale.value)


def mesh_last_fragments(ws):
    """(cell, triangle) fragments of the workspace's last mesh_to_voxel_grid (measurement aid)"""
    n = C.c_int64(0)
    check(lib().svoslam_mesh_last_fragments(ws._h, C.byref(n)))
    return int(n.value)


def voxel_grid_to_mesh(ws, centers, colors, scale_factor, cube_vbo, cube_ibo, cube_nbo):
    """voxelization::voxelGridToMesh: centers, colors cuda float32 [n,4]; cube arrays numpy -> cuda (vbo, ibo, nbo, cbo)."""
    import torch
    cv = np.ascontiguousarray(cube_vbo, np.float32).reshape(-1)
    cn = np.ascontiguousarray(cube_nbo, np.float32).reshape(-1)
    ci = np.ascontiguousarray(cube_ibo, np.int32).reshape(-1)
    if cv.size != cn.size:
        raise ValueError("cube vbo and nbo have different sizes")
    n = int(centers.shape[0])
    vbo, nbo, cbo = (torch.empty(n * cv.size, dtype=torch.float32, device="cuda") for _ in range(3))
    ibo = torch.empty(n * ci.size, dtype=torch.int32, device="cuda")
    check(lib().svoslam_voxel_grid_to_mesh(ws._h, _ptr(centers), _ptr(colors), n, float(scale_factor),
                                           cv.ctypes.data_as(_fp), cv.size, ci.ctypes.data_as(C.POINTER(C.c_int32)), ci.size,
                                           cn.ctypes.data_as(_fp), _ptr(vbo), _ptr(ibo), _ptr(nbo), _ptr(cbo), _stream()))
    return vbo, ibo, nbo, cbo


class Runner:
    """Native frame scheduler (csrc/runner.hip): track -> back-project -> fuse -> raycast of a list of frames on four
    HIP streams, enqueued by ONE call."""

    def __init__(self, cam, pool, width, height, max_depth, center, edge, fx, fy, render_mode):
        self._h = C.c_void_p()
        self._cam, self._pool = cam, pool          # keep the handles alive
        check(lib().svoslam_runner_create(C.byref(self._h), cam._h, C.byref(pool._p), width, height, max_depth, _fa(center, 3),
                                          float(edge), float(fx), float(fy), int(render_mode)))

    def run(self, depths, rgbs, timestamps, views, image, row_first, rows, counters=None):
        n = len(timestamps)
        dp = (C.c_void_p * n)(*[d.data_ptr() for d in depths])
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in rgbs])
        ts = (C.c_longlong * n)(*[int(t) for t in timestamps])
        vw = np.ascontiguousarray(np.stack([np.asarray(v, np.float32).reshape(16) for v in views]), np.float32)
        check(lib().svoslam_runner_run(self._h, dp, rp, ts, vw.ctypes.data_as(_fp), n, _ptr(image), int(row_first), int(rows),
                                       _ptr(counters), _stream()))

    def run_sharded(self, depths, rgbs, timestamps, views, deltas, delta_events, march, images, row_first, rows, counters=None):
        """one rank of a frame-sharded session (svoslam_runner_run_sharded): deltas = per-frame tensors (or None) of
        DELTA_FLOATS floats, delta_events = per-frame torch.cuda.Event (recorded) or None, march = per-frame bool,
        images = per-frame output tensor (or None where march is False)"""
        n = len(timestamps)
        dp = (C.c_void_p * n)(*[d.data_ptr() for d in depths])
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in rgbs])
        ts = (C.c_longlong * n)(*[int(t) for t in timestamps])
        vw = np.ascontiguousarray(np.stack([np.asarray(v, np.float32).reshape(16) for v in views]), np.float32)
        dl = (C.c_void_p * n)(*[(d.data_ptr() if d is not None else None) for d in deltas])
        ev = (C.c_void_p * n)(*[(e.cuda_event if e is not None else None) for e in delta_events]) if delta_events is not None else None
        mf = (C.c_uint8 * n)(*[1 if m else 0 for m in march])
        im = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in images])
        self._keep = (depths, rgbs, deltas, delta_events, images)   # alive until the next call (the work is asynchronous)
        check(lib().svoslam_runner_run_sharded(self._h, dp, rp, ts, vw.ctypes.data_as(_fp), n, dl, ev, mf, im, int(row_first), int(rows),
                                               _ptr(counters), _stream()))

    def run_sharded_presorted(self, depths, rgbs, timestamps, views, deltas, delta_events, march, images, sorted_keys, sorted_idx,
                              sorted_events, row_first, rows, counters=None):
        """run_sharded with every frame's sorted keys / point indices supplied (svoslam_runner_run_sharded_presorted):
        sorted_keys / sorted_idx = per-frame int64 / int32 cuda tensors, sorted_events = per-frame torch.cuda.Event or None"""
        n = len(timestamps)
        dp = (C.c_void_p * n)(*[d.data_ptr() for d in depths])
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in rgbs])
        ts = (C.c_longlong * n)(*[int(t) for t in timestamps])
        vw = np.ascontiguousarray(np.stack([np.asarray(v, np.float32).reshape(16) for v in views]), np.float32)
        dl = (C.c_void_p * n)(*[(d.data_ptr() if d is not None else None) for d in deltas])
        ev = (C.c_void_p * n)(*[(e.cuda_event if e is not None else None) for e in delta_events]) if delta_events is not None else None
        mf = (C.c_uint8 * n)(*[1 if m else 0 for m in march])
        im = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in images])
        sk = (C.c_void_p * n)(*[k.data_ptr() for k in sorted_keys])
        si = (C.c_void_p * n)(*[k.data_ptr() for k in sorted_idx])
        se = (C.c_void_p * n)(*[(e.cuda_event if e is not None else None) for e in sorted_events]) if sorted_events is not None else None
        self._keep = (depths, rgbs, deltas, delta_events, images, sorted_keys, sorted_idx, sorted_events)
        check(lib().svoslam_runner_run_sharded_presorted(self._h, dp, rp, ts, vw.ctypes.data_as(_fp), n, dl, ev, mf, im, sk, si, se,
                                                         int(row_first), int(rows), _ptr(counters), _stream()))

    def run_model(self, depths, rgbs, timestamps, views, image, row_first, rows, min_coverage=0.5, counters=None):
        """the frame loop with frame-to-model tracking inside the library (svoslam_runner_run_model); returns the number of frames
        whose ray-cast model was accepted.  Blocking."""
        n = len(timestamps)
        dp = (C.c_void_p * n)(*[d.data_ptr() for d in depths])
        rp = (C.c_void_p * n)(*[r.data_ptr() for r in rgbs])
        ts = (C.c_longlong * n)(*[int(t) for t in timestamps])
        vw = np.ascontiguousarray(np.stack([np.asarray(v, np.float32).reshape(16) for v in views]), np.float32)
        used = C.c_int32(0)
        check(lib().svoslam_runner_run_model(self._h, dp, rp, ts, vw.ctypes.data_as(_fp), n, _ptr(image), int(row_first), int(rows),
                                             _ptr(counters), float(min_coverage), C.byref(used), _stream()))
        return int(used.value)

    def bbox(self):
        """{min xyz, max xyz, any} of the last frame's point cloud (main.cpp:43)"""
        b = (C.c_float * 7)()
        check(lib().svoslam_runner_bbox(self._h, b))
        return np.array(list(b), np.float32)

    def timeline(self, max_frames=4096):
        """[frames, 10] stage times in ms of the last run (svoslam_config.runner_timeline = 1 when the runner was created)"""
        out = np.full((max_frames, 10), -1.0, np.float32)
        n = _i32(0)
        check(lib().svoslam_runner_timeline(self._h, out.ctypes.data_as(_fp), max_frames, C.byref(n)))
        return out[: n.value]

    def close(self):
        if self._h:
            lib().svoslam_runner_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mailbox:
    """One-shot peer-to-peer exchange of small records between the ranks of a node (csrc/mailbox.hip): all_gather of up to
    2 KB per rank, all_reduce of up to 256 float64 added in rank order; enqueued on the current stream, no host round trip."""

    def __init__(self, rank, world):
        self._h = C.c_void_p()
        self.rank, self.world = int(rank), int(world)
        check(lib().svoslam_mailbox_create(C.byref(self._h), self.rank, self.world))

    def handle(self):
        """64 bytes (numpy uint8) that the other PROCESSES need to map this rank's inbox"""
        buf = (C.c_uint8 * 64)()
        check(lib().svoslam_mailbox_handle(self._h, buf))
        return np.frombuffer(bytes(buf), dtype=np.uint8).copy()

    def connect(self, handles):
        """handles: uint8 array [world, 64], row r from rank r (this rank's own row is ignored)"""
        h = np.ascontiguousarray(handles, dtype=np.uint8)
        assert h.shape == (self.world, 64)
        check(lib().svoslam_mailbox_connect(self._h, h.ctypes.data_as(C.c_void_p)))

    def connect_local(self, boxes):
        """all mailboxes of the session live in this process (tests; one process driving several devices)"""
        arr = (C.c_void_p * self.world)(*[b._h for b in boxes])
        check(lib().svoslam_mailbox_connect_local(self._h, arr))

    def all_gather(self, src, dst):
        """dst[world, ...] <- every rank's src (same byte count on every rank, a multiple of 4, <= 2048)"""
        nbytes = src.numel() * src.element_size()
        assert dst.numel() * dst.element_size() == nbytes * self.world
        check(lib().svoslam_mailbox_all_gather(self._h, _ptr(src), nbytes, _ptr(dst), _stream()))

    def all_reduce_f64(self, values):
        """values (float64 cuda tensor, <= 256 entries) <- sum over ranks, added in rank order: the same bits everywhere"""
        check(lib().svoslam_mailbox_all_reduce_f64(self._h, _ptr(values), int(values.numel()), _stream()))

    def post(self, src):
        """first half of a collective: this rank's record into every inbox"""
        check(lib().svoslam_mailbox_post(self._h, _ptr(src), src.numel() * src.element_size(), _stream()))

    def collect(self, dst, nbytes, reduce_f64=False):
        """second half: every rank's record of the epoch just posted -> dst[world, ...] (or their rank-ordered float64 sum)"""
        check(lib().svoslam_mailbox_collect(self._h, _ptr(dst), int(nbytes), 1 if reduce_f64 else 0, _stream()))

    def set_wait_limit(self, polls):
        check(lib().svoslam_mailbox_set_wait_limit(self._h, int(polls)))

    def failed(self):
        f = _i32(0)
        check(lib().svoslam_mailbox_failed(self._h, C.byref(f)))
        return bool(f.value)

    def close(self):
        if self._h:
            lib().svoslam_mailbox_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Mirror of world::Scene (include/octree_slam/world/scene.h)."""

    def __init__(self):
        self._h = C.c_void_p()
        check(lib().svoslam_scene_create(C.byref(self._h)))

    def load_obj(self, path):
        check(lib().svoslam_scene_load_obj(self._h, str(path).encode()))

    def load_bmp(self, path):
        check(lib().svoslam_scene_load_bmp(self._h, str(path).encode()))

    def set_octree(self, resolution, center, size, depth_override=0):
        check(lib().svoslam_scene_set_octree(self._h, float(resolution), _fa(center, 3), float(size), int(depth_override)))

    def voxelize_meshes(self, octree=False, log_n=0):
        check(lib().svoslam_scene_voxelize_meshes(self._h, 1 if octree else 0, int(log_n), _stream()))

    def extract_voxel_grid_from_octree(self):
        check(lib().svoslam_scene_extract_voxel_grid(self._h, _stream()))

    def add_point_cloud_to_octree(self, origin, points, colors, bbox0, bbox1):
        check(lib().svoslam_scene_add_point_cloud(self._h, _fa(origin, 3), _ptr(points), _ptr(colors), int(points.numel() // 3),
                                                  _fa(bbox0, 3), _fa(bbox1, 3), _stream()))

    def voxel_grid(self):
        pc, pk, n, sc = C.c_void_p(), C.c_void_p(), C.c_int32(0), C.c_float(0)
        check(lib().svoslam_scene_voxel_grid(self._h, C.byref(pc), C.byref(pk), C.byref(n), C.byref(sc)))
        ce = copy_from_device(pc.value, (n.value, 4), np.float32) if n.value else np.zeros((0, 4), np.float32)
        co = copy_from_device(pk.value, (n.value, 4), np.float32) if n.value else np.zeros((0, 4), np.float32)
        return ce, co, float(sc.value)

    def svo(self):
        """-> dict(data_ptr, center, size, num_nodes, max_depth): the SVO view of Scene::svo()."""
        pd, c, sz, nn, md = C.c_void_p(), (C.c_float * 3)(), C.c_float(0), C.c_int32(0), C.c_int32(0)
        check(lib().svoslam_scene_svo(self._h, C.byref(pd), c, C.byref(sz), C.byref(nn), C.byref(md)))
        return {"data_ptr": int(pd.value or 0), "center": np.array(list(c), np.float32), "size": float(sz.value),
                "num_nodes": int(nn.value), "max_depth": int(md.value)}

    def pool_words(self):
        v = self.svo()
        return copy_from_device(v["data_ptr"], (2 * v["num_nodes"],), np.uint32)

    def close(self):
        if self._h:
            lib().svoslam_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- rendering
def cone_trace_svo(out, fov, view, pool_ptr, center, size, mode=RENDER_REFERENCE, counters=None):
    """rendering::coneTraceSVO into `out` (cuda uint8 [h,w,4])."""
    h, w = int(out.shape[0]), int(out.shape[1])
    check(lib().svoslam_cone_trace_svo(_ptr(out), w, h, float(fov), _fa(view, 16), C.c_void_p(int(pool_ptr)),
                                       _fa(center, 3), float(size), int(mode), _ptr(counters), _stream()))
    return out


def raycast_model_depth(out, fx, fy, pool_ptr, center, size, cam_to_world=None, cam_to_world_ptr=None, counters=None):
    """the map ray-cast into a depth image in the sensor's pixel grid (`out`: cuda uint16 [h, w]); the pose as a host matrix
    (16 floats, column-major: the fusion transform) or as a device pointer (Camera.fusion_transform_ptr()).  Own
    specification: include/svoslam.h, svoslam_raycast_model_depth."""
    h, w = int(out.shape[0]), int(out.shape[1])
    if (cam_to_world is None) == (cam_to_world_ptr is None):
        raise ValueError("exactly one of cam_to_world / cam_to_world_ptr")
    host = _fa(cam_to_world, 16) if cam_to_world is not None else None
    check(lib().svoslam_raycast_model_depth(_ptr(out), w, h, float(fx), float(fy), host,
                                            C.c_void_p(int(cam_to_world_ptr)) if cam_to_world_ptr is not None else C.c_void_p(0),
                                            C.c_void_p(int(pool_ptr)), _fa(center, 3), float(size), _ptr(counters), _stream()))
    return out


def cone_trace_svo_band(out, row_first, rows, fov, view, pool_ptr, center, size, mode=RENDER_REFERENCE, counters=None):
    """rows [row_first, row_first+rows) of the same render into the full-frame buffer `out`."""
    h, w = int(out.shape[0]), int(out.shape[1])
    check(lib().svoslam_cone_trace_svo_band(_ptr(out), w, h, int(row_first), int(rows), float(fov), _fa(view, 16),
                                            C.c_void_p(int(pool_ptr)), _fa(center, 3), float(size), int(mode),
                                            _ptr(counters), _stream()))
    return out


def cone_trace_timing(enable):
    """HIP events around every trace kernel from now on (on its launch stream)"""
    check(lib().svoslam_cone_trace_timing(1 if enable else 0))


def cone_trace_timing_read():
    """(summed kernel ms, launches) since the last read; blocking"""
    ms, n = C.c_float(0), C.c_int32(0)
    check(lib().svoslam_cone_trace_timing_read(C.byref(ms), C.byref(n)))
    return float(ms.value), int(n.value)


(STAGE_MARCH, STAGE_TRACKER, STAGE_FUSE_SORT, STAGE_FUSE_PLAN, STAGE_FUSE_COMMIT, STAGE_MAPS, STAGE_MESH_RASTER, STAGE_MESH_SORT,
 STAGE_MESH_EMIT) = range(9)   # SVOSLAM_STAGE_*
STAGE_NAMES = ("march", "tracker", "fuse_sort", "fuse_plan", "fuse_commit", "maps", "mesh_raster", "mesh_sort", "mesh_emit")


def stage_timing(stages):
    """HIP events around the launches of the given stages (iterable of STAGE_*) from now on; clears every log"""
    mask = 0
    for s in stages:
        mask |= 1 << int(s)
    check(lib().svoslam_stage_timing(mask))


def stage_timing_read(stage):
    """(summed ms, bracketed launches / launch groups) of one stage since the last read; blocking"""
    ms, n = C.c_float(0), C.c_int32(0)
    check(lib().svoslam_stage_timing_read(int(stage), C.byref(ms), C.byref(n)))
    return float(ms.value), int(n.value)


# ----------------------------------------------------------------------------- recorded sensor
def image_load(path):
    """PNG / PGM / PPM -> numpy array [h, w] uint16 or [h, w, 3] uint8 (host side, no device needed)"""
    data, w, h, ch, bits = _vp(), _i32(0), _i32(0), _i32(0), _i32(0)
    check(lib().svoslam_image_load(str(path).encode(), C.byref(data), C.byref(w), C.byref(h), C.byref(ch), C.byref(bits)))
    try:
        n = w.value * h.value * ch.value
        if bits.value == 16:
            a = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint16)), shape=(n,)).copy()
        else:
            a = np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    finally:
        C.CDLL(None).free(data)
    return a.reshape(h.value, w.value) if ch.value == 1 else a.reshape(h.value, w.value, ch.value)


def focal_from_fov(width, height, hfov_rad, vfov_rad):
    fx, fy = C.c_float(0), C.c_float(0)
    check(lib().svoslam_focal_from_fov(width, height, float(hfov_rad), float(vfov_rad), C.byref(fx), C.byref(fy)))
    return float(fx.value), float(fy.value)


class FrameReader:
    """sensor::OpenNIDevice replaced by a TUM-style association list of depth / colour images"""

    def __init__(self, association_file, depth_units_per_metre=1000.0):
        self._h = _vp()
        check(lib().svoslam_frame_reader_open(C.byref(self._h), str(association_file).encode(), float(depth_units_per_metre)))
        w, h, n = _i32(0), _i32(0), _i32(0)
        check(lib().svoslam_frame_reader_info(self._h, C.byref(w), C.byref(h), C.byref(n)))
        self.width, self.height, self.num_frames = w.value, h.value, n.value

    def rewind(self):
        check(lib().svoslam_frame_reader_rewind(self._h))

    def next_host(self):
        """(depth uint16 [h,w] in mm, rgb uint8 [h,w,3], timestamp_us) or None at the end"""
        d = np.empty((self.height, self.width), np.uint16)
        c = np.empty((self.height, self.width, 3), np.uint8)
        ts, got = C.c_longlong(0), _i32(0)
        check(lib().svoslam_frame_reader_next_host(self._h, C.c_void_p(d.ctypes.data), C.c_void_p(c.ctypes.data), C.byref(ts),
                                                   C.byref(got)))
        return (d, c, int(ts.value)) if got.value else None

    def next_device(self, depth_out, rgb_out):
        """uploads the next frame into cuda tensors (int16/uint16 [h,w], uint8 [h,w,3]); returns the timestamp or None"""
        ts, got = C.c_longlong(0), _i32(0)
        check(lib().svoslam_frame_reader_next(self._h, _ptr(depth_out), _ptr(rgb_out), C.byref(ts), C.byref(got), _stream()))
        return int(ts.value) if got.value else None

    def close(self):
        if self._h:
            lib().svoslam_frame_reader_close(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- sensor
def generate_vertex_map(depth, out, fx, fy, img_w, img_h):
    h, w = int(depth.shape[0]), int(depth.shape[1])
    check(lib().svoslam_generate_vertex_map(_ptr(depth), _ptr(out), w, h, float(fx), float(fy), img_w, img_h, _stream()))
    return out


def generate_vertex_map_rows(depth, out, first_row, rows, fx, fy, img_w, img_h):
    h, w = int(depth.shape[0]), int(depth.shape[1])
    check(lib().svoslam_generate_vertex_map_rows(_ptr(depth), _ptr(out), w, h, int(first_row), int(rows), float(fx), float(fy),
                                                 img_w, img_h, _stream()))
    return out


def generate_normal_map(vmap, out):
    h, w = int(vmap.shape[0]), int(vmap.shape[1])
    check(lib().svoslam_generate_normal_map(_ptr(vmap), _ptr(out), w, h, _stream()))
    return out


def bilateral_filter(depth, out):
    h, w = int(depth.shape[0]), int(depth.shape[1])
    check(lib().svoslam_bilateral_filter(_ptr(depth), _ptr(out), w, h, _stream()))
    return out


def subsample_depth(data, tmp, w, h):
    import torch
    fn = lib().svoslam_subsample_depth_u16 if data.dtype in (torch.uint16, torch.int16) else lib().svoslam_subsample_depth_f32
    check(fn(_ptr(data), _ptr(tmp), w, h, _stream()))


def subsample(data, tmp, w, h):
    import torch
    fn = lib().svoslam_subsample_rgb8 if data.dtype == torch.uint8 else lib().svoslam_subsample_f32
    check(fn(_ptr(data), _ptr(tmp), w, h, _stream()))


def color_to_intensity(rgb, out):
    check(lib().svoslam_color_to_intensity(_ptr(rgb), _ptr(out), int(out.numel()), _stream()))
    return out


def transform_vertex_map(v, trans):
    check(lib().svoslam_transform_vertex_map(_ptr(v), _fa(trans, 16), int(v.numel() // 3), _stream()))


def transform_normal_map(v, trans):
    check(lib().svoslam_transform_normal_map(_ptr(v), _fa(trans, 16), int(v.numel() // 3), _stream()))


def transform_vertex_map_dmat(v, d_trans_ptr):
    check(lib().svoslam_transform_vertex_map_dmat(_ptr(v), C.c_void_p(int(d_trans_ptr)), int(v.numel() // 3), _stream()))


def point_cloud_bbox(points, bbox0=(0, 0, 0), bbox1=(0, 0, 0)):
    b0, b1 = _fa(bbox0, 3), _fa(bbox1, 3)
    check(lib().svoslam_point_cloud_bbox(_ptr(points), int(points.numel() // 3), b0, b1, _stream()))
    return np.array(list(b0), np.float32), np.array(list(b1), np.float32)


def point_cloud_bbox_device(ws, points, out7):
    check(lib().svoslam_point_cloud_bbox_device(ws._h, _ptr(points), int(points.numel() // 3), _ptr(out7), _stream()))
    return out7


def gradient(intensity, out):
    """Sobel / 8 of a float image [h, w] into out [h, w, 2] (own specification, see svoslam.h)"""
    h, w = int(intensity.shape[0]), int(intensity.shape[1])
    check(lib().svoslam_gradient(_ptr(intensity), _ptr(out), w, h, _stream()))
    return out


def difference(a, b, out):
    check(lib().svoslam_difference(_ptr(a), _ptr(b), _ptr(out), int(out.numel()), _stream()))
    return out


def rgbd_cost(last_i, last_g, last_v, cur_i, cur_v, fx, fy, img_w, img_h):
    """photometric normal equations (computeRGBDCost, own specification) -> (A [6,6], b [6])"""
    h, w = int(last_i.shape[0]), int(last_i.shape[1])
    A, b = (C.c_float * 36)(), (C.c_float * 6)()
    check(lib().svoslam_rgbd_cost(_ptr(last_i), _ptr(last_g), _ptr(last_v), _ptr(cur_i), _ptr(cur_v), w, h, float(fx), float(fy),
                                  int(img_w), int(img_h), A, b, _stream()))
    return np.array(list(A), np.float32).reshape(6, 6), np.array(list(b), np.float32)


def icp_cost2(last_v, last_n, cur_v, cur_n):
    h, w = int(last_v.shape[0]), int(last_v.shape[1])
    A, b = (C.c_float * 36)(), (C.c_float * 6)()
    check(lib().svoslam_icp_cost2(_ptr(last_v), _ptr(last_n), _ptr(cur_v), _ptr(cur_n), w, h, A, b, _stream()))
    return np.array(list(A), np.float32).reshape(6, 6), np.array(list(b), np.float32)


def icp_cost(last_v, last_n, cur_v, cur_n, A0=None, b0=None):
    """sensor::computeICPCost (correspondence variant).  Returns (A, b, num_correspondences); without correspondences
    A and b keep A0 / b0 (zeros by default), as the reference leaves its outputs untouched."""
    h, w = int(last_v.shape[0]), int(last_v.shape[1])
    A, b = (C.c_float * 36)(), (C.c_float * 6)()
    if A0 is not None:
        A[:] = [float(v) for v in np.asarray(A0, np.float32).reshape(36)]
    if b0 is not None:
        b[:] = [float(v) for v in np.asarray(b0, np.float32).reshape(6)]
    m = _i32(0)
    check(lib().svoslam_icp_cost(_ptr(last_v), _ptr(last_n), _ptr(cur_v), _ptr(cur_n), w, h, A, b, C.byref(m), _stream()))
    return np.array(list(A), np.float32).reshape(6, 6), np.array(list(b), np.float32), int(m.value)


def icp_accumulate(last_v, last_n, cur_v, cur_n, first_pixel, num_pixels, acc):
    h, w = int(last_v.shape[0]), int(last_v.shape[1])
    check(lib().svoslam_icp_accumulate(_ptr(last_v), _ptr(last_n), _ptr(cur_v), _ptr(cur_n), w, h, first_pixel, num_pixels,
                                       _ptr(acc), _stream()))


# ----------------------------------------------------------------------------- tracker
class Camera:
    """Mirror of sensor::RGBDCamera (include/octree_slam/sensor/rgbd_camera.h)."""

    def __init__(self, width, height, fx, fy):
        self._h = C.c_void_p()
        self.width, self.height = width, height
        check(lib().svoslam_camera_create(C.byref(self._h), width, height, float(fx), float(fy)))

    def update(self, depth, rgb, timestamp):
        used = C.c_int32(0)
        check(lib().svoslam_camera_update(self._h, _ptr(depth), _ptr(rgb), int(timestamp), C.byref(used), _stream()))
        return int(used.value)

    def set_rgbd(self, enable=True):
        """photometric RGB-D term in every ICP iteration (rgbd_camera.cpp:126-141 switched on); before the first frame"""
        check(lib().svoslam_camera_set_rgbd(self._h, 1 if enable else 0))

    def set_strict_reference(self, strict=True):
        """False: this build's corrected tracker (include/svoslam.h svoslam_camera_set_strict_reference); before the first frame"""
        check(lib().svoslam_camera_set_strict_reference(self._h, 1 if strict else 0))

    def set_model_depth(self, depth):
        """frame-to-model tracking: `depth` (cuda uint16 [h, w], e.g. raycast_model_depth from the pose of the frame just
        tracked) becomes the map set the ICP tracks against (own specification: include/svoslam.h)"""
        check(lib().svoslam_camera_set_model_depth(self._h, _ptr(depth), _stream()))   # (None: no model, frame to frame until the next one)

    def set_frame_to_model(self, enable=True):
        check(lib().svoslam_camera_set_frame_to_model(self._h, 1 if enable else 0))

    def reset(self):
        """identity pose, no frame seen; buffers and recorded graphs kept"""
        check(lib().svoslam_camera_reset(self._h))

    def prepare(self, depth, rgb, timestamp):
        """first half of update(): bilateral filter + pyramids of the next frame (may run ahead on another stream)"""
        used = C.c_int32(0)
        check(lib().svoslam_camera_prepare(self._h, _ptr(depth), _ptr(rgb), int(timestamp), C.byref(used), _stream()))
        return int(used.value)

    def pair_delta(self, depth_prev, rgb_prev, depth_cur, rgb_cur, out):
        """update_trans of frame `cur` tracked against frame `prev` -> out (DELTA_FLOATS floats); this camera is scratch"""
        check(lib().svoslam_camera_pair_delta(self._h, _ptr(depth_prev), _ptr(rgb_prev), _ptr(depth_cur), _ptr(rgb_cur), _ptr(out),
                                              _stream()))

    def apply_delta(self, delta, timestamp):
        """the pose step of update() for a frame tracked elsewhere (delta = None: pose unchanged)"""
        used = C.c_int32(0)
        check(lib().svoslam_camera_apply_delta(self._h, _ptr(delta), int(timestamp), C.byref(used), _stream()))
        return int(used.value)

    def track_prepared(self):
        """second half of update(): pose of the oldest prepared frame"""
        check(lib().svoslam_camera_track(self._h, _stream()))

    # multi-GPU stepping (all-reduce between accumulate and solve)
    def set_band(self, first_row, rows):
        check(lib().svoslam_camera_set_band(self._h, first_row, rows))

    def set_acc(self, acc_tensor):
        self._acc_keepalive = acc_tensor
        check(lib().svoslam_camera_set_acc(self._h, _ptr(acc_tensor)))

    def begin(self, depth, rgb, timestamp):
        used = C.c_int32(0)
        check(lib().svoslam_camera_begin(self._h, _ptr(depth), _ptr(rgb), int(timestamp), C.byref(used), _stream()))
        return int(used.value)

    def icp_accumulate(self, level, it):
        check(lib().svoslam_camera_icp_accumulate(self._h, level, it, _stream()))

    def icp_solve(self, level, it):
        check(lib().svoslam_camera_icp_solve(self._h, level, it, _stream()))

    def end(self):
        check(lib().svoslam_camera_end(self._h, _stream()))

    def pose(self):
        p, o = (C.c_float * 3)(), (C.c_float * 9)()
        check(lib().svoslam_camera_pose(self._h, p, o, _stream()))
        return np.array(list(p), np.float32), np.array(list(o), np.float32)

    def last_system(self):
        A, b, x = (C.c_float * 36)(), (C.c_float * 6)(), (C.c_float * 6)()
        check(lib().svoslam_camera_last_system(self._h, A, b, x, _stream()))
        return np.array(list(A), np.float32).reshape(6, 6), np.array(list(b), np.float32), np.array(list(x), np.float32)

    def fusion_transform_ptr(self):
        return int(lib().svoslam_camera_fusion_transform_device(self._h))

    def last_vertex_ptr(self, level):
        return int(lib().svoslam_camera_last_vertex(self._h, level))

    def last_normal_ptr(self, level):
        return int(lib().svoslam_camera_last_normal(self._h, level))

    def track_profile(self):
        """device clock stamps [32, 8] of the last one-launch tracker (SVO_TRK_PROF builds; zeros otherwise)"""
        out = np.zeros((32, 8), np.uint64)
        check(lib().svoslam_camera_track_profile(self._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), _stream()))
        return out

    def tracking_lost_count(self):
        n = C.c_int32(0)
        check(lib().svoslam_camera_tracking_lost_count(self._h, C.byref(n), _stream()))
        return int(n.value)

    def close(self):
        if self._h:
            lib().svoslam_camera_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def subtree_file_words(file):
    """the stand-alone linear tree stored in a paged-out sub-tree file, as uint32 words (2 per node)"""
    w, n = C.POINTER(C.c_uint32)(), _i32(0)
    check(lib().svoslam_subtree_file_nodes(str(file).encode(), C.byref(w), C.byref(n)))
    try:
        return np.ctypeslib.as_array(w, shape=(2 * n.value,)).copy()
    finally:
        C.CDLL(None).free(w)


def copy_from_device(ptr, shape, dtype):
    """Copy a raw device buffer to a numpy array (tests)."""
    import torch
    torch.cuda.synchronize()
    out = np.empty(shape, dtype=dtype)
    r = _hip().hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(int(ptr)), C.c_size_t(out.nbytes), 2)
    if r != 0:
        raise SvoslamError("hipMemcpy D2H failed: %d" % r)
    return out
