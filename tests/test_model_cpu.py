"""Frame-to-model tracking (second half; OWN specification -- the reference leaves it as the TODO of
src/sensor/rgbd_camera.cpp:185): properties of the CPU restatement (oracle/svoslam_oracle.c ora_raycast_model_depth,
ora_camera_set_model_depth, ora_camera_set_frame_to_model) that pin what the specification says.  The HIP side is
compared with these functions bit for bit in tests/test_gpu_model.py."""
import importlib

import numpy as np
import pytest

import svoslam_pkg

svoslam_pkg.load()     # registers the package by name; nothing of the HIP library is called here
synth = importlib.import_module("octree_slam_amd.synth")

IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


def wall_map(oracle, w, h, f, z_mm, depth, center, edge, inserts):
    """a wall at z_mm in front of the identity pose, observed `inserts` times (A grows by 2 per observation from 129)"""
    d = np.full((h, w), z_mm, np.uint16)
    v = oracle.vertex_map(d, f, f, w, h).reshape(-1, 3)
    col = np.full((w * h, 3), 200, np.uint8)
    pool = oracle.Pool()
    for _ in range(inserts):
        pool.insert_cloud(v, col, depth, center, edge)
    return pool


def test_model_depth_of_a_saturated_wall_and_of_nothing(oracle):
    w, h, f = 64, 48, 57.0
    center, edge, depth = (0.0, 0.0, 2.0), 4.096, 8
    pool = wall_map(oracle, w, h, f, 2000, depth, center, edge, 64)
    model, steps = oracle.raycast_model_depth(pool, w, h, f, f, IDENTITY, center, edge)
    assert steps > w * h
    # LOD 7 at two metres with this focal length: samples are 32 mm apart and a level-7 node is 64 mm deep
    assert (model > 0).all() and np.abs(model.astype(np.int32) - 2000).max() <= 70
    # half a metre closer: half a metre less (column-major matrix: translation in elements 12..14)
    closer = IDENTITY.copy(); closer[14] = 0.5
    model2, _ = oracle.raycast_model_depth(pool, w, h, f, f, closer, center, edge)
    inner = (slice(12, 36), slice(16, 48))      # (pixels whose rays still meet the observed part of the wall)
    hit = model2[inner] > 0     # (LOD 8 from here: the leaves themselves, and a ray may pass between two observed leaves)
    assert hit.mean() > 0.8 and np.abs(model2[inner][hit].astype(np.int32) - 1500).max() <= 70
    # looking the other way (rotation by pi about y): nothing within range
    back = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32).reshape(16)
    model3, _ = oracle.raycast_model_depth(pool, w, h, f, f, back, center, edge)
    assert (model3 == 0).all()
    # 40 observations: A = 129 + 2 x 39 < 254, nothing retires a ray yet
    young = wall_map(oracle, w, h, f, 2000, depth, center, edge, 40)
    assert (oracle.raycast_model_depth(young, w, h, f, f, IDENTITY, center, edge)[0] == 0).all()
    # an empty map (the root alone)
    assert (oracle.raycast_model_depth(np.zeros(16, np.uint32), w, h, f, f, IDENTITY, center, edge)[0] == 0).all()


def test_model_set_from_the_previous_frame_is_frame_to_frame_tracking(oracle):
    """the hook replaces WHICH maps the ICP associates with and nothing else: fed with the previous frame's own depth
    image it reproduces the reference's frame-to-frame tracker bit for bit"""
    w, h = 96, 72
    f = synth.focal_length(w)
    a, b = oracle.Camera(w, h, f, f), oracle.Camera(w, h, f, f)
    assert b.set_frame_to_model(True) == 0
    prev = None
    for k in range(5):
        d, c = synth.render_frame(4 * k, w, h)
        dn, cn = d.numpy().view(np.uint16), c.numpy()
        if prev is not None:
            assert b.set_model_depth(prev) == 0
        assert a.update(dn, cn, k) == b.update(dn, cn, k) == 1
        pa, oa = a.pose(); pb, ob = b.pose()
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)) and np.array_equal(oa.view(np.uint32), ob.view(np.uint32)), k
        prev = dn
    assert np.abs(oa.reshape(3, 3) - np.eye(3)).max() > 1e-3     # (the camera did move)
    # a model that is NOT the previous frame changes the estimate; without a model the mode is the reference's tracker
    c2 = oracle.Camera(w, h, f, f)
    c2.set_frame_to_model(True)
    for k in range(3):
        d, c = synth.render_frame(4 * k, w, h)
        dn = d.numpy().view(np.uint16)
        if k == 2:
            c2.set_model_depth(first)        # frame 0's depth instead of frame 1's
        c2.update(dn, c.numpy(), k)
        if k == 0:
            first = dn
    a2 = oracle.Camera(w, h, f, f)
    for k in range(3):
        d, c = synth.render_frame(4 * k, w, h)
        a2.update(d.numpy().view(np.uint16), c.numpy(), k)
    assert not np.array_equal(a2.pose()[1], c2.pose()[1])


def test_frame_to_model_and_the_photometric_term_exclude_each_other(oracle):
    cam = oracle.Camera(64, 48, 57.0, 57.0)
    cam.set_rgbd(True)
    assert cam.set_frame_to_model(True) == -1
