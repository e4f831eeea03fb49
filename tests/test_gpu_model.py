"""GPU parity of frame-to-model tracking (second half; OWN specification, include/svoslam.h
svoslam_raycast_model_depth / svoslam_camera_set_model_depth / svoslam_camera_set_frame_to_model) against its CPU
restatement in the oracle (whose properties tests/test_model_cpu.py pins): the model depth image bit for bit from several
poses, the tracker fed with model maps bit for bit, and whole frames of pipeline.SlamPipeline(frame_to_model=True)."""
import importlib

import numpy as np
import pytest

from util import describe_mismatch, surface_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    synth = importlib.import_module("octree_slam_amd.synth")
    pl = importlib.import_module("octree_slam_amd.pipeline")
    return pkg, torch, synth, pl


def u16(t):
    return t.cpu().numpy().view(np.uint16)


def rigid(oracle, eye, target):
    """camera-to-world (column-major, 16 floats) of a sensor at `eye` whose optical axis (+z) points at `target`"""
    e, t = np.asarray(eye, np.float64), np.asarray(target, np.float64)
    fwd = (t - e) / np.linalg.norm(t - e)
    right = np.cross((0.0, 1.0, 0.0), fwd); right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    m = np.zeros(16, np.float32)
    m[0:3], m[4:7], m[8:11], m[12:15], m[15] = right, up, fwd, e, 1.0
    return m


@pytest.mark.parametrize("depth,w,h", [(9, 160, 120), (11, 96, 72)])
def test_model_depth_matches_oracle(env, oracle, depth, w, h):
    pkg, torch, synth, pl = env
    rng = np.random.default_rng(depth)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    center, edge = (0.0, 0.0, 0.0), 1.0
    pts, col = surface_cloud(rng, 30000, jitter=0.001)
    for it in range(66):      # A = 129 + 2 x 65 >= 254 on every observed leaf (and on its ancestors: the mip keeps the maximum)
        pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), depth, pool, center, edge)
        opool.insert_cloud(pts, col, depth, center, edge)
    assert np.array_equal(pool.words()[:2 * pool.size], opool.words()[:2 * pool.size])
    f = 570.3 * w / 640.0
    out = torch.full((h, w), 7, dtype=torch.int16, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    seen = 0
    total = 0
    for eye, tgt in (((0.1, 0.2, -2.6), (0, 0, 0)), ((0.3, 0.1, 1.4), (0.2, 0.0, 0.2)), ((0.12, -0.18, 0.3), (0.1, -0.2, -0.3)),
                     ((3.0, 0.0, 0.0), (4.0, 0.0, 0.0))):     # the last one stands outside the cube and looks away: like coneTrace,
                                                              # the walk has no bounds check -- samples alias into the boundary nodes
        m = rigid(oracle, eye, tgt)
        ref, steps = oracle.raycast_model_depth(opool, w, h, f, f, m, center, edge)
        pkg.raycast_model_depth(out, f, f, pool.data_ptr, center, edge, cam_to_world=m, counters=cnt)
        got = u16(out)
        assert np.array_equal(got, ref), (eye, describe_mismatch(got, ref))
        total += steps
        assert int(cnt.item()) == total
        # the pose as a device pointer (what Camera.fusion_transform_ptr() hands over)
        md = torch.from_numpy(m).cuda()
        out.fill_(9)
        pkg.raycast_model_depth(out, f, f, pool.data_ptr, center, edge, cam_to_world_ptr=md.data_ptr())
        assert np.array_equal(u16(out), ref)
        seen += int((ref > 0).sum())
    assert seen > 1000
    with pytest.raises(Exception):
        pkg.raycast_model_depth(out, f, f, pool.data_ptr, center, edge, cam_to_world=m, cam_to_world_ptr=md.data_ptr())


def test_tracker_with_model_maps_matches_oracle(env, oracle):
    """the camera's hook alone: arbitrary depth images as the model (here: the frame two back), HIP == oracle bit for bit;
    and fed with the previous frame's depth the mode IS the frame-to-frame tracker"""
    pkg, torch, synth, pl = env
    w, h = 160, 120
    f = synth.focal_length(w)
    cam, ocam = pkg.Camera(w, h, f, f), oracle.Camera(w, h, f, f)
    ref_cam = pkg.Camera(w, h, f, f)     # frame-to-frame
    prev_cam = pkg.Camera(w, h, f, f)    # frame-to-model fed with the previous frame
    cam.set_frame_to_model(True); ocam.set_frame_to_model(True); prev_cam.set_frame_to_model(True)
    frames = []
    for k in range(6):
        d, c = synth.render_frame(3 * k, w, h)
        dn = d.numpy().view(np.uint16)
        frames.append(d)
        if k >= 2:
            cam.set_model_depth(frames[k - 2].cuda())
            assert ocam.set_model_depth(frames[k - 2].numpy().view(np.uint16)) == 0
        if k >= 1:
            prev_cam.set_model_depth(frames[k - 1].cuda())
        assert cam.update(d.cuda(), c.cuda(), k) == ocam.update(dn, c.numpy(), k) == 1
        ref_cam.update(d.cuda(), c.cuda(), k); prev_cam.update(d.cuda(), c.cuda(), k)
        p, o = cam.pose(); rp, ro = ocam.pose()
        assert np.array_equal(p.view(np.uint32), rp.view(np.uint32)) and np.array_equal(o.view(np.uint32), ro.view(np.uint32)), k
        if k >= 1:
            A, b, x = cam.last_system(); rA, rb, rx = ocam.last_system()
            assert np.array_equal(A, rA) and np.array_equal(b, rb) and np.array_equal(x, rx)
        p1, o1 = ref_cam.pose(); p2, o2 = prev_cam.pose()
        assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32)) and np.array_equal(p1.view(np.uint32), p2.view(np.uint32))
    assert not np.array_equal(cam.pose()[1], ref_cam.pose()[1])      # (two frames back is a different estimate)
    # after a reset the model is gone (the mode stays): the first frames track frame to frame again
    cam.reset(); ref_cam.reset()
    for k in range(3):
        d, c = synth.render_frame(3 * k, w, h)
        cam.update(d.cuda(), c.cuda(), k); ref_cam.update(d.cuda(), c.cuda(), k)
    assert np.array_equal(cam.pose()[1], ref_cam.pose()[1])
    rg = pkg.Camera(w, h, f, f)
    rg.set_rgbd(True)
    with pytest.raises(Exception):
        rg.set_frame_to_model(True)


def test_frame_to_model_frames_match_oracle(env, oracle):
    """whole frames: the first frame observed 64 times (a saturated map to stand on), then every frame is tracked against the
    map ray-cast from the previous frame's pose, fused with its own pose and the model refreshed: poses, model images, pools
    and renders equal the same loop assembled from oracle calls"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 160, 120, 8, (0.0, 1.5, 0.0), 4.096
    P = pl.SlamPipeline(w, h, depth, center, edge, frame_to_model=True, count_steps=True)
    ocam, opool = oracle.Camera(w, h, P.focal, P.focal), oracle.Pool()
    ocam.set_frame_to_model(True)
    f = P.focal

    def oracle_fuse(dn, cn, times=1):
        v = oracle.transform_vertex_map(oracle.vertex_map(dn, f, f, w, h), ocam.fusion_transform())
        for _ in range(times):
            opool.insert_cloud(v.reshape(-1, 3), cn.reshape(-1, 3), depth, center, edge)

    d, c = synth.render_frame(0, w, h)
    dn, cn = d.numpy().view(np.uint16), c.numpy()
    dg, cg = d.cuda(), c.cuda()
    P.track(dg, cg, 0); ocam.update(dn, cn, 0)
    P.backproject(dg)
    for _ in range(64):
        P.fuse(cg)
    oracle_fuse(dn, cn, 64)
    model = u16(P.refresh_model())
    omodel, steps = oracle.raycast_model_depth(opool, w, h, f, f, ocam.fusion_transform(), center, edge)
    assert np.array_equal(model, omodel), describe_mismatch(model, omodel)
    assert int(P.model_steps.item()) == steps
    assert (omodel > 0).mean() > 0.5                       # most of the first view is in the map
    valid = (omodel > 0) & (dn > 0)
    assert np.median(np.abs(omodel[valid].astype(np.int32) - dn[valid].astype(np.int32))) < 80   # (LOD 7-8: 32-64 mm samples)
    ocam.set_model_depth(omodel)
    for k in range(1, 5):
        d, c = synth.render_frame(2 * k, w, h)
        dn, cn = d.numpy().view(np.uint16), c.numpy()
        view = pl.ground_truth_view(2 * k, synth)
        img = P.frame(d.cuda(), c.cuda(), k, view).cpu().numpy()
        assert ocam.update(dn, cn, k) == 1
        oracle_fuse(dn, cn)
        omodel, _ = oracle.raycast_model_depth(opool, w, h, f, f, ocam.fusion_transform(), center, edge)
        ocam.set_model_depth(omodel if (omodel > 0).sum() >= 0.5 * w * h else None)     # the pipeline's acceptance rule
        rimg, _, _ = oracle.cone_trace(opool, w, h, 45.0, view, center, edge, 0)
        p, o = P.cam.pose(); rp, ro = ocam.pose()
        assert np.array_equal(p.view(np.uint32), rp.view(np.uint32)) and np.array_equal(o.view(np.uint32), ro.view(np.uint32)), k
        assert np.array_equal(u16(P.model_depth), omodel), (k, describe_mismatch(u16(P.model_depth), omodel))
        gw, cw = P.pool.words(), opool.words()
        assert P.pool.size == opool.size and np.array_equal(gw, cw), (k, describe_mismatch(gw, cw))
        assert np.array_equal(img, rimg), (k, describe_mismatch(img, rimg))
    assert P.cam.tracking_lost_count() == 0 == ocam.tracking_lost_count()
    assert P.model_used == 5
    # the same stream frame to frame gives another trajectory (the mode does something)
    Q = pl.SlamPipeline(w, h, depth, center, edge)
    for k in range(5):
        d, c = synth.render_frame(2 * k, w, h)
        Q.track(d.cuda(), c.cuda(), k)
    assert not np.array_equal(Q.cam.pose()[1], P.cam.pose()[1])


def test_young_map_has_no_model_and_tracks_frame_to_frame(env, oracle):
    """a node answers a model ray only after ~64 observations: on a fresh map the model image is empty, the pipeline
    does not accept it, and the frames are tracked exactly as without the mode"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 160, 120, 8, (0.0, 1.5, 0.0), 4.096
    A = pl.SlamPipeline(w, h, depth, center, edge, frame_to_model=True)
    B = pl.SlamPipeline(w, h, depth, center, edge)
    for k in range(4):
        d, c = synth.render_frame(2 * k, w, h, device="cuda")
        view = pl.ground_truth_view(2 * k, synth)
        ia = A.frame(d, c, k, view).cpu().numpy()
        ib = B.frame(d, c, k, view).cpu().numpy()
        assert np.array_equal(ia, ib)
        assert np.array_equal(A.cam.pose()[1], B.cam.pose()[1])
        assert int((A.model_depth != 0).sum().item()) < 0.5 * w * h
    assert A.model_used == 0
    assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words())
    # a model given by hand and taken away again (None): back to frame-to-frame
    cam, ref = pkg.Camera(w, h, A.focal, A.focal), pkg.Camera(w, h, A.focal, A.focal)
    cam.set_frame_to_model(True)
    for k in range(4):
        d, c = synth.render_frame(3 * k, w, h, device="cuda")
        if k == 1:
            cam.set_model_depth(d)
            cam.set_model_depth(None)
        cam.update(d, c, k); ref.update(d, c, k)
    assert np.array_equal(cam.pose()[1], ref.pose()[1])


def test_model_depth_full_size_cfg3_geometry(env, oracle):
    """BASELINE config 3's geometry (640x480, depth-12 map, half-edge 4.096 m about (0, 1.5, 0)): the first frame observed 64
    times, the model image from its pose and from the pose three frames on, against the oracle bit for bit"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 640, 480, 12, (0.0, 1.5, 0.0), 4.096
    P = pl.SlamPipeline(w, h, depth, center, edge, frame_to_model=True, count_steps=True, pool_capacity_nodes=1 << 24)
    f = P.focal
    ocam, opool = oracle.Camera(w, h, f, f), oracle.Pool()
    ocam.set_frame_to_model(True)
    d, c = synth.render_frame(0, w, h)
    dn, cn = d.numpy().view(np.uint16), c.numpy()
    P.track(d.cuda(), c.cuda(), 0); ocam.update(dn, cn, 0)
    P.backproject(d.cuda())
    v = oracle.transform_vertex_map(oracle.vertex_map(dn, f, f, w, h), ocam.fusion_transform())
    for _ in range(64):
        P.fuse(c.cuda())
        opool.insert_cloud(v.reshape(-1, 3), cn.reshape(-1, 3), depth, center, edge)
    assert P.pool.size == opool.size and np.array_equal(P.pool.words(), opool.words())
    total = 0
    for k in (0, 3):
        if k:
            d, c = synth.render_frame(k, w, h)
            P.track(d.cuda(), c.cuda(), k); ocam.update(d.numpy().view(np.uint16), c.numpy(), k)
            p, o = P.cam.pose(); rp, ro = ocam.pose()      # (tracked against the model of frame 0 on both sides)
            assert np.array_equal(o.view(np.uint32), ro.view(np.uint32)) and np.array_equal(p.view(np.uint32), rp.view(np.uint32))
        model = u16(P.refresh_model())
        omodel, steps = oracle.raycast_model_depth(opool, w, h, f, f, ocam.fusion_transform(), center, edge)
        total += steps
        assert np.array_equal(model, omodel), (k, describe_mismatch(model, omodel))
        assert int(P.model_steps.item()) == total
        ocam.set_model_depth(omodel if (omodel > 0).sum() >= 0.5 * w * h else None)
    valid = (omodel > 0)
    assert valid.mean() > 0.5


def test_native_model_loop_equals_the_pipeline_loop(env, oracle):
    """svoslam_runner_run_model (csrc/runner.hip: frame-to-model tracking inside the library's frame loop) against
    SlamPipeline.frame() -- the loop the test above checks against the oracle call by call -- from the same saturated starting map:
    poses, pools, accepted models and the last image equal bit for bit; and on a young map (no model accepted) it is the plain loop"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 160, 120, 8, (0.0, 1.5, 0.0), 4.096

    def start():
        P = pl.SlamPipeline(w, h, depth, center, edge, frame_to_model=True)
        d, c = synth.render_frame(0, w, h, device="cuda")
        P.track(d, c, 0)
        P.backproject(d)
        for _ in range(64):
            P.fuse(c)
        P.refresh_model()
        return P

    A, B = start(), start()
    assert A.model_used == 1 and np.array_equal(A.pool.words(), B.pool.words())
    frames = [synth.render_frame(2 * k, w, h, device="cuda") for k in range(1, 5)]
    views = [pl.ground_truth_view(2 * k, synth) for k in range(1, 5)]
    for k, ((d, c), view) in enumerate(zip(frames, views), start=1):
        img_a = A.frame(d, c, k, view)
    runner = pkg.Runner(B.cam, B.pool, w, h, depth, center, edge, B.focal, B.focal, B.mode)
    used = runner.run_model([d for d, _ in frames], [c for _, c in frames], [1, 2, 3, 4], views, B.image, 0, h, min_coverage=0.5)
    torch.cuda.synchronize()
    assert used == A.model_used - 1 == 4
    pa, oa = A.cam.pose(); pb, ob = B.cam.pose()
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)) and np.array_equal(oa.view(np.uint32), ob.view(np.uint32))
    assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words())
    assert np.array_equal(img_a.cpu().numpy(), B.image.cpu().numpy())
    # young map: nothing answers a model ray, no model is accepted, the frames are tracked frame to frame
    Y = pl.SlamPipeline(w, h, depth, center, edge)
    Z = pl.SlamPipeline(w, h, depth, center, edge)
    fr = [synth.render_frame(2 * k, w, h, device="cuda") for k in range(4)]
    vw = [pl.ground_truth_view(2 * k, synth) for k in range(4)]
    for k, ((d, c), view) in enumerate(zip(fr, vw)):
        img_y = Y.frame(d, c, k, view)
    rz = pkg.Runner(Z.cam, Z.pool, w, h, depth, center, edge, Z.focal, Z.focal, Z.mode)
    assert rz.run_model([d for d, _ in fr], [c for _, c in fr], [0, 1, 2, 3], vw, Z.image, 0, h) == 0
    torch.cuda.synchronize()
    assert np.array_equal(Y.cam.pose()[1], Z.cam.pose()[1]) and np.array_equal(Y.pool.words(), Z.pool.words())
    assert np.array_equal(img_y.cpu().numpy(), Z.image.cpu().numpy())
    with pytest.raises(pkg.SvoslamError):
        rz.run_model([fr[0][0]], [fr[0][1]], [3], vw[:1], Z.image, 0, h)      # a timestamp the camera has seen
