"""GPU parity: sensor image kernels, ICP normal equations and the RGBDCamera tracker through the
C ABI vs the CPU oracle.  All of these are bit-exact by construction (IEEE op-by-op float, exact
fixed-point ICP sums); float maps are compared bit for bit with NaN == NaN."""
import numpy as np
import pytest

from util import describe_mismatch, same_bits_or_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    import importlib
    synth = importlib.import_module("octree_slam_amd.synth")
    return pkg, torch, synth


def u16(t):
    return t.cpu().numpy().view(np.uint16)


def dev16(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, np.uint16).view(np.int16)).cuda()


def noisy_depth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    d = 1800 + 600 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + rng.normal(scale=3.0, size=(h, w))
    d[(xx // 40 + yy // 30) % 5 == 0] += 900      # depth discontinuities
    d = np.clip(d, 0, 65535)
    d[rng.random((h, w)) < 0.02] = 0              # dropouts
    d[0:3, 0:5] = 20000                           # > 15000 -> invalid
    return d.astype(np.uint16)


@pytest.mark.parametrize("h,w", [(48, 64), (120, 160), (61, 97), (480, 640)])
def test_bilateral(env, oracle, h, w):
    pkg, torch, _ = env
    rng = np.random.default_rng(h * w)
    d = noisy_depth(rng, h, w)
    out = torch.zeros((h, w), dtype=torch.int16, device="cuda")
    pkg.bilateral_filter(dev16(torch, d), out)
    ref = oracle.bilateral(d)
    assert np.array_equal(u16(out), ref), describe_mismatch(u16(out), ref)


def test_bilateral_constant_and_extremes(env, oracle):
    pkg, torch, _ = env
    for d in (np.full((20, 33), 1234, np.uint16), np.zeros((9, 9), np.uint16),
              np.random.default_rng(1).integers(0, 40000, (37, 41)).astype(np.uint16)):
        out = torch.zeros(d.shape, dtype=torch.int16, device="cuda")
        pkg.bilateral_filter(dev16(torch, d), out)
        assert np.array_equal(u16(out), oracle.bilateral(d))


def test_bilateral_int_wrap_and_overflowed_weights(env, oracle):
    """`(value - depth) * (value - depth)` is an `int` product (image_kernels.cu:168): a 65535 next to a small depth wraps it
    negative, the weight overflows to +inf and the quotient is NaN -> 0 (found by the second reading of the filter,
    tests/test_oracle_second_opinion.py); the device must do the same"""
    pkg, torch, _ = env
    rng = np.random.default_rng(8)
    d = (900 + rng.integers(0, 400, (40, 56))).astype(np.uint16)
    d[rng.random(d.shape) < 0.03] = 65535
    d[rng.random(d.shape) < 0.03] = 0
    d[5, 5], d[5, 6] = 65535, 19195            # difference 46340: the largest whose square still fits; 46341 next to it
    d[20, 20], d[20, 21] = 65535, 19194
    out = torch.zeros(d.shape, dtype=torch.int16, device="cuda")
    pkg.bilateral_filter(dev16(torch, d), out)
    ref = oracle.bilateral(d)
    assert np.array_equal(u16(out), ref), describe_mismatch(u16(out), ref)
    assert (ref == 0).sum() > (d == 0).sum() // 4      # the NaN -> 0 case occurs


@pytest.mark.parametrize("h,w,iw,ih", [(480, 640, 640, 480), (240, 320, 640, 480), (120, 160, 640, 480), (61, 97, 97, 61)])
def test_vertex_normal_maps(env, oracle, h, w, iw, ih):
    pkg, torch, _ = env
    rng = np.random.default_rng(h)
    d = noisy_depth(rng, h, w)
    f = 570.3 * iw / 640.0
    v = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    n = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    pkg.generate_vertex_map(dev16(torch, d), v, f, f, iw, ih)
    pkg.generate_normal_map(v, n)
    rv = oracle.vertex_map(d, f, f, iw, ih)
    rn = oracle.normal_map(rv)
    assert same_bits_or_nan(v.cpu().numpy(), rv)
    assert same_bits_or_nan(n.cpu().numpy(), rn)


def test_pyramid_subsample(env, oracle):
    pkg, torch, _ = env
    rng = np.random.default_rng(4)
    h, w = 120, 160
    d = noisy_depth(rng, h, w)
    buf, tmp = dev16(torch, d), torch.zeros((h // 2 * (w // 2),), dtype=torch.int16, device="cuda")
    pkg.subsample_depth(buf, tmp, w, h)
    ref = oracle.subsample_depth(d)
    assert np.array_equal(u16(buf).reshape(-1)[: ref.size].reshape(ref.shape), ref)
    fl = rng.random((h, w)).astype(np.float32) * 3000
    fb, ft = torch.from_numpy(fl).cuda(), torch.zeros((h // 2 * (w // 2),), dtype=torch.float32, device="cuda")
    pkg.subsample_depth(fb, ft, w, h)
    assert np.array_equal(fb.cpu().numpy().reshape(-1)[: ref.size].reshape(ref.shape), oracle.subsample_depth(fl))
    fb2 = torch.from_numpy(fl).cuda()
    pkg.subsample(fb2, ft, w, h)
    assert np.array_equal(fb2.cpu().numpy().reshape(-1)[: ref.size].reshape(ref.shape), oracle.subsample(fl))
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rb, rt = torch.from_numpy(rgb).cuda(), torch.zeros((h // 2 * (w // 2) * 3,), dtype=torch.uint8, device="cuda")
    pkg.subsample(rb, rt, w, h)
    assert np.array_equal(rb.cpu().numpy().reshape(-1)[: ref.size * 3].reshape(ref.shape + (3,)), oracle.subsample(rgb))
    inten = torch.zeros((h * w,), dtype=torch.float32, device="cuda")
    pkg.color_to_intensity(torch.from_numpy(rgb).cuda(), inten)
    assert np.array_equal(inten.cpu().numpy(), oracle.color_to_intensity(rgb))


def test_transforms_and_bbox(env, oracle):
    pkg, torch, _ = env
    rng = np.random.default_rng(6)
    d = noisy_depth(rng, 96, 128)
    v = oracle.vertex_map(d, 114.06, 114.06, 128, 96)
    T = oracle.icp_update_transform(np.array([0.01, -0.02, 0.015, 0.03, -0.01, 0.02], np.float32))
    tv = torch.from_numpy(v.copy()).cuda()
    pkg.transform_vertex_map(tv, T)
    assert same_bits_or_nan(tv.cpu().numpy(), oracle.transform_vertex_map(v, T))
    tn = torch.from_numpy(v.copy()).cuda()
    pkg.transform_normal_map(tn, T)
    assert same_bits_or_nan(tn.cpu().numpy(), oracle.transform_normal_map(v, T))
    pts = tv.cpu().numpy().reshape(-1, 3)
    b0, b1 = pkg.point_cloud_bbox(tv)
    r0, r1 = oracle.point_cloud_bbox(pts)
    assert np.array_equal(b0, r0) and np.array_equal(b1, r1)
    b0, b1 = pkg.point_cloud_bbox(tv, (-9, 0.1, 0.2), (0.3, 50, 0.4))
    r0, r1 = oracle.point_cloud_bbox(pts, (-9, 0.1, 0.2), (0.3, 50, 0.4))
    assert np.array_equal(b0, r0) and np.array_equal(b1, r1)
    allnan = torch.full((50, 3), float("nan"), device="cuda")
    b0, b1 = pkg.point_cloud_bbox(allnan)
    assert (b0 == 0).all() and (b1 == 0).all()


@pytest.mark.parametrize("h,w", [(120, 160), (240, 320), (480, 640), (75, 101)])
def test_icp_cost2(env, oracle, h, w):
    pkg, torch, _ = env
    rng = np.random.default_rng(w)
    f = 570.3 * w / 640.0
    d1 = noisy_depth(rng, h, w)
    v1 = oracle.vertex_map(d1, f, f, w, h); n1 = oracle.normal_map(v1)
    T = oracle.icp_update_transform(np.array([0.004, -0.003, 0.002, 0.004, -0.002, 0.003], np.float32))
    v2 = oracle.transform_vertex_map(v1, T); n2 = oracle.transform_normal_map(n1, T)
    A, b = pkg.icp_cost2(*(torch.from_numpy(x).cuda() for x in (v1, n1, v2, n2)))
    rA, rb = oracle.icp_cost2(v1, n1, v2, n2)
    assert np.array_equal(A, rA) and np.array_equal(b, rb), (A - rA, b - rb)
    assert np.abs(A).max() > 0
    # band partials add up exactly (what the multi-GPU all-reduce relies on)
    acc = torch.zeros(27, dtype=torch.float64, device="cuda")
    tens = [torch.from_numpy(x).cuda() for x in (v1, n1, v2, n2)]
    rows = [0, h // 3, h // 2 + 1, h]
    for r0, r1 in zip(rows[:-1], rows[1:]):
        pkg.icp_accumulate(*tens, r0 * w, (r1 - r0) * w, acc)
    raw = oracle.icp_cost2_raw(v1, n1, v2, n2)
    assert np.array_equal(acc.cpu().numpy(), raw.astype(np.float64))


@pytest.mark.parametrize("h,w", [(120, 160), (480, 640), (75, 101)])
def test_icp_cost_correspondence_variant(env, oracle, h, w):
    """computeICPCost (a19): stencil gates without the depth-range gate, order-preserving compaction, load size 10 with
    the floor(M/10) reduce; untouched outputs when nothing corresponds"""
    pkg, torch, _ = env
    rng = np.random.default_rng(h)
    f = 570.3 * w / 640.0
    d1 = noisy_depth(rng, h, w)
    v1 = oracle.vertex_map(d1, f, f, w, h); n1 = oracle.normal_map(v1)
    T = oracle.icp_update_transform(np.array([0.004, -0.003, 0.002, 0.004, -0.002, 0.003], np.float32))
    v2 = oracle.transform_vertex_map(v1, T); n2 = oracle.transform_normal_map(n1, T)
    tens = [torch.from_numpy(x).cuda() for x in (v1, n1, v2, n2)]
    A, b, m = pkg.icp_cost(*tens)
    rA, rb, rm = oracle.icp_cost(v1, n1, v2, n2)
    assert m == rm and m > 100 and m % 10 != 0 or m == rm
    assert np.array_equal(A, rA) and np.array_equal(b, rb), (A - rA, b - rb)
    assert np.abs(A).max() > 0
    A2, b2 = pkg.icp_cost2(*tens)
    assert not np.array_equal(A, A2)      # the two variants gate and truncate differently
    # fewer than 10 correspondences: zeros (empty reduce); none: outputs untouched
    few_v2 = np.full_like(v2, np.nan); few_n2 = n2.copy()
    ys, xs = np.nonzero(np.isfinite(v2).all(-1) & np.isfinite(n2).all(-1) & np.isfinite(n1).all(-1))
    for k in range(7):
        few_v2[ys[k * 50], xs[k * 50]] = v2[ys[k * 50], xs[k * 50]]
    A3, b3, m3 = pkg.icp_cost(tens[0], tens[1], torch.from_numpy(few_v2).cuda(), tens[3], A0=np.full(36, 5.0), b0=np.full(6, 7.0))
    rA3, rb3, rm3 = oracle.icp_cost(v1, n1, few_v2, n2, A0=np.full(36, 5.0), b0=np.full(6, 7.0))
    assert m3 == rm3 and 0 < m3 < 10 and np.array_equal(A3, rA3) and (A3 == 0).all() and (b3 == 0).all()
    none_v2 = np.full_like(v2, np.nan)
    A4, b4, m4 = pkg.icp_cost(tens[0], tens[1], torch.from_numpy(none_v2).cuda(), tens[3], A0=np.full(36, 5.0), b0=np.full(6, 7.0))
    assert m4 == 0 and (A4 == 5.0).all() and (b4 == 7.0).all()
    assert oracle.icp_cost(v1, n1, none_v2, n2, A0=np.full(36, 5.0), b0=np.full(6, 7.0))[2] == 0


def test_camera_tracker_matches_oracle(env, oracle):
    pkg, torch, synth = env
    w, h = 160, 120
    f = synth.focal_length(w)
    cam = pkg.Camera(w, h, f, f)
    ocam = oracle.Camera(w, h, f, f)
    for k in range(5):
        d, c = synth.render_frame(3 * k, w, h)       # 0.3 deg / 2.6 mm between frames
        dn = d.numpy().view(np.uint16)
        used = cam.update(d.cuda(), c.cuda(), k)
        assert used == ocam.update(dn, c.numpy(), k) == 1
        p, o = cam.pose(); rp, ro = ocam.pose()
        assert np.array_equal(p.view(np.uint32), rp.view(np.uint32)), (k, p, rp)
        assert np.array_equal(o.view(np.uint32), ro.view(np.uint32)), (k, o, ro)
        if k >= 1:
            A, b, x = cam.last_system(); rA, rb, rx = ocam.last_system()
            assert np.array_equal(A, rA) and np.array_equal(b, rb) and np.array_equal(x, rx)
        fus = pkg.copy_from_device(cam.fusion_transform_ptr(), (16,), np.float32)
        assert np.array_equal(fus, ocam.fusion_transform())
    assert cam.update(d.cuda(), c.cuda(), 2) == 0 == ocam.update(dn, c.numpy(), 2)  # stale timestamp skipped
    assert cam.tracking_lost_count() == 0
    # the tracker follows the ground-truth yaw (3 frames x 0.3 deg ... sign/axis by convention: just non-trivial)
    p, o = cam.pose()
    assert np.abs(o.reshape(3, 3) - np.eye(3)).max() > 1e-3


@pytest.mark.parametrize("w,h,nframes", [(160, 120, 3), (640, 480, 10)])
def test_camera_split_stepping_equals_update(env, oracle, w, h, nframes):
    """begin / accumulate(bands) / solve / end == update (the multi-GPU stepping API)"""
    pkg, torch, synth = env
    f = synth.focal_length(w)
    cam_a, cam_b = pkg.Camera(w, h, f, f), pkg.Camera(w, h, f, f)
    acc = torch.zeros(27, dtype=torch.float64, device="cuda")
    cam_b.set_acc(acc)
    for k in range(nframes):
        d, c = synth.render_frame(2 * k, w, h)
        cam_a.update(d.cuda(), c.cuda(), k)
        assert cam_b.begin(d.cuda(), c.cuda(), k) == 1
        for level in (2, 1, 0):
            for it in range(pkg.PYRAMID_ITERS[level]):
                for r0, r1 in ((0, h // 3), (h // 3, h)):     # two row bands into one accumulator
                    cam_b.set_band(r0, r1 - r0)
                    cam_b.icp_accumulate(level, it)
                cam_b.icp_solve(level, it)
        cam_b.end()
        pa, oa = cam_a.pose(); pb, ob = cam_b.pose()
        assert np.array_equal(pa, pb) and np.array_equal(oa, ob)


def test_camera_tracker_full_size_matches_oracle_and_is_repeatable(env, oracle):
    """640x480 (BASELINE config 3 size), 24 frames = 437 ICP iterations: pose, A, b, x bit-identical to the
    oracle on every frame, and a second device run gives identical bits (no cross-workgroup visibility race)."""
    pkg, torch, synth = env
    w, h = 640, 480
    f = synth.focal_length(w)
    frames = [synth.render_frame(k, w, h, device="cuda") for k in range(24)]
    ocam = oracle.Camera(w, h, f, f)
    runs = []
    for rep in range(2):
        cam = pkg.Camera(w, h, f, f)
        poses = []
        for k, (d, c) in enumerate(frames):
            cam.update(d, c, k)
            p, o = cam.pose()
            poses.append(o.copy())
            if rep == 0:
                ocam.update(d.cpu().numpy().view(np.uint16), c.cpu().numpy(), k)
                rp, ro = ocam.pose()
                assert np.array_equal(o.view(np.uint32), ro.view(np.uint32)), (k, o, ro)
                if k >= 1:
                    A, b, x = cam.last_system(); rA, rb, rx = ocam.last_system()
                    assert np.array_equal(A, rA) and np.array_equal(b, rb) and np.array_equal(x.view(np.uint32), rx.view(np.uint32))
        runs.append((np.stack(poses), cam.tracking_lost_count()))
    assert np.array_equal(runs[0][0].view(np.uint32), runs[1][0].view(np.uint32))
    assert runs[0][1] == runs[1][1]


def test_camera_tracking_lost_matches_oracle(env, oracle):
    """'Camera tracking is lost' (rgbd_camera.cpp:148-151): a frame without valid depth has no correspondences, the
    6x6 system is zero, Cholesky divides 0 by 0 and every pyramid level is abandoned on its first iteration; a frame
    with depth only in one corner loses some levels only.  Pose, last system, counters and the frames after the
    loss must match the oracle bit for bit (NaN == NaN); also through the stepping API."""
    pkg, torch, synth = env
    w, h = 160, 120
    f = synth.focal_length(w)
    cam, cam_s, ocam = pkg.Camera(w, h, f, f), pkg.Camera(w, h, f, f), oracle.Camera(w, h, f, f)
    acc = torch.zeros(27, dtype=torch.float64, device="cuda")
    cam_s.set_acc(acc)
    lost_seen = 0
    for k in range(7):
        d, c = synth.render_frame(3 * k, w, h)
        d = d.clone()
        if k == 2:
            d.zero_()                                   # nothing valid
        if k == 4:
            d[:, 12:] = 0                               # a sliver of the image: too little for the fine levels or all
            d[40:, :] = 0
        dn = d.numpy().view(np.uint16)
        assert cam.update(d.cuda(), c.cuda(), k) == ocam.update(dn, c.numpy(), k) == 1
        assert cam_s.begin(d.cuda(), c.cuda(), k) == 1
        for level in (2, 1, 0):
            for it in range(pkg.PYRAMID_ITERS[level]):
                cam_s.icp_accumulate(level, it)
                cam_s.icp_solve(level, it)
        cam_s.end()
        for cc in (cam, cam_s):
            p, o = cc.pose(); rp, ro = ocam.pose()
            assert np.array_equal(p.view(np.uint32), rp.view(np.uint32)), (k, p, rp)
            assert np.array_equal(o.view(np.uint32), ro.view(np.uint32)), (k, o, ro)
            assert cc.tracking_lost_count() == ocam.tracking_lost_count(), k
            if k >= 1:
                A, b, x = cc.last_system(); rA, rb, rx = ocam.last_system()
                # NaN == NaN: the sign / payload of a generated NaN is the FPU's choice (x86 host code in the reference)
                assert np.array_equal(A, rA, equal_nan=True) and np.array_equal(b, rb, equal_nan=True)
                assert np.array_equal(x, rx, equal_nan=True), (k, x, rx)
            fus = pkg.copy_from_device(cc.fusion_transform_ptr(), (16,), np.float32)
            assert np.array_equal(fus.view(np.uint32), ocam.fusion_transform().view(np.uint32))
        lost_seen = ocam.tracking_lost_count()
    assert lost_seen >= 6                                # frames 2 and 3 (no valid partner) lose all three levels each


def test_timer_start_stop(env):
    """startTiming / stopTiming (timing_utils.cu:11-32): a HIP-event pair around stream work; the elapsed time
    covers the kernels enqueued between the two calls"""
    pkg, torch, _ = env
    import ctypes as C
    L = pkg.lib()
    x = torch.zeros((1080, 1920), dtype=torch.int16, device="cuda")
    out = torch.zeros_like(x)
    torch.cuda.synchronize()
    ms = C.c_float(-1.0)
    pkg.check(L.svoslam_timer_start(pkg._stream()))
    pkg.check(L.svoslam_timer_stop(pkg._stream(), C.byref(ms)))
    empty = ms.value
    assert 0.0 <= empty < 5.0
    pkg.check(L.svoslam_timer_start(pkg._stream()))
    for _ in range(20):
        pkg.bilateral_filter(x, out)
    pkg.check(L.svoslam_timer_stop(pkg._stream(), C.byref(ms)))
    assert ms.value > empty and ms.value > 0.05          # 20 bilateral passes over 2 M pixels
    assert ms.value < 1000.0
    assert L.svoslam_timer_stop(pkg._stream(), None) != 0     # null output pointer is refused


def test_one_launch_tracker_streaming_mode_in_subprocess(oracle):
    """the one-launch tracker's fallback for levels that do not fit the registers (pixels re-read every iteration, chain
    replayed from LDS) is taken by default only above 640x480-class images, where the launch chain is used instead; forced
    here (4 workers -> 10 pixels per lane at 160x120) in a child process, against the oracle's poses"""
    import json
    import subprocess
    import sys
    import os
    import importlib
    import svoslam_pkg
    svoslam_pkg.load()
    synth = importlib.import_module("octree_slam_amd.synth")
    w, h, n = 160, 120, 5
    f = synth.focal_length(w)
    ocam = oracle.Camera(w, h, f, f)
    want = []
    for k in range(n):
        d, c = synth.render_frame(2 * k, w, h)
        ocam.update(d.numpy().view(np.uint16), c.numpy(), k)
        p, o = ocam.pose()
        want.append([p.view(np.uint32).tolist(), o.view(np.uint32).tolist()])
    code = r'''
import sys, json, importlib, numpy as np
sys.path.insert(0, %r)
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
w, h, n = 160, 120, 5
f = synth.focal_length(w)
cam = pkg.Camera(w, h, f, f)
out = []
for k in range(n):
    d, c = synth.render_frame(2 * k, w, h)      # CPU generator: the frames the parent gave the oracle
    cam.update(d.cuda(), c.cuda(), k)
    p, o = cam.pose()
    out.append([p.view(np.uint32).tolist(), o.view(np.uint32).tolist()])
print("RESULT" + json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVOSLAM_CONFIG="track_mode=2,track_workers=4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0]
    assert json.loads(line[6:]) == want
