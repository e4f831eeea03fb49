"""GPU tests of the library's state handling: per-stream raycast acceleration data,
pool replacement (set_words) followed by asynchronous fusion, pool replicas (svoslam_pool_copy + svoslam_svo_fuse_commit_to,
what the frame scheduler uses to march one replica while the next frame is committed to the other), and the runner's
argument validation."""
import importlib
import os

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


def build_pool(pkg, torch, seed, depth, frames=2, n=40000):
    rng = np.random.default_rng(seed)
    ws, pool = pkg.Workspace(), pkg.Pool(1 << 16)
    for k in range(frames):
        pts, col = surface_cloud(rng, n)
        pkg.svo_from_point_cloud(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), depth, pool, (0, 0, 0), 1.0)
    return pool


def test_two_pools_rendered_concurrently_on_two_streams(env, oracle):
    """two different maps ray-marched at the same time on two streams (each render rebuilds its level grid and tables in
    the per-stream acceleration buffer) == the same renders one after the other; repeated to give a race a chance"""
    pkg, torch, synth, pl = env
    pools = [build_pool(pkg, torch, 11, 8), build_pool(pkg, torch, 12, 9, frames=3)]
    views = [oracle.look_at((0.1, 0.2, -2.6), (0, 0, 0), (0, 1, 0)), oracle.look_at((1.9, 0.5, 1.2), (0, 0.1, 0), (0, 1, 0))]
    w, h = 320, 240
    ref = []
    for p, v in zip(pools, views):
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        pkg.cone_trace_svo(img, 45.0, v, p.data_ptr, (0, 0, 0), 1.0, pkg.RENDER_CARRY)
        ref.append(img.cpu().numpy().copy())
    assert not np.array_equal(ref[0], ref[1])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    imgs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(30):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                pkg.cone_trace_svo(imgs[k], 45.0, views[k], pools[k].data_ptr, (0, 0, 0), 1.0, pkg.RENDER_CARRY)
        if rep % 10 == 9:
            torch.cuda.synchronize()
            for k in (0, 1):
                assert np.array_equal(imgs[k].cpu().numpy(), ref[k]), (rep, k)
    torch.cuda.synchronize()
    for s in streams:
        pkg.check(pkg.lib().svoslam_cone_trace_release(s.cuda_stream, 0))


def test_set_words_then_async_fusion(env, oracle):
    """a pool whose contents were replaced from the host must allocate new tiles AFTER the loaded nodes (device-resident
    size reset), in the asynchronous and the phased path"""
    pkg, torch, synth, pl = env
    rng = np.random.default_rng(5)
    clouds = [surface_cloud(rng, 30000) for _ in range(3)]
    dev = [(torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda()) for p, c in clouds]
    depth = 9
    ws, ref = pkg.Workspace(), pkg.Pool()
    snaps = []
    for p, c in dev:
        pkg.svo_from_point_cloud_async(ws, p, c, depth, ref, (0, 0, 0), 1.0)
        snaps.append(ref.words().copy())
    B = pkg.Pool()
    pkg.svo_from_point_cloud_async(pkg.Workspace(), dev[2][0], dev[2][1], depth, B, (0, 0, 0), 1.0)   # B has its own history
    B.set_words(snaps[0])
    assert B.size == snaps[0].size // 2
    wsb = pkg.Workspace()
    pkg.svo_from_point_cloud_async(wsb, dev[1][0], dev[1][1], depth, B, (0, 0, 0), 1.0)
    assert np.array_equal(B.words(), snaps[1]), describe_mismatch(B.words(), snaps[1])
    pkg.svo_fuse_sort(wsb, dev[2][0], depth, (0, 0, 0), 1.0)
    pkg.svo_fuse_plan(wsb, dev[2][0].shape[0], depth, B)
    pkg.svo_fuse_commit(wsb, dev[2][1], depth, B)
    assert np.array_equal(B.words(), snaps[2])
    bad = snaps[0].copy()
    bad[0] = pkg.FLAG_CHILDREN | (bad.size // 2)          # child tile outside the pool
    with pytest.raises(pkg.SvoslamError):
        B.set_words(bad)


def test_pool_replicas_commit_to(env, oracle):
    """svoslam_pool_copy + one plan applied to two replicas (made against either of them): both equal the plain fusion"""
    pkg, torch, synth, pl = env
    rng = np.random.default_rng(8)
    clouds = [surface_cloud(rng, 50000) for _ in range(4)]
    dev = [(torch.from_numpy(p).cuda(), torch.from_numpy(c).cuda()) for p, c in clouds]
    depth = 10
    ref, wsr = pkg.Pool(1 << 22), pkg.Workspace()
    A, Bp, ws = pkg.Pool(1 << 22), pkg.Pool(), pkg.Workspace()
    pkg.svo_from_point_cloud_async(ws, dev[0][0], dev[0][1], depth, A, (0, 0, 0), 1.0)
    pkg.svo_from_point_cloud_async(wsr, dev[0][0], dev[0][1], depth, ref, (0, 0, 0), 1.0)
    Bp.copy_from(A)
    assert Bp.size == A.size and Bp.capacity >= A.capacity and np.array_equal(Bp.words(), A.words())
    s2 = torch.cuda.Stream()
    for k in (1, 2, 3):
        pkg.svo_from_point_cloud_async(wsr, dev[k][0], dev[k][1], depth, ref, (0, 0, 0), 1.0)
        planned, other = (A, Bp) if k % 2 else (Bp, A)
        pkg.svo_fuse_sort(ws, dev[k][0], depth, (0, 0, 0), 1.0)
        pkg.svo_fuse_plan(ws, dev[k][0].shape[0], depth, planned)
        torch.cuda.synchronize()
        if other.capacity < planned.capacity:      # the plan grew the replica it reserves on: the other follows
            other.reserve(planned.capacity)
        pkg.svo_fuse_commit_to(ws, dev[k][1], depth, other, 0, True)          # the replica the plan did not read, first
        with torch.cuda.stream(s2):
            pkg.svo_fuse_commit_to(ws, dev[k][1], depth, planned, 1, False)   # concurrently, own scratch slot
        torch.cuda.synchronize()
        want = ref.words()
        assert A.size == Bp.size == ref.size
        assert np.array_equal(A.words(), want) and np.array_equal(Bp.words(), want), k
    with pytest.raises(pkg.SvoslamError):          # the plan was consumed by the last application
        pkg.svo_fuse_commit_to(ws, dev[3][1], depth, A, 0, False)


@pytest.mark.parametrize("replicas", [2, 1])
def test_runner_replicas_and_validation(env, replicas):
    """the scheduler with two map replicas (opt-in) and with one (default): final image, pool, pose, counters of the sequential
    loop; a call with bad timestamps is refused BEFORE anything is enqueued and leaves the runner usable"""
    pkg, torch, synth, pl = env
    before = pkg.configure(runner_replicas=replicas)     # (taken when the runner is created)
    try:
        w, h, depth, center, edge, n = 160, 120, 8, (0.0, 1.5, 0.0), 4.096, 13
        frames = [synth.render_frame(k, w, h, device="cuda") for k in range(n)]
        views = [pl.ground_truth_view(k, synth) for k in range(n)]
        ds, cs = [f[0] for f in frames], [f[1] for f in frames]
        A = pl.SlamPipeline(w, h, depth, center, edge, count_steps=True)
        for k in range(n):
            ref = A.frame(ds[k], cs[k], k, views[k])
        ref = ref.cpu().numpy().copy()
        B = pl.SlamPipeline(w, h, depth, center, edge, count_steps=True, pool_capacity_nodes=1 << 12)   # grows mid-stream
        B.run_stream(ds[:5], cs[:5], list(range(5)), views[:5])
        with pytest.raises(pkg.SvoslamError):
            B.run_stream(ds[5:7], cs[5:7], [4, 5], views[5:7])        # not newer than the camera's latest frame
        with pytest.raises(pkg.SvoslamError):
            B.run_stream(ds[5:8], cs[5:8], [5, 7, 7], views[5:8])     # not strictly increasing
        B.run_stream(ds[5:6], cs[5:6], [5], views[5:6])               # a one-frame call
        B.run_stream(ds[6:], cs[6:], list(range(6, n)), views[6:])
        torch.cuda.synchronize()
        assert np.array_equal(B.image.cpu().numpy(), ref)
        assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words())
        assert np.array_equal(A.cam.pose()[0], B.cam.pose()[0]) and np.array_equal(A.cam.pose()[1], B.cam.pose()[1])
        assert A.counters.tolist() == B.counters.tolist()
    finally:
        pkg.configure(**before)


def test_obj_loader_refuses_bad_indices(env, tmp_path):
    """indices outside the file's vertices / texture coordinates, two-corner faces and partly textured files are
    refused (the reference's loader reads out of bounds, objloader.cpp)"""
    pkg, torch, synth, pl = env
    base = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n"
    good = tmp_path / "good.obj"
    good.write_text(base + "f 1/1 2/2 3/3\n")
    assert pkg.Mesh(good).n_tris == 1
    for name, faces in (("big", "f 1 2 4\n"), ("zero", "f 0 1 2\n"), ("neg", "f -1 1 2\n"), ("two", "f 1 2\n"),
                        ("tcbig", "f 1/1 2/2 3/9\n"), ("mixed", "f 1/1 2/2 3/3\nf 1 2 3\n")):
        p = tmp_path / (name + ".obj")
        p.write_text(base + faces)
        with pytest.raises(pkg.SvoslamError):
            pkg.Mesh(p)


@pytest.mark.parametrize("depth", [3, 5, 6, 9])
def test_incremental_level_grid_follows_async_fusions(env, oracle, depth):
    """the level grid of the ray march belongs to the pool and is refreshed block by block (pool_grid.hpp): after every
    asynchronous fusion -- first frames with splits near the root, trees shallower than the block level, colour-only
    updates of an old map -- the render must equal the oracle's; also after the blocking path (whole-grid rebuild)"""
    pkg, torch, synth, pl = env
    rng = np.random.default_rng(depth)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(1 << 20), oracle.Pool()
    view = oracle.look_at((0.3, 0.2, -2.4), (0, 0, 0), (0, 1, 0))
    w, h = 96, 72
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    for k in range(7):
        n = 300 if k < 2 else 20000          # sparse first frames: large childless cubes get split later
        pts, col = surface_cloud(rng, n)
        if k == 5:
            pts = np.ascontiguousarray(prev)          # same cloud again: colours / alpha only (plus the Q4 splits)
        prev = pts
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        if k == 4:
            pkg.svo_from_point_cloud(ws, tp, tc, depth, pool, (0, 0, 0), 1.0)      # blocking path
        else:
            pkg.svo_from_point_cloud_async(ws, tp, tc, depth, pool, (0, 0, 0), 1.0)
        opool.insert_cloud(pts, col, depth, (0, 0, 0), 1.0)
        for mode in (1, 0):
            pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, (0, 0, 0), 1.0, mode)
            ref, _, _ = oracle.cone_trace(opool, w, h, 45.0, view, (0, 0, 0), 1.0, mode)
            got = img.cpu().numpy()
            assert np.array_equal(got, ref), (k, mode, describe_mismatch(got, ref))
    assert np.array_equal(pool.words(), opool.words())


def test_pool_touch_after_external_write(env, oracle):
    """node memory written behind the library's back: svoslam_pool_touch makes the next render rebuild the grid"""
    pkg, torch, synth, pl = env
    a = build_pool(pkg, torch, 21, 8, frames=7, n=60000)      # a denser, older map: visibly different render
    b = build_pool(pkg, torch, 22, 8)
    wa, wb = a.words().copy(), b.words().copy()
    view = oracle.look_at((0.1, 0.2, -2.6), (0, 0, 0), (0, 1, 0))
    w, h = 96, 72
    img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    pkg.cone_trace_svo(img, 45.0, view, a.data_ptr, (0, 0, 0), 1.0, 1)
    ra, _, _ = oracle.cone_trace(wa, w, h, 45.0, view, (0, 0, 0), 1.0, 1)
    assert np.array_equal(img.cpu().numpy(), ra)
    n = min(wa.size, wb.size)
    assert a.capacity * 2 >= wb.size
    import ctypes as C
    torch.cuda.synchronize()
    pkg._hip().hipMemcpy(C.c_void_p(a.data_ptr), C.c_void_p(wb.ctypes.data), C.c_size_t(wb.nbytes), 1)   # raw overwrite
    pkg.check(pkg.lib().svoslam_pool_touch(C.byref(a._p)))
    pkg.cone_trace_svo(img, 45.0, view, a.data_ptr, (0, 0, 0), 1.0, 1)
    rb, _, _ = oracle.cone_trace(wb, w, h, 45.0, view, (0, 0, 0), 1.0, 1)
    assert np.array_equal(img.cpu().numpy(), rb) and not np.array_equal(ra, rb)
