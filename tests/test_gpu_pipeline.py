"""End-to-end GPU parity: whole SLAM frames (bilateral + ICP + back-projection + fusion + raycast) through
pipeline.SlamPipeline vs the same frame assembled from CPU-oracle calls; and the row-band / RCCL code path
(run here with one rank) vs the single-GPU path."""
import importlib
import os

import numpy as np
import pytest

from util import describe_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    synth = importlib.import_module("octree_slam_amd.synth")
    pl = importlib.import_module("octree_slam_amd.pipeline")
    return pkg, torch, synth, pl


def oracle_frame(oracle, cam, pool, d, c, k, view, w, h, depth, center, edge, mode):
    f = 570.3 * w / 640.0
    cam.update(d, c, k)
    v = oracle.vertex_map(d, f, f, w, h)
    v = oracle.transform_vertex_map(v, cam.fusion_transform())
    pool.insert_cloud(v.reshape(-1, 3), c.reshape(-1, 3), depth, center, edge)
    img, steps, levels = oracle.cone_trace(pool, w, h, 45.0, view, center, edge, mode)
    return v, img


@pytest.mark.parametrize("w,h,depth,edge,frames,mode", [(160, 120, 8, 4.096, 4, 0), (320, 240, 10, 4.096, 3, 1)])
def test_frames_match_oracle(env, oracle, w, h, depth, edge, frames, mode):
    pkg, torch, synth, pl = env
    center = (0.0, 1.5, 0.0)
    P = pl.SlamPipeline(w, h, depth, center, edge, render_mode=mode)
    ocam, opool = oracle.Camera(w, h, P.focal, P.focal), oracle.Pool()
    for k in range(frames):
        d, c = synth.render_frame(2 * k, w, h)
        view = pl.ground_truth_view(2 * k, synth)
        img = P.frame(d.cuda(), c.cuda(), k, view).cpu().numpy()
        rv, rimg = oracle_frame(oracle, ocam, opool, d.numpy().view(np.uint16), c.numpy(), k, view, w, h, depth, center, edge, mode)
        gw, cw = P.pool.words(), opool.words()
        assert P.pool.size == opool.size and np.array_equal(gw, cw), (k, describe_mismatch(gw, cw))
        assert np.array_equal(img, rimg), (k, describe_mismatch(img, rimg))
        p, o = P.cam.pose(); rp, ro = ocam.pose()
        assert np.array_equal(o.view(np.uint32), ro.view(np.uint32))
        b = P.bbox.cpu().numpy()
        pts = rv.reshape(-1, 3)
        ok = np.isfinite(pts[:, 0]) & np.isfinite(pts[:, 2])
        assert b[6] == 1.0 and np.array_equal(b[:3], pts[ok].min(0)) and np.array_equal(b[3:6], pts[ok].max(0))


@pytest.mark.parametrize("exchange", ["allreduce", "none"])
def test_row_band_path_equals_single_gpu_path(env, exchange):
    """the multi-GPU code paths -- "allreduce": stepping tracker + all-reduce, band back-projection + all-gather;
    "none": whole-frame tracker and fusion on every rank; band raycast in both -- run with ONE rank over RCCL must
    reproduce the single-GPU path bit for bit"""
    pkg, torch, synth, pl = env
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        w, h, depth, center, edge = 320, 240, 10, (0.0, 1.5, 0.0), 4.096
        A = pl.SlamPipeline(w, h, depth, center, edge)
        B = pl.SlamPipeline(w, h, depth, center, edge, dist=pl.DistContext(0, 1, force=True, exchange=exchange))
        for k in range(4):
            d, c = synth.render_frame(k, w, h, device="cuda")
            view = pl.ground_truth_view(k, synth)
            ia = A.frame(d, c, k, view).cpu().numpy()
            ib = B.frame(d, c, k, view).cpu().numpy()
            assert np.array_equal(ia, ib)
            assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words())
            assert np.array_equal(A.cam.pose()[1], B.cam.pose()[1])
    finally:
        if created:
            dist.destroy_process_group()


def test_band_render_tiles_assemble_full_image(env, oracle):
    """rendering rows in bands (any split) gives the same image as one full render"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 160, 120, 8, (0.0, 1.5, 0.0), 4.096
    P = pl.SlamPipeline(w, h, depth, center, edge)
    for k in range(3):
        d, c = synth.render_frame(k, w, h, device="cuda")
        full = P.frame(d, c, k, pl.ground_truth_view(k, synth)).cpu().numpy().copy()
    view = pl.ground_truth_view(2, synth)
    for world in (2, 3, 8):
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        for r in range(world):
            first, rows = pl.band_rows(h, r, world)
            pkg.cone_trace_svo_band(img, first, rows, 45.0, view, P.pool.data_ptr, center, edge)
        assert np.array_equal(img.cpu().numpy(), full)


@pytest.mark.parametrize("capacity", [1 << 20, 1 << 12])
def test_stream_overlap_equals_sequential(env, capacity):
    """run_stream (maps of frame k+2, ICP of frame k+1, preparation of frame k+1 and commit + raycast of frame k on
    four HIP streams) must give exactly the images, pool and poses of the strictly sequential frame() loop -- also when
    the pool has to grow in the middle of the stream (capacity 4096 nodes)"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 320, 240, 10, (0.0, 1.5, 0.0), 4.096
    n = 20
    frames = [synth.render_frame(k, w, h, device="cuda") for k in range(n)]
    views = [pl.ground_truth_view(k, synth) for k in range(n)]
    A = pl.SlamPipeline(w, h, depth, center, edge, render_mode=1)
    seq_imgs = [A.frame(frames[k][0], frames[k][1], k, views[k]).cpu().numpy().copy() for k in range(n)]
    for rep in range(3):
        B = pl.SlamPipeline(w, h, depth, center, edge, render_mode=1, pool_capacity_nodes=capacity)
        imgs = []
        B.run_stream([f[0] for f in frames], [f[1] for f in frames], list(range(n)), views,
                     on_render=lambda i, im: imgs.append(im.clone()) if im is not None else None)
        torch.cuda.synchronize()
        assert len(imgs) == n
        for k in range(n):
            assert np.array_equal(imgs[k].cpu().numpy(), seq_imgs[k]), (rep, k)
        assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words())
        assert np.array_equal(A.cam.pose()[1], B.cam.pose()[1])


def test_phase_api_contracts(env):
    """the split entry points refuse calls out of order instead of running on stale state"""
    pkg, torch, synth, pl = env
    ws, pool = pkg.Workspace(), pkg.Pool()
    pts = torch.rand((1000, 3), device="cuda") - 0.5
    col = torch.randint(0, 256, (1000, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(pkg.SvoslamError):      # plan before sort
        pkg.svo_fuse_plan(ws, 1000, 8, pool)
    pkg.svo_fuse_sort(ws, pts, 8, (0, 0, 0), 1.0)
    with pytest.raises(pkg.SvoslamError):      # commit before plan
        pkg.svo_fuse_commit(ws, col, 8, pool)
    pkg.svo_fuse_plan(ws, 1000, 8, pool)
    with pytest.raises(pkg.SvoslamError):      # commit of a different batch size than planned
        pkg.svo_fuse_commit(ws, col[:500], 8, pool)
    pkg.svo_fuse_commit(ws, col, 8, pool)
    with pytest.raises(pkg.SvoslamError):      # a plan is consumed by its commit
        pkg.svo_fuse_commit(ws, col, 8, pool)
    ref = pkg.Pool()
    pkg.svo_from_point_cloud(pkg.Workspace(), pts, col, 8, ref, (0, 0, 0), 1.0)
    assert pool.size == ref.size and np.array_equal(pool.words(), ref.words())
    # camera: at most two frames prepared ahead; track needs a prepared frame; update refuses to interleave
    w, h = 160, 120
    cam = pkg.Camera(w, h, 142.6, 142.6)
    frames = [synth.render_frame(k, w, h, device="cuda") for k in range(3)]
    with pytest.raises(pkg.SvoslamError):
        cam.track_prepared()
    assert cam.prepare(frames[0][0], frames[0][1], 0) == 1
    assert cam.prepare(frames[1][0], frames[1][1], 1) == 1
    with pytest.raises(pkg.SvoslamError):
        cam.prepare(frames[2][0], frames[2][1], 2)
    with pytest.raises(pkg.SvoslamError):
        cam.update(frames[2][0], frames[2][1], 2)
    cam.track_prepared(); cam.track_prepared()
    assert cam.prepare(frames[1][0], frames[1][1], 1) == 0        # stale timestamp: not used (rgbd_camera.cpp:55-59)
    other = pkg.Camera(w, h, 142.6, 142.6)
    for k in range(2):
        other.update(frames[k][0], frames[k][1], k)
    assert np.array_equal(cam.pose()[1], other.pose()[1]) and np.array_equal(cam.pose()[0], other.pose()[0])


def test_pipeline_reset_replays_identically(env):
    """reset() (bench.py's initialisation pass relies on it): empty map + fresh tracker on the same allocations and
    recorded graphs reproduce a fresh pipeline bit for bit"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge, n = 160, 120, 8, (0.0, 1.5, 0.0), 4.096, 7
    frames = [synth.render_frame(k, w, h, device="cuda") for k in range(n)]
    views = [pl.ground_truth_view(k, synth) for k in range(n)]
    ds, cs = [f[0] for f in frames], [f[1] for f in frames]
    A = pl.SlamPipeline(w, h, depth, center, edge)
    ref = []
    A.run_stream(ds, cs, list(range(n)), views, on_render=lambda i, im: ref.append(im.clone()) if im is not None else None)
    torch.cuda.synchronize()
    words, pose = A.pool.words().copy(), A.cam.pose()
    A.reset()
    assert A.pool.size == 8 and not A.pool.words().any()
    again = []
    A.run_stream(ds, cs, list(range(n)), views, on_render=lambda i, im: again.append(im.clone()) if im is not None else None)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, again))
    assert np.array_equal(A.pool.words(), words) and np.array_equal(A.cam.pose()[1], pose[1])


@pytest.mark.parametrize("capacity,band", [(1 << 22, None), (4096, None), (1 << 22, (2, 5))])
def test_native_runner_equals_sequential(env, capacity, band):
    """the frame scheduler inside the library (csrc/runner.hip, what run_stream uses without hooks) against the
    sequential frame() loop: last image, pool, pose and step counters bit for bit -- in several chunks, across a
    pool growth (capacity 4096), for one row band of five (what a rank of a multi-GPU job renders), and against the
    scripted scheduler"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 320, 240, 10, (0.0, 1.5, 0.0), 4.096
    n = 23
    frames = [synth.render_frame(k, w, h, device="cuda") for k in range(n)]
    views = [pl.ground_truth_view(k, synth) for k in range(n)]
    ds, cs = [f[0] for f in frames], [f[1] for f in frames]
    A = pl.SlamPipeline(w, h, depth, center, edge, count_steps=True)
    for k in range(n):
        ref = A.frame(ds[k], cs[k], k, views[k])
    ref = ref.cpu().numpy().copy()
    for rep in range(2):
        B = pl.SlamPipeline(w, h, depth, center, edge, pool_capacity_nodes=capacity, count_steps=band is None)
        if band is not None:
            B.first, B.rows = pl.band_rows(h, band[0], band[1])
            B.dist.force = True
        for a in range(0, n, 9):                      # three calls back to back, no synchronisation in between
            b = min(n, a + 9)
            B.run_stream(ds[a:b], cs[a:b], list(range(a, b)), views[a:b])
        assert hasattr(B, "_runner")                  # the native path was taken
        torch.cuda.synchronize()
        got = B.image.cpu().numpy()
        rows = slice(B.first, B.first + B.rows)
        assert np.array_equal(got[rows], ref[rows]), rep
        assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words())
        assert np.array_equal(A.cam.pose()[0], B.cam.pose()[0]) and np.array_equal(A.cam.pose()[1], B.cam.pose()[1])
        if band is None:
            assert A.counters.tolist() == B.counters.tolist()
        # computePointCloudBoundingBox of the last frame: the runner's fused front end (svoslam_svo_fuse_sort_frame)
        # against the stand-alone kernel of the sequential loop
        assert np.array_equal(B._runner.bbox(), A.bbox.cpu().numpy()) and B._runner.bbox()[6] == 1.0
    os.environ["SVOSLAM_PY_SCHEDULER"] = "1"
    try:
        C_ = pl.SlamPipeline(w, h, depth, center, edge)
        C_.run_stream(ds, cs, list(range(n)), views)
        torch.cuda.synchronize()
        assert not hasattr(C_, "_runner")
        assert np.array_equal(C_.image.cpu().numpy(), ref) and np.array_equal(C_.pool.words(), A.pool.words())
    finally:
        del os.environ["SVOSLAM_PY_SCHEDULER"]
    with pytest.raises(pkg.SvoslamError):             # timestamps must increase
        B.run_stream(ds[:2], cs[:2], [0, 0], views[:2])


def test_graph_replay_and_chain_tracker_in_subprocess(env):
    """HIP-graph replay of the launch sequences (svoslam_config.graphs = 1; off by default since round 2) together with the
    launch-chain tracker (track_mode = 1), in a child process (settings through SVOSLAM_CONFIG): same final image, pool and pose as the default
    (direct launches, one-launch tracker) in this process"""
    import hashlib
    import json
    import subprocess
    import sys
    pkg, torch, synth, pl = env
    code = r'''
import sys, json, hashlib, importlib, numpy as np, torch
sys.path.insert(0, %r)
import svoslam_pkg
pkg = svoslam_pkg.load()
synth = importlib.import_module("octree_slam_amd.synth")
pl = importlib.import_module("octree_slam_amd.pipeline")
w, h, depth, center, edge, n = 160, 120, 8, (0.0, 1.5, 0.0), 4.096, 11
frames = [synth.render_frame(k, w, h) for k in range(n)]
ds, cs = [f[0].cuda() for f in frames], [f[1].cuda() for f in frames]
views = [pl.ground_truth_view(k, synth) for k in range(n)]
P = pl.SlamPipeline(w, h, depth, center, edge)
P.run_stream(ds[:6], cs[:6], list(range(6)), views[:6])
P.run_stream(ds[6:], cs[6:], list(range(6, n)), views[6:])
torch.cuda.synchronize()
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
print("RESULT" + json.dumps([sha(P.image.cpu().numpy()), sha(P.pool.words()), sha(P.cam.pose()[1]), int(P.pool.size)]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(extra):
        e = dict(os.environ, **pkg.config_env(**extra))
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0][6:])
    base = run({})
    assert run(dict(graphs=1, track_mode=1)) == base
    assert run(dict(graphs=1)) == base
    # launch chain by direct launches (work maps: iteration it applies one matrix to what iteration it - 1 stored)
    assert run(dict(track_mode=1)) == base
    # the scheduler with deferred commits (commit of frame k+1 computed beside the march of frame k, then applied: the
    # default at this image size since round 3) and with in-place commits; with and without the occupancy bricks
    assert run(dict(runner_deferred=1)) == base
    assert run(dict(runner_deferred=1, runner_lead=0)) == base
    assert run(dict(runner_deferred=0)) == base
    assert run(dict(runner_deferred=0, march_bricks=0)) == base
    assert run(dict(march_bricks=0)) == base
    assert base[3] > 8


def test_fuse_frame_equals_backproject_plus_fuse(env):
    """the frame loop's fusion in its sequential form (fused front end from the depth image, plan, early split, commit)
    against back-projection into a point cloud + the one-call asynchronous fusion: pool, bounding box, image"""
    pkg, torch, synth, pl = env
    w, h, depth, center, edge = 320, 240, 10, (0.0, 1.5, 0.0), 4.096
    A = pl.SlamPipeline(w, h, depth, center, edge)
    B = pl.SlamPipeline(w, h, depth, center, edge)
    for k in range(5):
        d, c = synth.render_frame(k, w, h, device="cuda")
        view = pl.ground_truth_view(k, synth)
        ia = A.frame(d, c, k, view).cpu().numpy()
        B.track(d, c, k)
        B.fuse_frame(d, c)
        ib = B.render(view).cpu().numpy()
        assert np.array_equal(ia, ib), k
        assert A.pool.size == B.pool.size and np.array_equal(A.pool.words(), B.pool.words()), k
        assert np.array_equal(A.bbox.cpu().numpy(), B.bbox.cpu().numpy()) and float(B.bbox[6]) == 1.0
