"""GPU parity: coneTraceSVO through the C ABI vs the CPU oracle, byte for byte."""
import json
import os

import numpy as np
import pytest

from util import describe_mismatch, surface_cloud

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat_appendix_c.json")))


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    return svoslam_pkg.load(), torch


def build_pool(pkg, torch, oracle, depth, frames, n=20000, seed=3, edge=1.0, center=(0, 0, 0), scale=1.0):
    rng = np.random.default_rng(seed)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    pts, col = surface_cloud(rng, n)
    pts = (pts * np.float32(scale) + np.asarray(center, np.float32)).astype(np.float32)
    for _ in range(frames):
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        pkg.svo_from_point_cloud(ws, tp, tc, depth, pool, center, edge)
        opool.insert_cloud(pts, col, depth, center, edge)
    return ws, pool, opool


def render_both(pkg, torch, oracle, pool, words, w, h, view, center, size, mode):
    img = torch.full((h, w, 4), 7, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, center, size, mode, counters=cnt)
    ref, steps, levels = oracle.cone_trace(words, w, h, 45.0, view, center, size, mode)
    got = img.cpu().numpy()
    assert np.array_equal(got, ref), describe_mismatch(got, ref)
    assert cnt.cpu().tolist() == [steps, levels]
    return got


def test_kat_c6_on_gpu(env, oracle):
    pkg, torch = env
    c1, c2, c6 = KAT["C1_keys"], KAT["C2_insert"], KAT["C6_render"]
    for case in c6["cases"]:
        opool = oracle.Pool()
        opool.insert_cloud(c1["points"], c2["colors"], 2, c1["center"], c1["half_edge"])
        w = opool.words()
        r, g, b, a = case["rgba"]
        w[2 * c6["node"] + 1] = r | (g << 8) | (b << 16) | (a << 24)
        pool = pkg.Pool()
        pool.set_words(w)
        view = oracle.look_at(c6["origin"], [1.0, 0.3, 0.2], [0, 1, 0])
        img = torch.zeros((12, 16, 4), dtype=torch.uint8, device="cuda")
        pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, c6["center"], c6["size"])
        assert (img.cpu().numpy().reshape(-1, 4) == np.array(case["pixel"], np.uint8)).all()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("depth,frames", [(5, 1), (8, 3), (10, 70)])
def test_render_matches_oracle(env, oracle, depth, frames, mode):
    pkg, torch = env
    ws, pool, opool = build_pool(pkg, torch, oracle, depth, frames, n=12000 if frames > 10 else 20000)
    words = opool.words()
    for eye, tgt, (w, h) in (((0.1, 0.2, -2.6), (0, 0, 0), (160, 120)), ((1.5, 0.9, 1.2), (0.1, -0.2, -0.3), (97, 61)),
                             ((0.05, 0.05, 0.3), (0.4, 0.1, 0.2), (64, 48))):
        view = oracle.look_at(eye, tgt, (0, 1, 0))
        render_both(pkg, torch, oracle, pool, words, w, h, view, (0, 0, 0), 1.0, mode)


def test_render_empty_pool_and_outside_root(env, oracle):
    pkg, torch = env
    pool = pkg.Pool()
    words = np.zeros(16, np.uint32)
    view = oracle.look_at((20.0, 3.0, 1.0), (0, 0, 0), (0, 1, 0))  # camera far outside the root cube (Q11)
    got = render_both(pkg, torch, oracle, pool, words, 40, 30, view, (0, 0, 0), 1.0, 0)
    assert (got[..., 3] == 255).all()


def test_render_640x480_matches_oracle(env, oracle):
    pkg, torch = env
    ws, pool, opool = build_pool(pkg, torch, oracle, 9, 2, n=60000)
    view = oracle.look_at((0.2, 0.3, -2.2), (0, 0, 0), (0, 1, 0))
    render_both(pkg, torch, oracle, pool, opool.words(), 640, 480, view, (0, 0, 0), 1.0, 1)


@pytest.mark.parametrize("mode", [0, 1])
def test_render_depth16_close_range_and_odd_geometry(env, oracle, mode):
    """depth-16 pool, root centre / half-edge that are not dyadic, cameras a few centimetres from the surface:
    the LOD depth exceeds 12, so the walk leaves the LDS split-plane table for the fine (global) one"""
    pkg, torch = env
    center, edge = (0.37, -0.21, 0.11), 0.873
    ws, pool, opool = build_pool(pkg, torch, oracle, 16, 2, n=15000, edge=edge, center=center, scale=0.8)
    words = opool.words()
    c = np.asarray(center)
    for eye, tgt, (w, h) in (((0.0, 0.0, 0.22), (0.0, 0.0, 0.0), (160, 400)), ((0.3, 0.1, 0.5), (0.2, -0.1, 0.1), (64, 48)),
                             ((0.1, 0.2, -2.0), (0, 0, 0), (80, 60))):
        view = oracle.look_at(tuple(np.asarray(eye) * 0.8 + c), tuple(np.asarray(tgt) * 0.8 + c), (0, 1, 0))
        render_both(pkg, torch, oracle, pool, words, w, h, view, center, edge, mode)


def test_render_far_from_origin_falls_back_to_the_chain(env, oracle):
    """root cube far from the origin: its split planes are closer together than one float ulp of the coordinates,
    the sorted-table bracket cannot be confirmed and the lanes take the reference's own comparison chain"""
    pkg, torch = env
    center, edge = (1000.25, 3.5, -777.125), 0.75
    ws, pool, opool = build_pool(pkg, torch, oracle, 12, 2, n=8000, edge=edge, center=center, scale=0.6)
    c = np.asarray(center)
    for eye, tgt in (((0.1, 0.2, -1.4), (0, 0, 0)), ((0.05, 0.0, 0.3), (0.3, 0.1, 0.2))):
        view = oracle.look_at(tuple(np.asarray(eye) + c), tuple(np.asarray(tgt) + c), (0, 1, 0))
        render_both(pkg, torch, oracle, pool, opool.words(), 72, 54, view, center, edge, 0)


def test_cone_trace_timing_log(env, oracle):
    pkg, torch = env
    ws, pool, opool = build_pool(pkg, torch, oracle, 6, 1, n=4000)
    img = torch.zeros((30, 40, 4), dtype=torch.uint8, device="cuda")
    view = oracle.look_at((0.1, 0.2, -2.6), (0, 0, 0), (0, 1, 0))
    pkg.cone_trace_timing(True)
    for _ in range(3):
        pkg.cone_trace_svo(img, 45.0, view, pool.data_ptr, (0, 0, 0), 1.0)
    ms, n = pkg.cone_trace_timing_read()
    pkg.cone_trace_timing(False)
    assert n == 3 and 0.0 < ms < 100.0
    assert pkg.cone_trace_timing_read() == (0.0, 0)


@pytest.mark.parametrize("mode", [0, 1])
def test_render_tall_image_large_root_walks_three_levels_below_the_lds_table(env, oracle, mode):
    """1080-row images of an 8 m cube at depth 14: LOD depths 12..14, i.e. up to three levels below the 11-level LDS
    table, continued by the centre chain (round 1 selected a 12-level table here; it is now opt-in, see the
    subprocess test below)"""
    pkg, torch = env
    center, edge = (0.0, 0.0, 0.0), 8.192
    ws, pool, opool = build_pool(pkg, torch, oracle, 14, 2, n=20000, edge=edge, center=center, scale=5.0)
    words = opool.words()
    for eye, tgt in (((0.5, 1.0, -11.0), (0, 0, 0)), ((1.0, 0.5, 2.6), (0.5, -1.0, -1.5))):
        view = oracle.look_at(eye, tgt, (0, 1, 0))
        render_both(pkg, torch, oracle, pool, words, 40, 1100, view, center, edge, mode)


def test_render_megapixel_uses_the_level8_grid(env, oracle):
    """>= 2^20 rays: the 256^3 grid variant is selected (rendering memory that is not a registered pool)"""
    pkg, torch = env
    center, edge = (0.0, 0.0, 0.0), 8.192
    ws, pool, opool = build_pool(pkg, torch, oracle, 14, 2, n=30000, edge=edge, center=center, scale=5.0)
    view = oracle.look_at((0.5, 1.0, -11.0), (0, 0, 0), (0, 1, 0))
    render_both(pkg, torch, oracle, pool, opool.words(), 960, 1100, view, center, edge, 0)



@pytest.mark.parametrize("mode", [0, 1])
def test_render_from_far_outside_a_small_root_uses_the_grid_pyramid(env, oracle, mode):
    """the eye 3 .. 8 half-edges outside a SMALL root cube whose content reaches its faces (a mesh seen from
    outside, configs 2 and 5): the cone LOD is coarser than the level grid for most of the way (LODs 3 .. 7), samples beyond the
    cube clamp into boundary cells (Q11) and rays run on to MAX_RANGE = 33 half-edges.  Every such sample is answered by the grid's
    pyramid (pool_grid.hpp) -- images and step / level counters byte-equal to the oracle, for a pool of this library (full build,
    then block-wise updates after more fusions through the asynchronous path) and for foreign node memory (grid built per render)."""
    pkg, torch = env
    center, edge, depth = (0.05, -0.02, 0.01), 0.3, 10
    rng = np.random.default_rng(11)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    c = np.asarray(center, np.float32)
    views = [oracle.look_at(tuple(c + np.asarray(o, np.float32) * np.float32(edge)), tuple(c), (0, 1, 0))
             for o in ((3.1, 0.2, 0.4), (-2.2, 0.1, 0.3), (4.5, 3.0, -5.5))]
    for frame in range(3):
        pts, col = surface_cloud(rng, 9000)
        pts = (pts * np.float32(edge / 0.85) + c).astype(np.float32)          # reaches (and leaves: clamped, Q11) the faces
        pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), depth, pool, center, edge)
        opool.insert_cloud(pts, col, depth, center, edge)
        words = opool.words()
        for vi, view in enumerate(views):
            w, h = ((96, 72), (64, 48), (40, 30))[vi]
            render_both(pkg, torch, oracle, pool, words, w, h, view, center, edge, mode)
    # deferred commit + apply (the frame loop's default at 640x480: the marks go to the other dirty state): a render between the two
    # halves still sees the old map, one after the apply the new one -- pyramid entries included
    for frame in range(2):
        pts, col = surface_cloud(rng, 7000)
        pts = (pts * np.float32(edge / 0.9) + c).astype(np.float32)
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        pkg.svo_fuse_sort(ws, tp, depth, center, edge)
        pkg.svo_fuse_plan(ws, len(pts), depth, pool)
        pkg.svo_fuse_commit_deferred(ws, tc, depth, pool)
        render_both(pkg, torch, oracle, pool, words, 96, 72, views[0], center, edge, mode)       # still the old map
        pkg.svo_fuse_apply(ws, pool)
        opool.insert_cloud(pts, col, depth, center, edge)
        words = opool.words()
        for vi in (0, 2):
            render_both(pkg, torch, oracle, pool, words, (96, 40)[vi // 2], (72, 30)[vi // 2], views[vi], center, edge, mode)
    # the same nodes as memory the library does not know: the per-render grid + pyramid (level 7; level 8 for a megapixel)
    foreign = torch.from_numpy(words.view(np.int32).copy()).cuda()
    img = torch.zeros((72, 96, 4), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    pkg.cone_trace_svo(img, 45.0, views[0], foreign.data_ptr(), center, edge, mode, counters=cnt)
    ref, steps, levels = oracle.cone_trace(words, 96, 72, 45.0, views[0], center, edge, mode)
    assert np.array_equal(img.cpu().numpy(), ref) and cnt.cpu().tolist() == [steps, levels]
    if mode == 1:
        return
    big = torch.zeros((1100, 960, 4), dtype=torch.uint8, device="cuda")
    cnt.zero_()
    pkg.cone_trace_svo(big, 45.0, views[1], foreign.data_ptr(), center, edge, mode, counters=cnt)
    ref, steps, levels = oracle.cone_trace(words, 960, 1100, 45.0, views[1], center, edge, mode)
    assert np.array_equal(big.cpu().numpy(), ref) and cnt.cpu().tolist() == [steps, levels]
