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


def build_pool(pkg, torch, oracle, depth, frames, n=20000, seed=3, edge=1.0):
    rng = np.random.default_rng(seed)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    pts, col = surface_cloud(rng, n)
    for _ in range(frames):
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        pkg.svo_from_point_cloud(ws, tp, tc, depth, pool, (0, 0, 0), edge)
        opool.insert_cloud(pts, col, depth, (0, 0, 0), edge)
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
