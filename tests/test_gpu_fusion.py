"""GPU parity: svoFromPointCloud / svoFromVoxelGrid / extractVoxelGridFromSVO through the C ABI
vs the CPU oracle.  Integer work: the whole node pool must be bit-identical (node indices, child
pointers, colours, alpha)."""
import json
import os

import numpy as np
import pytest

from util import describe_mismatch, random_cloud, rgba, surface_cloud

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat_appendix_c.json")))


@pytest.fixture(scope="module")
def env():
    import torch
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    assert torch.cuda.is_available()
    assert pkg.device_arch().startswith("gfx950")
    return pkg, torch


def gpu_insert(pkg, torch, ws, pool, pts, col, depth, center, edge):
    tp = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).cuda()
    tc = torch.from_numpy(np.ascontiguousarray(col, np.uint8)).cuda()
    return pkg.svo_from_point_cloud(ws, tp, tc, depth, pool, center, edge)


def assert_pools_equal(pool, opool):
    assert pool.size == opool.size
    g, c = pool.words(), opool.words()
    assert np.array_equal(g, c), describe_mismatch(g, c)


def test_kat_c2_on_gpu(env, oracle):
    pkg, torch = env
    c1, c2 = KAT["C1_keys"], KAT["C2_insert"]
    ws, pool = pkg.Workspace(), pkg.Pool()
    st = gpu_insert(pkg, torch, ws, pool, c1["points"], c2["colors"], 2, c1["center"], c1["half_edge"])
    assert pool.size == c2["size_after_first"] and st.num_split == 2 and list(st.pass_sizes)[:2] == [2, 0]
    w = pool.words()
    assert int(w[0]) == c2["node0_word0_after_first"] and int(w[14]) == c2["node7_word0_after_first"]
    for node, val in c2["nodes_after_first"].items():
        assert rgba(int(w[2 * int(node) + 1])) == val, node
    gpu_insert(pkg, torch, ws, pool, c1["points"], c2["colors"], 2, c1["center"], c1["half_edge"])
    assert pool.size == c2["size_after_second"]
    w = pool.words()
    assert int(w[2 * 23]) == c2["node23_word0_after_second"]
    for node, val in c2["nodes_after_second"].items():
        assert rgba(int(w[2 * int(node) + 1])) == val, node
    gpu_insert(pkg, torch, ws, pool, c1["points"], c2["colors"], 2, c1["center"], c1["half_edge"])
    assert pool.size == c2["size_after_second"]


@pytest.mark.parametrize("depth,n,frames", [(1, 300, 2), (2, 2000, 3), (5, 20000, 3), (8, 30000, 3), (10, 40000, 2),
                                            (12, 50000, 2), (16, 20000, 2)])
def test_cloud_fusion_matches_oracle(env, oracle, depth, n, frames):
    pkg, torch = env
    rng = np.random.default_rng(100 + depth)
    ws, pool = pkg.Workspace(), pkg.Pool()
    opool = oracle.Pool()
    center, edge = (0.05, -0.02, 0.01), 1.0
    for f in range(frames):
        pts, col = (surface_cloud(rng, n) if f % 2 == 0 else random_cloud(rng, n, nan_every=53, dup_frac=0.1))
        pts = pts + np.float32(0.003 * f)
        st = gpu_insert(pkg, torch, ws, pool, pts, col, depth, center, edge)
        before = opool.size if opool.size else 8
        opool.insert_cloud(pts, col, depth, center, edge)
        assert st.pool_size_after == opool.size and st.pool_size_before == before
        assert_pools_equal(pool, opool)


def test_split_pass_sizes_match_reference_passes(env, oracle):
    pkg, torch = env
    rng = np.random.default_rng(5)
    depth, center, edge = 7, (0, 0, 0), 1.0
    ws, pool = pkg.Workspace(), pkg.Pool()
    opool = oracle.Pool()
    opool.insert_cloud(np.zeros((0, 3)), np.zeros((0, 3)), depth, center, edge)
    for f in range(3):
        pts, col = surface_cloud(rng, 8000)
        keys = oracle.compute_keys(pts, depth, center, edge)
        total, sizes, _ = opool.prepass(keys, depth)
        st = gpu_insert(pkg, torch, ws, pool, pts, col, depth, center, edge)
        assert st.num_split == total
        assert list(st.pass_sizes)[:depth] == sizes
        opool.insert_cloud(pts, col, depth, center, edge)
        assert_pools_equal(pool, opool)


def test_edge_cases(env, oracle):
    pkg, torch = env
    ws = pkg.Workspace()
    center, edge = (0, 0, 0), 1.0
    # empty input: pool initialised to 8 zero nodes, nothing else
    pool = pkg.Pool()
    st = pkg.svo_from_point_cloud(ws, None, None, 5, pool, center, edge)
    assert pool.size == 8 and st.num_split == 0 and (pool.words() == 0).all()
    # all points invalid
    pts = np.full((100, 3), np.nan, np.float32)
    col = np.zeros((100, 3), np.uint8)
    st = gpu_insert(pkg, torch, ws, pool, pts, col, 5, center, edge)
    assert pool.size == 8 and (pool.words() == 0).all()
    # a single point, all points identical, and points outside the root cube (Q11 clamp)
    for pts in (np.array([[0.3, 0.3, 0.3]], np.float32), np.tile(np.array([[0.7, -0.7, 0.2]], np.float32), (500, 1)),
                np.array([[5, 5, 5], [-9, 2, 0.1], [0.1, 0.1, 40]], np.float32)):
        pool, opool = pkg.Pool(), oracle.Pool()
        col = (np.arange(pts.shape[0] * 3) % 251).astype(np.uint8).reshape(-1, 3)
        for _ in range(2):
            gpu_insert(pkg, torch, ws, pool, pts, col, 6, center, edge)
            opool.insert_cloud(pts, col, 6, center, edge)
            assert_pools_equal(pool, opool)
    # depth out of range is refused
    with pytest.raises(pkg.SvoslamError):
        gpu_insert(pkg, torch, ws, pool, pts, col, 17, center, edge)


def test_alpha_saturation_and_q4(env, oracle):
    """re-observing the same cloud: alpha += 2 up to 255, octant-7 leaves split exactly once"""
    pkg, torch = env
    rng = np.random.default_rng(9)
    pts, col = random_cloud(rng, 3000)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    sizes = []
    for f in range(70):
        gpu_insert(pkg, torch, ws, pool, pts, col, 4, (0, 0, 0), 1.0)
        sizes.append(pool.size)
        if f in (0, 1, 2, 69):
            opool.insert_cloud(pts, col, 4, (0, 0, 0), 1.0)
            if f < 3:
                assert_pools_equal(pool, opool)
        elif f < 69:
            opool.insert_cloud(pts, col, 4, (0, 0, 0), 1.0)
    assert_pools_equal(pool, opool)
    assert sizes[1] > sizes[0] and sizes[2] == sizes[1]  # Q4 growth happens once
    assert (pool.words()[1::2] >> 24).max() == 255


def test_voxel_grid_path(env, oracle):
    pkg, torch = env
    rng = np.random.default_rng(21)
    n, depth, center, edge = 6000, 6, (0.0, 0.1, 0.0), 0.5
    ce = np.ones((n, 4), np.float32)
    ce[:, :3] = (rng.random((n, 3)) - 0.5) * 0.9 + np.array(center)
    ce[::31, 0] = np.nan
    co = rng.random((n, 4)).astype(np.float32)
    co[::17, :3] = 1.0  # Q21: 1.0*256 overflows into the next channel when alpha == 0
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    for _ in range(2):
        pkg.svo_from_voxel_grid(ws, torch.from_numpy(ce).cuda(), torch.from_numpy(co).cuda(), depth, pool, center, edge)
        opool.insert_voxel_grid(ce, co, depth, center, edge)
        assert_pools_equal(pool, opool)


@pytest.mark.parametrize("depth", [3, 6, 9])
def test_extract_matches_oracle(env, oracle, depth):
    pkg, torch = env
    rng = np.random.default_rng(33 + depth)
    pts, col = surface_cloud(rng, 15000)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    for _ in range(2):
        gpu_insert(pkg, torch, ws, pool, pts, col, depth, (0, 0, 0), 1.0)
        opool.insert_cloud(pts, col, depth, (0, 0, 0), 1.0)
    for d in (depth, max(1, depth - 2)):
        ce, co = pkg.extract_voxel_grid(ws, pool, d, (0, 0, 0), 1.0)
        rce, rco = opool.extract(d, (0, 0, 0), 1.0)
        assert ce.shape == rce.shape and ce.shape[0] > 0
        assert np.array_equal(ce.view(np.uint32), rce.view(np.uint32))
        assert np.array_equal(co.view(np.uint32), rco.view(np.uint32))


def test_large_cloud_invariants(env):
    """full-size (640x480 points, depth 12) properties that need no oracle: tree invariants,
    idempotent structure on re-insertion, every child pointer in range and tile aligned"""
    pkg, torch = env
    rng = np.random.default_rng(77)
    n, depth = 640 * 480, 12
    pts, col = surface_cloud(rng, n, jitter=0.002)
    ws, pool = pkg.Workspace(), pkg.Pool()
    st1 = gpu_insert(pkg, torch, ws, pool, pts, col, depth, (0, 0, 0), 1.024)
    w = pool.words()
    w0 = w[0::2]
    flagged = (w0 & 0x40000000) != 0
    child = w0[flagged] & 0x3FFFFFFF
    assert st1.pool_size_after == 8 + 8 * st1.num_split == pool.size
    assert flagged.sum() == st1.num_split
    assert (child % 8 == 0).all() and child.min() >= 8 and child.max() + 8 <= pool.size
    assert np.unique(child).size == child.size          # every tile has exactly one parent
    assert (w0[~flagged] == 0).all()
    size1 = pool.size
    st2 = gpu_insert(pkg, torch, ws, pool, pts, col, depth, (0, 0, 0), 1.024)
    w2 = pool.words()
    assert (w2[0:2 * size1:2] & 0x3FFFFFFF == w[0::2] & 0x3FFFFFFF)[flagged].all()  # old pointers untouched
    st3 = gpu_insert(pkg, torch, ws, pool, pts, col, depth, (0, 0, 0), 1.024)
    assert st3.num_split == 0 and st2.num_split < st1.num_split  # only the one-off Q4 splits in pass 2


_BIG_CHILD = r"""
import hashlib, sys
import numpy as np, torch
import svoslam_pkg
pkg = svoslam_pkg.load()
n, depth, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
g = torch.Generator(device="cuda"); g.manual_seed(4242)
pts = torch.rand((n, 3), generator=g, device="cuda", dtype=torch.float32) * 1.9 - 0.95
pts[::1001] = float("nan")
col = torch.randint(0, 256, (n, 3), generator=g, device="cuda", dtype=torch.uint8)
ws, pool = pkg.Workspace(), pkg.Pool()
if mode == "grid":     # svoFromVoxelGrid: keys alone are sorted, colour i stays with sorted key i (Q20)
    ce = torch.cat([torch.nan_to_num(pts, nan=0.0), torch.ones((n, 1), device="cuda")], 1).contiguous()
    co = torch.cat([col.float() / 255.0, torch.ones((n, 1), device="cuda")], 1).contiguous()
    pkg.svo_from_voxel_grid(ws, ce, co, depth, pool, (0, 0, 0), 1.0)
else:
    pkg.svo_from_point_cloud(ws, pts, col, depth, pool, (0, 0, 0), 1.0)
torch.cuda.synchronize()
print("DIGEST", pool.size, hashlib.sha256(pool.words().tobytes()).hexdigest())
"""


@pytest.mark.parametrize("n,depth,mode", [(20_000_000, 7, "cloud"), (70_000_000, 7, "cloud"), (70_000_000, 8, "grid")])
def test_large_inputs_packed_sort_equals_pair_sort(env, n, depth, mode):
    """the blocking insert's packed sort at sizes no oracle run reaches -- 9-bit digits and the column scan in chunks (20 M points:
    9766 tiles), 8-bit digits (70 M), keys alone on the voxel-grid path -- against the independent (key, index) PAIR sort of round 1
    (svoslam_config.sort_pairs = 1) in a child process: the pools (node indices, colours of ten-fold duplicated leaves: the
    lowest point index wins) must be identical"""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    digests = []
    for cfg in ("", "sort_pairs=1"):
        e = dict(os.environ, SVOSLAM_CONFIG=cfg, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", _BIG_CHILD, str(n), str(depth), mode], env=e, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        digests.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert digests[0] == digests[1], digests
    assert int(digests[0].split()[1]) > 100000


@pytest.mark.parametrize("depth,n,frames", [(2, 500, 3), (6, 20000, 4), (10, 40000, 4), (12, 60000, 3), (16, 20000, 2)])
def test_async_fusion_matches_oracle(env, oracle, depth, n, frames):
    """the asynchronous entry point (no readback, all splits in one launch) builds the same pool"""
    pkg, torch = env
    rng = np.random.default_rng(700 + depth)
    ws, pool = pkg.Workspace(), pkg.Pool()
    opool = oracle.Pool()
    center, edge = (0.05, -0.02, 0.01), 1.0
    for f in range(frames):
        pts, col = (surface_cloud(rng, n) if f % 2 == 0 else random_cloud(rng, n, nan_every=53, dup_frac=0.1))
        pts = pts + np.float32(0.003 * f)
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        pkg.svo_from_point_cloud_async(ws, tp, tc, depth, pool, center, edge)
        opool.insert_cloud(pts, col, depth, center, edge)
        if f % 2 == 1 or f == frames - 1:      # leave some calls un-synchronised in between
            assert_pools_equal(pool, opool)
    # mixing with the blocking entry point keeps working
    pts, col = surface_cloud(rng, n)
    st = gpu_insert(pkg, torch, ws, pool, pts, col, depth, center, edge)
    opool.insert_cloud(pts, col, depth, center, edge)
    assert st.pool_size_after == opool.size
    assert_pools_equal(pool, opool)


def test_async_fusion_many_frames_without_sync(env, oracle):
    pkg, torch = env
    rng = np.random.default_rng(808)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    for f in range(25):
        pts, col = surface_cloud(rng, 15000, jitter=0.004)
        pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), 9, pool, (0, 0, 0), 1.0)
        opool.insert_cloud(pts, col, 9, (0, 0, 0), 1.0)
    assert_pools_equal(pool, opool)


def test_async_fusion_long_runs_of_duplicates_and_invalid_points(env, oracle):
    """runs of equal keys and of rejected points longer than a fill workgroup (512 sorted keys): whole workgroups of
    the commit's leaf kernel without a single head, so the 'first head after this workgroup' is several workgroups
    away (its fallback search), and straddling nodes whose run continues across them"""
    pkg, torch = env
    rng = np.random.default_rng(4242)
    ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
    center, edge, depth = (0.0, 0.0, 0.0), 1.0, 9
    for f in range(3):
        base, col = surface_cloud(rng, 6000)
        heavy = base[rng.integers(0, 6000, 6)]                       # six points repeated 700-1900 times each
        reps = [np.repeat(h[None, :], int(k), 0) for h, k in zip(heavy, (700, 1100, 1900, 800, 1300, 1000))]
        bad = np.full((1500, 3), np.nan, np.float32)                # a long run of rejected points (key 1)
        outside = np.full((900, 3), 5.0, np.float32)                # outside the root cube
        pts = np.concatenate([base] + reps + [bad, outside]).astype(np.float32)
        pts += np.float32(0.002 * f)
        colors = rng.integers(0, 256, (len(pts), 3), dtype=np.uint8)
        perm = rng.permutation(len(pts))
        pts, colors = pts[perm], colors[perm]
        pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(colors).cuda(), depth, pool, center, edge)
        opool.insert_cloud(pts, colors, depth, center, edge)
        assert_pools_equal(pool, opool)


@pytest.mark.parametrize("depth,n,frames", [(2, 500, 3), (6, 20000, 4), (10, 40000, 4), (12, 60000, 3), (16, 20000, 2)])
def test_deferred_commit_then_apply_matches_oracle(env, oracle, depth, n, frames):
    """svoslam_svo_fuse_commit_deferred leaves every node below the pool's size as it was (what a concurrent ray march
    reads), svoslam_svo_fuse_apply then makes the pool the oracle's; a render between the two sees the old map"""
    pkg, torch = env
    rng = np.random.default_rng(900 + depth)
    ws, pool = pkg.Workspace(), pkg.Pool()
    opool = oracle.Pool()
    center, edge = (0.05, -0.02, 0.01), 1.0
    view = oracle.look_at((0.2, 0.3, -2.2), (0, 0, 0), (0, 1, 0))
    for f in range(frames):
        pts, col = (surface_cloud(rng, n) if f % 2 == 0 else random_cloud(rng, n, nan_every=53, dup_frac=0.1))
        pts = pts + np.float32(0.003 * f)
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        if f == 0:  # (the first fusion creates the pool)
            pkg.svo_from_point_cloud_async(ws, tp, tc, depth, pool, center, edge)
            opool.insert_cloud(pts, col, depth, center, edge)
            continue
        size_before = pool.size
        before = pool.words().copy()
        img_before = torch.zeros((30, 40, 4), dtype=torch.uint8, device="cuda")
        pkg.cone_trace_svo(img_before, 45.0, view, pool.data_ptr, center, edge, 1)
        pkg.svo_fuse_sort(ws, tp, depth, center, edge)
        pkg.svo_fuse_plan(ws, n, depth, pool)
        pkg.svo_fuse_commit_deferred(ws, tc, depth, pool)
        img_mid = torch.zeros((30, 40, 4), dtype=torch.uint8, device="cuda")
        pkg.cone_trace_svo(img_mid, 45.0, view, pool.data_ptr, center, edge, 1)
        torch.cuda.synchronize()
        mid = pool.words()[: 2 * size_before]
        assert np.array_equal(mid, before[: 2 * size_before])       # nothing visible yet
        assert torch.equal(img_mid, img_before)
        with pytest.raises(pkg.SvoslamError):                      # one deferred commit at a time
            pkg.svo_fuse_plan(ws, n, depth, pool); pkg.svo_fuse_commit(ws, tc, depth, pool)
        pkg.svo_fuse_apply(ws, pool)
        opool.insert_cloud(pts, col, depth, center, edge)
        assert_pools_equal(pool, opool)


@pytest.mark.parametrize("depth,n,frames", [(1, 300, 3), (2, 500, 3), (6, 20000, 4), (10, 40000, 4), (12, 60000, 3), (16, 20000, 2)])
def test_early_split_then_commit_matches_oracle(env, oracle, depth, n, frames):
    """sort -> plan -> split_early -> commit: the splits' child tiles are written ahead of the commit (beyond the pool's
    size), the commit's leaf kernel writes the links (Q4 links of octant-7 leaves included): same pool as the oracle;
    and between split_early and commit the pool still renders / reads as before the commit"""
    pkg, torch = env
    rng = np.random.default_rng(900 + depth)
    ws, pool = pkg.Workspace(), pkg.Pool(1 << 22)
    opool = oracle.Pool()
    center, edge = (0.05, -0.02, 0.01), 1.0
    for f in range(frames):
        pts, col = (surface_cloud(rng, n) if f % 2 == 0 else random_cloud(rng, n, nan_every=53, dup_frac=0.1))
        pts = pts + np.float32(0.003 * f)
        if f == frames - 1:
            pts[: n // 4] = np.abs(pts[: n // 4])        # many keys ending in octant 7 paths (Q4 leaves gain children)
        tp, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
        pkg.svo_fuse_sort(ws, tp, depth, center, edge)
        pkg.svo_fuse_plan(ws, n, depth, pool)
        before = pool.words().copy() if f > 0 else None
        pkg.svo_fuse_split_early(ws, n, depth, pool)
        if before is not None:
            torch.cuda.synchronize()
            assert np.array_equal(pool.words(), before)   # nothing a reader of the pool can see has changed
        pkg.svo_fuse_commit(ws, tc, depth, pool)
        opool.insert_cloud(pts, col, depth, center, edge)
        assert_pools_equal(pool, opool)
    # a plain commit on the same workspace afterwards is not affected
    pts, col = surface_cloud(rng, n)
    pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), depth, pool, center, edge)
    opool.insert_cloud(pts, col, depth, center, edge)
    assert_pools_equal(pool, opool)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4097, 6145])
def test_async_fusion_sizes_around_the_sort_and_plan_tiles(env, oracle, n):
    """the asynchronous path (packed sort: tiles of 2048 keys; plan tiles of 512; fill tiles of 512) with element counts at
    and around every tile boundary, ragged last tiles, all-invalid batches and a single point, at depths whose packed word
    uses few (depth 1: 4 + idx bits) and many (depth 16: 49 + idx bits, exactly 64 at n = 2^15) bits"""
    pkg, torch = env
    rng = np.random.default_rng(31 + n)
    for depth in (1, 3, 9, 16):
        ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
        center, edge = (0.01, 0.02, -0.01), 1.0
        for f in range(3):
            pts, col = random_cloud(rng, n, nan_every=7 if n > 20 else 0, dup_frac=0.2)
            if f == 2:
                pts[:] = np.nan                      # a batch without a single valid point
            pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), depth, pool, center, edge)
            opool.insert_cloud(pts, col, depth, center, edge)
            assert_pools_equal(pool, opool)


def test_async_fusion_packed_word_exactly_64_bits(env, oracle):
    """3 x 16 + 1 key bits + 15 index bits = 64: the largest batch the packed sort takes at depth 16 (one more point
    falls back to the pair sort); both must build the oracle's pool"""
    pkg, torch = env
    rng = np.random.default_rng(77)
    for n in (1 << 15, (1 << 15) + 1):
        ws, pool, opool = pkg.Workspace(), pkg.Pool(), oracle.Pool()
        for f in range(2):
            pts, col = surface_cloud(rng, n, jitter=0.002)
            pkg.svo_from_point_cloud_async(ws, torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), 16, pool, (0, 0, 0), 1.0)
            opool.insert_cloud(pts, col, 16, (0, 0, 0), 1.0)
        assert_pools_equal(pool, opool)


@pytest.mark.parametrize("depth,n", [(2, 500), (9, 30000), (12, 60000), (16, 20000)])
def test_structure_chain_equals_sequential_fusion(env, oracle, depth, n):
    """svoslam_svo_fuse_plan_structure: plans + splits of three frames run AHEAD of every commit (one stream each, the
    commits ordered only after their own plan), then the commits in frame order: the oracle's pool"""
    pkg, torch = env
    rng = np.random.default_rng(1200 + depth)
    center, edge = (0.02, -0.01, 0.03), 1.0
    pool, opool = pkg.Pool(1 << 23), oracle.Pool()
    wss = [pkg.Workspace() for _ in range(3)]
    s_struct, s_col = torch.cuda.Stream(), torch.cuda.Stream()
    for rnd in range(2):
        clouds = []
        for f in range(3):
            pts, col = (surface_cloud(rng, n) if f != 1 else random_cloud(rng, n, nan_every=41, dup_frac=0.15))
            pts = pts + np.float32(0.004 * (3 * rnd + f))
            if f == 2:
                pts[: n // 5] = np.abs(pts[: n // 5])
            clouds.append((torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda(), pts, col))
        torch.cuda.synchronize()
        evs = []
        with torch.cuda.stream(s_struct):
            pkg.pool_structure_begin(pool)
            for f in range(3):
                pkg.svo_fuse_sort(wss[f], clouds[f][0], depth, center, edge)
                pkg.svo_fuse_plan_structure(wss[f], n, depth, pool)
                e = torch.cuda.Event(); e.record(); evs.append(e)
        with torch.cuda.stream(s_col):
            for f in range(3):
                s_col.wait_event(evs[f])
                pkg.svo_fuse_commit(wss[f], clouds[f][1], depth, pool)
        torch.cuda.synchronize()
        for f in range(3):
            opool.insert_cloud(clouds[f][2], clouds[f][3], depth, center, edge)
        assert_pools_equal(pool, opool)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_band_sort_merge_equals_whole_frame_sort(env, oracle, world):
    """SURVEY 8e's sharded fusion on the device: every band's keys are computed and sorted on their own
    (svoslam_svo_fuse_sort_frame_band, whole-image pixel indices), the lists are merged (svoslam_svo_fuse_merge_sorted) and
    adopted: the merged list equals the whole-frame sort bit for bit, and plan + commit on it build the oracle's pool."""
    import importlib
    pkg, torch = env
    synth = importlib.import_module("octree_slam_amd.synth")
    pl = importlib.import_module("octree_slam_amd.pipeline")
    w, h, depth, center, edge = 320, 240 + (1 if world == 3 else 0), 10, (0.0, 1.5, 0.0), 4.096
    f = synth.focal_length(w)
    cam = pkg.Camera(w, h, f, f)
    opool, pool = oracle.Pool(), pkg.Pool()
    ocam = oracle.Camera(w, h, f, f)
    for k in range(3):
        d, c = synth.render_frame(k, w, h)
        dn, cn = d.numpy().view(np.uint16), c.numpy()
        dd, cc = d.cuda(), c.cuda()
        cam.update(dd, cc, k); ocam.update(dn, cn, k)
        pose = cam.fusion_transform_ptr()
        ws_full = pkg.Workspace()
        pkg.svo_fuse_sort_frame(ws_full, dd, pose, f, f, depth, center, edge)
        fk = torch.empty(w * h, dtype=torch.int64, device="cuda"); fi = torch.empty(w * h, dtype=torch.int32, device="cuda")
        pkg.svo_fuse_export_sorted(ws_full, w * h, fk, fi)
        ks, is_ = [], []
        for r in range(world):
            first, rows = pl.band_rows(h, r, world)
            wsb = pkg.Workspace()
            pkg.svo_fuse_sort_frame_band(wsb, dd, pose, f, f, depth, center, edge, first, rows)
            bk = torch.empty(rows * w, dtype=torch.int64, device="cuda"); bi = torch.empty(rows * w, dtype=torch.int32, device="cuda")
            pkg.svo_fuse_export_sorted(wsb, rows * w, bk, bi)
            ks.append(bk); is_.append(bi)
        mk = torch.empty(w * h, dtype=torch.int64, device="cuda"); mi = torch.empty(w * h, dtype=torch.int32, device="cuda")
        pkg.svo_fuse_merge_sorted(ks, is_, mk, mi)
        torch.cuda.synchronize()
        assert torch.equal(mk, fk) and torch.equal(mi, fi), k
        ws = pkg.Workspace()
        pkg.svo_fuse_adopt_sorted(ws, mk, mi, depth)
        pkg.svo_fuse_plan(ws, w * h, depth, pool)
        pkg.svo_fuse_commit(ws, cc.view(-1, 3), depth, pool)
        v = oracle.transform_vertex_map(oracle.vertex_map(dn, f, f, w, h), ocam.fusion_transform())
        opool.insert_cloud(v.reshape(-1, 3), cn.reshape(-1, 3), depth, center, edge)
        assert_pools_equal(pool, opool)
