"""world_size-2 CPU test (gloo) of the N > 1 path: row-band partition, exact all-reduce of the
ICP fixed-point sums, all-gather of point bands.  The band partials come from the CPU oracle (the
GPU kernels cannot run here); what is under test is the sharding/communication logic that
pipeline.SlamPipeline uses unchanged on RCCL."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, h, w, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svoslam_pkg
    svoslam_pkg.load()
    pl = importlib.import_module("octree_slam_amd.pipeline")
    from oracle import oracle as ora
    ctx = pl.DistContext(rank, world)
    first, rows = pl.band_rows(h, rank, world)
    # identical inputs on every rank (same seed), as in bench.py
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w]
    d = (1500 + 200 * np.sin(xx / 9.0) + 100 * np.cos(yy / 7.0) + rng.normal(scale=1.0, size=(h, w))).astype(np.uint16)
    f = 570.3 * w / 640
    v1 = ora.vertex_map(d, f, f, w, h); n1 = ora.normal_map(v1)
    T = ora.icp_update_transform(np.array([0.003, -0.002, 0.001, 0.003, -0.002, 0.001], np.float32))
    v2 = ora.transform_vertex_map(v1, T); n2 = ora.transform_normal_map(n1, T)
    # 1) ICP: band partial -> all-reduce(sum, float64) == full-image sums, bit for bit
    part = ora.icp_cost2_raw(v1, n1, v2, n2, first * w, rows * w).astype(np.float64)
    acc = torch.from_numpy(part.copy())
    ctx.all_reduce_sum(acc)
    full = ora.icp_cost2_raw(v1, n1, v2, n2).astype(np.float64)
    ok_icp = bool(np.array_equal(acc.numpy(), full)) and bool(np.abs(full).max() > 0)
    A, b = ora.icp_finish(acc.numpy().astype(np.int64))
    rA, rb = ora.icp_cost2(v1, n1, v2, n2)
    ok_icp = ok_icp and np.array_equal(A, rA) and np.array_equal(b, rb)
    # 2) point bands: each rank fills only its rows, all-gather restores the full map
    pts = torch.full((h, w, 3), float("nan"), dtype=torch.float32)
    pts[first:first + rows] = torch.from_numpy(v2[first:first + rows])
    ctx.all_gather_rows(pts, h)
    ok_gather = bool(np.array_equal(np.nan_to_num(pts.numpy(), nan=-7, posinf=9), np.nan_to_num(v2, nan=-7, posinf=9)))
    # 3) every rank then holds identical fusion input -> identical pool (replicas byte-identical)
    pool = ora.Pool()
    col = (np.arange(h * w * 3) % 251).astype(np.uint8).reshape(-1, 3)
    pool.insert_cloud(pts.numpy().reshape(-1, 3), col, 6, (0, 0, 1.5), 2.0)
    digest = torch.tensor([int(pool.words().astype(np.uint64).sum() % (1 << 62)), pool.size], dtype=torch.int64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    ok_pool = all(bool((g == digest).all()) for g in gathered)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array([ok_icp, ok_gather, ok_pool, first, rows]))
    dist.destroy_process_group()


@pytest.mark.parametrize("h,w", [(48, 64), (45, 64)])
def test_two_rank_bands_gloo(tmp_path, h, w):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, h, w, str(tmp_path)), nprocs=world, join=True)
    covered = 0
    for r in range(world):
        ok_icp, ok_gather, ok_pool, first, rows = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert ok_icp and ok_gather and ok_pool, (r, ok_icp, ok_gather, ok_pool)
        assert first == covered
        covered += rows
    assert covered == h


def merge_rule(keys_lists, idx_lists):
    """the rank formula of merge_sorted_kernel (csrc/svo_build.hip), restated with numpy searches: element e of list a goes
    to (its place in a) + sum_{b<a} #{keys of b <= e.key} + sum_{b>a} #{keys of b < e.key}"""
    total = sum(len(k) for k in keys_lists)
    out_k, out_i = np.zeros(total, np.int64), np.zeros(total, np.int64)
    for a, (ka, ia) in enumerate(zip(keys_lists, idx_lists)):
        rank = np.arange(len(ka))
        for b, kb in enumerate(keys_lists):
            if b != a:
                rank = rank + np.searchsorted(kb, ka, side="right" if b < a else "left")
        out_k[rank], out_i[rank] = ka, ia
    return out_k, out_i


def _worker_keys(rank, world, port, h, w, out_dir):
    """SURVEY 8e's sharded fusion on the CPU: every rank computes and sorts the keys of its row band (oracle keys, a stable
    sort by (key, pixel)), the sorted lists are all-gathered through pipeline.DistContext.all_gather_sorted (padded bands,
    as the GPU path sends them), merged by the device kernel's rank rule, and must equal the one-rank stable sort of the
    whole frame on EVERY rank -- the list every replica numbers its new nodes from (num_nodes + 8 x rank in that list)."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svoslam_pkg
    svoslam_pkg.load()
    pl = importlib.import_module("octree_slam_amd.pipeline")
    from oracle import oracle as ora
    ctx = pl.DistContext(rank, world)
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    d = (1400 + 150 * np.sin(xx / 6.0) + 90 * np.cos(yy / 5.0) + rng.normal(scale=1.0, size=(h, w))).astype(np.uint16)
    d[rng.random((h, w)) < 0.05] = 0                       # dropouts: invalid points (key 1), sorted first
    d[10:14] = d[10]                                       # equal rows -> many duplicate keys ACROSS bands
    f = 570.3 * w / 640
    v = ora.vertex_map(d, f, f, w, h).reshape(-1, 3)
    depth, center, edge = 5, (0.0, 0.0, 1.5), 2.0
    keys = ora.compute_keys(v, depth, center, edge).astype(np.int64)
    n = h * w
    want = np.lexsort((np.arange(n), keys))                # one stable sort of the whole frame
    first, rows = pl.band_rows(h, rank, world)
    lo, nb = first * w, rows * w
    order = np.lexsort((np.arange(lo, lo + nb), keys[lo:lo + nb]))
    base, rem = divmod(h, world)
    pad = (base + (1 if rem else 0)) * w
    mine_k, mine_i = torch.zeros(pad, dtype=torch.int64), torch.zeros(pad, dtype=torch.int32)
    mine_k[:nb] = torch.from_numpy(keys[lo:lo + nb][order]); mine_i[:nb] = torch.from_numpy((order + lo).astype(np.int32))
    all_k, all_i = torch.empty((world, pad), dtype=torch.int64), torch.empty((world, pad), dtype=torch.int32)
    ctx.all_gather_sorted(all_k, mine_k); ctx.all_gather_sorted(all_i, mine_i)
    counts = [pl.band_rows(h, r, world)[1] * w for r in range(world)]
    mk, mi = merge_rule([all_k[r, :counts[r]].numpy() for r in range(world)], [all_i[r, :counts[r]].numpy() for r in range(world)])
    ok = bool(np.array_equal(mk, keys[want])) and bool(np.array_equal(mi, want))
    dup = int((np.diff(mk) == 0).sum())
    np.save(os.path.join(out_dir, "keys_rank%d.npy" % rank), np.array([ok, dup, int((mk == 1).sum())]))
    dist.destroy_process_group()


@pytest.mark.parametrize("h,w,world", [(48, 64, 2), (45, 64, 2), (50, 32, 3)])
def test_band_key_lists_merge_gloo(tmp_path, h, w, world):
    port = _free_port()
    mp.spawn(_worker_keys, args=(world, port, h, w, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, dup, invalid = np.load(os.path.join(str(tmp_path), "keys_rank%d.npy" % r))
        assert ok, r
        assert dup > 0 and invalid > 0          # the case had duplicate keys and invalid points to get right


def test_band_rows_partition():
    import sys
    sys.path.insert(0, ROOT)
    import svoslam_pkg
    svoslam_pkg.load()
    pl = importlib.import_module("octree_slam_amd.pipeline")
    for h in (480, 1080, 2160, 7, 61):
        for world in (1, 2, 3, 4, 8):
            bands = [pl.band_rows(h, r, world) for r in range(world)]
            assert bands[0][0] == 0 and sum(b[1] for b in bands) == h
            for (f0, n0), (f1, _) in zip(bands[:-1], bands[1:]):
                assert f0 + n0 == f1
            assert max(b[1] for b in bands) - min(b[1] for b in bands) <= 1


# ---- frame-sharded tracking (exchange "deltas") -----------------------------------------
def _delta_worker(rank, world, port, n, w, h, per_rank, out_dir):
    """Each rank tracks the frames g with g % world == rank as PAIRS (g-1, g) on a scratch oracle camera, the update_trans
    records are all-gathered chunk by chunk with pipeline.frame_shards' layout, every rank composes the pose chain with
    apply_delta: poses, fusion transforms and lost counts must equal the sequential oracle camera's, bit for bit."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svoslam_pkg
    svoslam_pkg.load()
    pl = importlib.import_module("octree_slam_amd.pipeline")
    synth = importlib.import_module("octree_slam_amd.synth")
    from oracle import oracle as ora
    ctx = pl.DistContext(rank, world, exchange="deltas")
    depth, rgb = synth.render_stream(n, w, h, device="cpu")
    depth = depth.numpy().view(np.uint16).copy()
    depth[4, :, : w // 2] = 0  # a frame with half its pixels missing: ICP on what is left
    depth[5] = 0               # a frame without any measurement: frames 5 and 6 abandon every level (singular systems)
    rgb = rgb.numpy()
    f = synth.focal_length(w)
    FL = 20                    # == octree_slam_amd.DELTA_FLOATS
    cam = ora.Camera(w, h, f, f)
    poses = []
    for (a, b, slots) in pl.frame_shards(n, 0, world, per_rank):
        mine = torch.zeros((per_rank, FL), dtype=torch.float32)
        for i in range(a, b):
            r, row = slots[i - a]
            if r == rank and i > 0:
                scratch = ora.Camera(w, h, f, f)
                scratch.update(depth[i - 1], rgb[i - 1], 0)
                scratch.update(depth[i], rgb[i], 1)
                rec = np.zeros(FL, np.float32)
                rec[:16] = scratch.last_update()
                rec[16:17] = np.array([scratch.tracking_lost_count()], np.int32).view(np.float32)
                mine[row] = torch.from_numpy(rec)
        allr = torch.empty((world, per_rank, FL), dtype=torch.float32)
        ctx.all_gather_deltas(allr, mine)
        for i in range(a, b):
            r, row = slots[i - a]
            rec = allr[r, row].numpy()
            cam.apply_delta(None if i == 0 else rec[:16], int(rec[16:17].view(np.int32)[0]), i)
            p, o = cam.pose()
            poses.append(np.concatenate([p, o, cam.fusion_transform(), [cam.tracking_lost_count()]]))
    ref = ora.Camera(w, h, f, f)
    want = []
    for i in range(n):
        ref.update(depth[i], rgb[i], i)
        p, o = ref.pose()
        want.append(np.concatenate([p, o, ref.fusion_transform(), [ref.tracking_lost_count()]]))
    got, want = np.array(poses, np.float32), np.array(want, np.float32)
    ok = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    moved = bool(np.abs(want[-1][3:12] - want[3][3:12]).max() > 0) and bool(want[-1][-1] == 6)   # tracked again after the gap
    np.save(os.path.join(out_dir, "delta_rank%d.npy" % rank), np.array([ok, moved]))
    dist.destroy_process_group()


@pytest.mark.parametrize("per_rank", [1, 3])
def test_two_rank_frame_sharded_tracking_gloo(tmp_path, per_rank):
    world, port = 2, _free_port()
    mp.spawn(_delta_worker, args=(world, port, 8, 160, 120, per_rank, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, moved = np.load(os.path.join(str(tmp_path), "delta_rank%d.npy" % r))
        assert ok and moved, (r, ok, moved)


def test_frame_shards_cover_every_frame_once():
    import sys
    sys.path.insert(0, ROOT)
    import svoslam_pkg
    svoslam_pkg.load()
    pl = importlib.import_module("octree_slam_amd.pipeline")
    for n in (1, 5, 16, 37):
        for world in (1, 2, 3, 8):
            for per_rank in (1, 2, 5):
                for first in (0, 3):
                    seen, covered = set(), 0
                    for (a, b, slots) in pl.frame_shards(n, first, world, per_rank):
                        assert a == covered and b - a == len(slots) <= world * per_rank
                        covered = b
                        for i, (r, row) in zip(range(a, b), slots):
                            assert r == (first + i) % world and 0 <= row < per_rank
                            assert (a, r, row) not in seen          # one record per slot of the gathered block
                            seen.add((a, r, row))
                    assert covered == n
