"""Key-range sharded fusion, CPU-proven: world-2 / world-3 gloo test of the PROTOCOL by which
a frame's plan + commit can be cut across ranks by key range instead of being replicated on every rank -- with the CPU oracle as the
per-rank worker (this pins what the ranks must compute and exchange; the HIP side -- csrc/svo_build.hip "key-range sharded commit", with ONE
all-gather per frame: the numbering comes from the bucket sizes inside the deltas -- is checked by tests/test_gpu_keyrange.py).

Protocol (every rank holds a byte-identical replica of the pool and the frame's keys):
 1. splitters: the distinct level-L prefixes of the frame's keys, cut into `world` contiguous runs of about equal key count; rank r owns
    the keys under its run (whole level-L subtrees: no node at or below level L is shared between ranks);
 2. rank r plans (svo.cu:179-237) and commits (expand :239-289, fill :291-382, mip :384-465) ITS keys on its replica;
 3. numbering: the reference numbers the new tiles of pass p by the rank of their key among the pass's sorted unique keys -- numeric order,
    i.e. depth-major, prefix order within a depth.  Owned keys (depth >= L) of different ranks interleave only by depth, so an
    all-gather of the counts [pass][depth] gives every rank the global index of each of its tiles; the few shared keys above the
    splitter level (depth < L: a young map) are all-gathered themselves and ranked in their union;
 4. all-gather of the deltas {node index, word0, word1} in global numbering: the nodes a rank changed or created;
 5. apply: a node at level >= L has one owner; of a tile created by several ranks (children of a shared node) the owner's non-empty
    node wins; word0 of a shared node: the children pointer (everyone computed the same); word1 of the shared nodes (levels < L) is
    recomputed from the merged children, level by level, then the root pass (Q6) -- mipmapNodes restricted to the top levels.
The merged pool must equal the one-rank pool byte for byte.  The test also records the bytes all-gathered per frame."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAG, MASK, EMPTY1 = 0x40000000, 0x3FFFFFFF, 127 << 24


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def key_depth(k):
    return (int(k).bit_length() - 1) // 3


def average_children(words, child_idx):
    """averageChildren (svo.cu:384-441) with Q5: always divides by 8; sums of 8 bytes and / 8 are exact"""
    w1 = words[2 * child_idx + 1: 2 * child_idx + 17: 2].astype(np.int64)
    r, g, b, a = (w1 & 0xFF).sum() // 8, ((w1 >> 8) & 0xFF).sum() // 8, ((w1 >> 16) & 0xFF).sum() // 8, ((w1 >> 24) & 0xFF).max()
    return int(r) | (int(g) << 8) | (int(b) << 16) | (int(a) << 24)


def walk(words, key):
    """node index of the node with prefix `key` (leading 1), and its children tile"""
    node, child = 0, 0
    digits = []
    k = int(key)
    while k != 1:
        digits.append(k & 7); k >>= 3
    for d in reversed(digits):
        node = child + d
        child = int(words[2 * node]) & MASK
    return node, child


def splitters(keys, depth, L, world):
    """owner rank of every key: contiguous runs of the sorted distinct level-L prefixes with about equal key counts"""
    valid = keys != 1
    pre = keys >> (3 * (depth - L))
    up, cnt = np.unique(pre[valid], return_counts=True)
    cum = np.cumsum(cnt)
    owner_of = np.minimum((cum - 1) * world // max(1, int(cum[-1])), world - 1)
    owner = np.full(keys.shape, -1, np.int64)
    owner[valid] = owner_of[np.searchsorted(up, pre[valid])]
    return owner


def _worker(rank, world, port, young, out_dir):
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as ora
    from util import surface_cloud
    depth, L, center, edge = 8, 3, (0.0, 0.0, 0.0), 1.0
    rng = np.random.default_rng(5)
    pts0, col0 = surface_cloud(rng, 6000)
    pts, col = surface_cloud(rng, 6000)
    pts = (pts * np.float32(0.97) + np.float32(0.013)).astype(np.float32)
    pts[::211] = np.nan                                            # invalid points belong to nobody (key 1)
    base = ora.Pool()
    if not young:
        base.insert_cloud(pts0, col0, depth, center, edge)         # a mature map: the top levels exist
    else:
        base.insert_cloud(pts0[:1], col0[:1], depth, center, edge)  # (a pool must exist: one point's path)
    w0 = base.words()
    n0 = base.size
    # ---- one rank: the whole frame
    ref = ora.Pool(); ref.load_words(w0)
    ref.insert_cloud(pts, col, depth, center, edge)
    want = ref.words()
    # ---- 1. splitters
    keys = ora.compute_keys(pts, depth, center, edge)
    owner = splitters(keys, depth, L, world)
    mine = owner == rank
    # ---- 2. plan + commit of the own keys on the own replica (local numbering)
    loc = ora.Pool(); loc.load_words(w0)
    total, sizes, codes = loc.prepass(keys[mine], depth)
    loc.insert_cloud(pts[mine], col[mine], depth, center, edge)
    lw = loc.words()
    passes, o = [], 0
    for p in range(depth):
        passes.append([int(c) for c in codes[o:o + sizes[p]]]); o += sizes[p]
    # ---- 3. numbering: counts [pass][depth] of the owned codes, the shared codes themselves
    cnt = np.zeros((depth, depth + 1), np.int64)
    top = [[c for c in ps if key_depth(c) < L] for ps in passes]
    for p, ps in enumerate(passes):
        for c in ps:
            if key_depth(c) >= L:
                cnt[p, key_depth(c)] += 1
    gathered = [None] * world
    dist.all_gather_object(gathered, {"cnt": cnt, "top": top})
    bytes_numbering = sum(g["cnt"].nbytes + 8 * sum(len(t) for t in g["top"]) for g in gathered)
    top_union = [sorted(set(c for g in gathered for c in g["top"][p])) for p in range(depth)]
    tot = np.zeros((depth, depth + 1), np.int64)                    # global count per (pass, depth)
    for p in range(depth):
        for c in top_union[p]:
            tot[p, key_depth(c)] += 1
        for d in range(L, depth + 1):
            tot[p, d] = sum(int(g["cnt"][p, d]) for g in gathered)
    pass_base = np.concatenate([[0], np.cumsum(tot.sum(1))])
    remap = {}                                                      # local tile -> global tile
    local_base = 0
    for p, ps in enumerate(passes):
        seen = np.zeros(depth + 1, np.int64)
        for j, c in enumerate(ps):
            d = key_depth(c)
            if d < L:
                within = top_union[p].index(c) - sum(1 for t in top_union[p] if key_depth(t) < d)
            else:
                within = sum(int(g["cnt"][p, d]) for g in gathered[:rank]) + int(seen[d])
                seen[d] += 1
            g_idx = int(pass_base[p]) + int(tot[p, :d].sum()) + within
            remap[n0 + 8 * (local_base + j)] = n0 + 8 * g_idx
        local_base += len(ps)
    n_total = n0 + 8 * int(pass_base[-1])

    def glob(i):                                                    # node index: local -> global
        return i if i < n0 else remap[i - (i - n0) % 8] + (i - n0) % 8

    def glob_w0(x):
        x = int(x)
        return x if not (x & FLAG) or (x & MASK) < n0 else FLAG | remap[x & MASK]

    # ---- 4. deltas in global numbering: changed old nodes, every node of the tiles created here
    delta = []
    old = np.nonzero((lw[:2 * n0:2] != w0[::2]) | (lw[1:2 * n0:2] != w0[1::2]))[0]
    for i in old:
        delta.append((int(i), glob_w0(lw[2 * i]), int(lw[2 * i + 1])))
    for i in range(n0, loc.size):
        delta.append((glob(i), glob_w0(lw[2 * i]), int(lw[2 * i + 1])))
    all_deltas = [None] * world
    dist.all_gather_object(all_deltas, np.array(delta, np.int64).reshape(-1, 3))
    bytes_deltas = sum(12 * len(d) for d in all_deltas)             # 4-byte index + two words per node
    # ---- 5. apply
    merged = np.zeros(2 * n_total, np.uint32)
    merged[:2 * n0] = w0
    merged[2 * n0 + 1::2] = EMPTY1                                   # a created tile starts as eight empty children (:269-275)
    for d in all_deltas:
        for i, a, b in d:
            i, a, b = int(i), int(a), int(b)
            if a & FLAG:
                merged[2 * i] = a                                   # (every rank that set it computed the same pointer)
            is_init = a == 0 and b == EMPTY1 and i >= n0
            if not is_init:
                merged[2 * i + 1] = b                               # one owner at levels >= L; the shared levels are redone below
    # word1 of the shared nodes (levels < L on the path of any key of the frame), level by level, then the root pass (Q6)
    valid = keys[keys != 1]
    for l in range(L - 1, 0, -1):
        for k in np.unique(valid >> (3 * (depth - l))):
            node, child = walk(merged, k)
            merged[2 * node + 1] = average_children(merged, child)
    if valid.size:
        merged[1] = average_children(merged, 0)
    ok = n_total == ref.size and bool(np.array_equal(merged, want))
    np.save(os.path.join(out_dir, "kr_rank%d.npy" % rank),
            np.array([int(ok), int(mine.sum()), n_total - n0, bytes_numbering, bytes_deltas, int(sum(len(t) for t in top_union)), ref.size - n0]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,young", [(2, False), (3, False), (2, True)])
def test_keyrange_sharded_commit_equals_one_rank(tmp_path, world, young):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, young, str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(os.path.join(str(tmp_path), "kr_rank%d.npy" % r)) for r in range(world)]
    print("world %d young %s: per rank [ok, keys owned, new nodes, numbering bytes all-gathered, delta bytes all-gathered, shared splits, new nodes (one rank)]" % (world, young),
          [r.tolist() for r in rows])
    for r in rows:
        assert r[0] == 1, rows                                      # merged pool == the one-rank pool, byte for byte
        assert r[2] == r[6] > 0                                     # same number of new nodes
    assert sum(int(r[1]) for r in rows) > 5000                      # every valid point has exactly one owner
    assert all(int(r[1]) > 1000 for r in rows)                      # ... and the runs are balanced enough to be worth it
    if young:
        assert rows[0][5] > 0                                       # the shared (top-level) splits were exercised
    else:
        assert rows[0][5] == 0
    # what crossed the wire: the numbering exchange is a few KB, the deltas ~12 bytes per touched node
    assert rows[0][3] < 64 * 1024 and rows[0][4] < 12 * 4 * (6000 * 9)
