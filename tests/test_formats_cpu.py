"""The CPU restatement of the linear-tree formats (oracle/formats.py) on oracle-built pools: the reference's
reader (`pullFromLinearTree`, octree.cpp:151-167) reads our pool layout; its writer's layout (`addToLinearTree`,
:118-149, with the flag the reader needs) round-trips through the reader; the checkpoint container and the sub-tree
paging file round-trip and refuse damaged input; re-rooting keeps every node's place in space.  The GPU side
(tests/test_gpu_io.py) compares what libsvoslam_hip writes with these restatements."""
import numpy as np
import pytest

from oracle import formats as fm
from util import surface_cloud


@pytest.fixture(scope="module")
def words(oracle):
    rng = np.random.default_rng(77)
    pool = oracle.Pool()
    for k in range(2):
        pts, col = surface_cloud(rng, 3000)
        pool.insert_cloud(pts + np.float32(0.02 * k), col, 6, (0, 0, 0), 1.0)
    return pool.words()


def test_reader_reads_the_pool_and_writer_round_trips(words):
    tree = fm.pull_to_cpu(words)                       # the pool IS a stackless array of the root node
    n_nodes = sum(fm.count_nodes(t) for t in tree)
    assert n_nodes == words.size // 2                  # every node of the pool is reachable exactly once
    lin = fm.add_to_linear_tree(tree)                  # the reference writer's numbering: depth first, with gaps
    assert not np.array_equal(lin[:words.size], words) or lin.size != words.size   # a different layout ...
    assert fm.pull_to_cpu(lin) == tree                 # ... of the same tree
    # children flags and values survive: spot-check the first flagged top node against the raw words
    i = next(k for k in range(8) if words[2 * k] & fm.FLAG)
    off = int(words[2 * i] & fm.MASK)
    assert tree[i][0] == int(words[2 * i + 1]) and tree[i][1][3][0] == int(words[2 * (off + 3) + 1])


def test_reader_refuses_a_child_index_outside_the_array(words):
    bad = words.copy()
    i = next(k for k in range(8) if bad[2 * k] & fm.FLAG)
    bad[2 * i] = fm.FLAG | (bad.size // 2)             # first child one past the end
    with pytest.raises(ValueError):
        fm.pull_to_cpu(bad)
    with pytest.raises(ValueError):
        fm.pull_to_cpu(words[:8])                      # fewer than 8 nodes (octree.cpp:88-91)


def test_checkpoint_container_round_trip(words, tmp_path):
    p = tmp_path / "a.svopool"
    fm.write_pool_file(p, words, (0.1, -0.05, 0.02), 1.25, 9)
    w, c, e, d = fm.read_pool_file(p)
    assert np.array_equal(w, words) and c == tuple(float(np.float32(x)) for x in (0.1, -0.05, 0.02)) and e == 1.25 and d == 9
    raw = bytearray(p.read_bytes())
    assert raw[:8] == b"SVOPOOL1" and len(raw) == 64 + 4 * words.size
    raw[200] ^= 1
    (tmp_path / "bad").write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        fm.read_pool_file(tmp_path / "bad")
    (tmp_path / "short").write_bytes(bytes(raw[:100]))
    with pytest.raises(ValueError):
        fm.read_pool_file(tmp_path / "short")


def test_subtree_evict_restore(words, tmp_path):
    tree = fm.pull_to_cpu(words)
    top = next(k for k in range(8) if words[2 * k] & fm.FLAG)
    tiles, blob, after, node = fm.evict_subtree(words, [top])
    assert node == top and tiles[0] == (words[2 * top] & fm.MASK)
    # the blob is a stackless array of its own: the reference's reader gives the evicted node's children
    assert fm.pull_to_cpu(blob) == tree[top][1]
    # the pool afterwards: the node is a leaf with its colour, everything else in place
    t2 = fm.pull_to_cpu(after)
    assert t2[top] == (tree[top][0], None) and all(t2[k] == tree[k] for k in range(8) if k != top)
    f = tmp_path / "s.svosub"
    fm.write_subtree_file(f, [top], node, words.size // 2, tiles, blob)
    sub = fm.read_subtree_file(f)
    assert sub["path"] == [top] and np.array_equal(sub["tiles"], tiles) and np.array_equal(sub["nodes"], blob)
    assert np.array_equal(fm.restore_subtree(after, sub), words)
    # two levels down, and refusals
    inner = next(k for k in range(8) if tree[top][1][k][1] is not None)
    tiles2, blob2, after2, node2 = fm.evict_subtree(words, [top, inner])
    assert fm.pull_to_cpu(blob2) == tree[top][1][inner][1]
    assert np.array_equal(fm.restore_subtree(after2, {"path": [top, inner], "node_index": node2, "tiles": tiles2, "nodes": blob2}), words)
    with pytest.raises(ValueError):
        fm.restore_subtree(words, sub)                 # the node has children (again)
    raw = bytearray(f.read_bytes()); raw[70] ^= 4
    (tmp_path / "bad").write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        fm.read_subtree_file(tmp_path / "bad")


def test_expand_root(words):
    tree = fm.pull_to_cpu(words)
    new, c2, e2 = fm.expand_root(tree, (0.0, 0.0, 0.0), 1.0, (5.0, -5.0, 5.0))
    assert c2 == (1.0, -1.0, 1.0) and e2 == 2.0
    # the old centre lies at -x, +y, -z of the new one: octant 0b010 holds the old root, with the mip value of its children
    assert [k for k in range(8) if new[k][1] is not None] == [2] and new[2][1] == tree
    assert new[2][0] == fm.average_children(tree) and all(new[k] == (0, None) for k in range(8) if k != 2)
