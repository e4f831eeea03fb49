"""CPU restatement of the linear-tree formats of SURVEY 8f.2 -- TEST INFRASTRUCTURE ONLY (like everything under oracle/:
imported by tests/ alone, never by the product path).

What the reference defines, and what it does not:
  * `OctreeNode::pullFromLinearTree` (src/world/octree.cpp:151-167) is the READER of the "stackless" array and therefore
    the definition of the format: node i = words 2i, 2i+1; word0 bit 30 (0x40000000) = has children, low 30 bits = index
    of the first of 8 consecutive children inside the same array; word1 = the node's value.  `pull_from_linear_tree`
    below follows it literally (iteratively: the recursion is depth <= 16 but 8-way).
  * `OctreeNode::addToLinearTree` (:118-149) is the WRITER: depth-first, a node's 8 children at `offset`, the next free
    slot threaded through the recursion.  As shipped it never sets the children flag (it keeps the top two bits of
    whatever the malloc'ed array held, :122), so its output cannot be read back by the reader above: there is no byte
    stream to match.  `add_to_linear_tree` restates its layout WITH the flag the reader expects -- the one coherent
    reading -- so that writer -> reader round trips, and so that trees can be compared independent of node numbering.
  * `Octree::expandBySize` (:362-378) only rescales `size_` (Q16); what a correct re-rooting must do is stated in
    `expand_root` below in terms of the host tree the reader produces.
The product's own containers around that array -- the 64-byte checkpoint header of svoslam_pool_save and the sub-tree
file of svoslam_pool_evict_subtree (include/svoslam.h) -- are restated here field by field from their documentation, so
that tests/test_gpu_io.py compares what the HIP library writes with a CPU reading of it instead of with a HIP twin
(missing item 3).
"""
import struct

import numpy as np

FLAG = 0x40000000
MASK = 0x3FFFFFFF


# ------------------------------------------------------------------------------------------------ the reference's reader
def pull_from_linear_tree(words, position, _budget=None):
    """octree.cpp:151-167 for the node at `position`: returns the host tree as nested tuples (value, children) with
    children = None or a tuple of 8 such nodes."""
    words = np.asarray(words, dtype=np.uint32)
    n_nodes = words.size // 2
    # iterative post-order so that 16-level trees with millions of nodes do not recurse in Python
    out = {}
    stack = [(int(position), False)]
    while stack:
        pos, done = stack.pop()
        if not 0 <= pos < n_nodes:
            raise ValueError("child index %d outside the array of %d nodes" % (pos, n_nodes))
        w0 = int(words[2 * pos])
        if done:
            off = w0 & MASK
            out[pos] = (int(words[2 * pos + 1]), tuple(out.pop(off + i) for i in range(8)))
        elif w0 & FLAG:                                   # :153 "if the has child flag is set"
            off = w0 & MASK                               # :157 low 30 bits
            stack.append((pos, True))
            for i in range(8):                            # :159-161
                stack.append((off + i, False))
        else:
            out[pos] = (int(words[2 * pos + 1]), None)    # :165 data_ = octree[2*position + 1]
    return out[int(position)]


def pull_to_cpu(words):
    """OctreeNode::pullToCPU (octree.cpp:81-111): the 8 top nodes of a stackless array are the children of the node that
    owns it; returns the tuple of those 8 host sub-trees."""
    words = np.asarray(words, dtype=np.uint32)
    if words.size < 16:
        raise ValueError("insufficient size (octree.cpp:88-91)")
    return tuple(pull_from_linear_tree(words, i) for i in range(8))


# ------------------------------------------------------------------------------------------------ the reference's writer
def add_to_linear_tree(children):
    """OctreeNode::pushToGPU + addToLinearTree (octree.cpp:41-79, 118-149) for a node whose 8 children are `children`
    (host trees as returned by pull_to_cpu): the reference's depth-first layout -- child i of the owner at position i, a
    node's own children at the `offset` handed down, `new_offset + 8` threaded through the recursion -- with the flag
    bit SET for nodes that have children (the reference leaves it to chance, see the module docstring)."""
    words = []

    def ensure(n_nodes):
        while len(words) < 2 * n_nodes:
            words.append(0)

    def add(node, position, offset):                      # :118
        value, kids = node
        new_offset = offset
        ensure(max(position + 1, offset))
        words[2 * position + 1] = value                   # :123
        if kids is not None:                              # :125 has_children_
            words[2 * position] = FLAG | (offset & MASK)  # :122 with the flag the reader needs
            ensure(offset + 8)
            for i in range(8):                            # :127-129
                new_offset = add(kids[i], offset + i, new_offset + 8)
            new_offset += 8                               # :130
        else:
            words[2 * position] = 0                       # :135-137
        return new_offset

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    offset = 8                                            # :58
    for i in range(8):                                    # :59-61
        offset = add(children[i], i, offset)
    # the reference's threading of `offset` leaves gaps (it adds 8 both on the way down and after the loop): unreferenced
    # slots stay zero.  Size = highest slot written.
    return np.array(words, dtype=np.uint32)


def count_nodes(tree):
    """nodes of a host tree (the node itself included)"""
    total, stack = 0, [tree]
    while stack:
        value, kids = stack.pop()
        total += 1
        if kids is not None:
            stack.extend(kids)
    return total


def subtree_at(children, path):
    """the host node reached from the owner's children by the octant path"""
    node = (None, children)
    for p in path:
        if node[1] is None:
            raise ValueError("the path leaves the tree")
        node = node[1][p]
    return node


# ------------------------------------------------------------------------------------------------ checkpoint container
POOL_HEADER = struct.Struct("<8sIi3ffiIQ16s")     # magic, version, num_nodes, center[3], edge, max_depth, reserved, checksum, pad
assert POOL_HEADER.size == 64


def fnv1a_words(words):
    """FNV-1a over 32-bit words (svoslam_pool_save's checksum: one xor + multiply per WORD)"""
    h = 1469598103934665603
    for w in np.asarray(words, dtype=np.uint32).tolist():
        h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv1a_bytes(data, h=1469598103934665603):
    """FNV-1a over bytes (the sub-tree file's checksum)"""
    for b in bytes(data):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def write_pool_file(path, words, center, edge, depth):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    assert words.size % 16 == 0
    c = [float(np.float32(x)) for x in center]
    hdr = POOL_HEADER.pack(b"SVOPOOL1", 1, words.size // 2, c[0], c[1], c[2], float(np.float32(edge)), int(depth), 0, fnv1a_words(words), b"\0" * 16)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(words.tobytes())


def read_pool_file(path):
    """-> (words, center, edge, depth); raises ValueError on anything svoslam_pool_load must refuse"""
    raw = open(path, "rb").read()
    if len(raw) < 64:
        raise ValueError("short file")
    magic, version, num_nodes, cx, cy, cz, edge, depth, _res, checksum, _pad = POOL_HEADER.unpack(raw[:64])
    if magic != b"SVOPOOL1" or version != 1 or num_nodes < 8 or num_nodes % 8:
        raise ValueError("bad header")
    body = raw[64:]
    if len(body) < 8 * num_nodes:
        raise ValueError("truncated")
    words = np.frombuffer(body[:8 * num_nodes], dtype=np.uint32).copy()
    if fnv1a_words(words) != checksum:
        raise ValueError("checksum")
    return words, (cx, cy, cz), edge, depth


# ------------------------------------------------------------------------------------------------ sub-tree paging
SUBTREE_HEADER = struct.Struct("<8sIi16sIIiIQ8s")  # magic, version, levels, path[16], node_index, num_tiles, pool_size, reserved, checksum, pad
assert SUBTREE_HEADER.size == 64


def walk_path(words, path):
    """index and word0 of the node reached by the octant path (children of the root = nodes 0..7)"""
    base = node = w0 = 0
    for k, p in enumerate(path):
        node = base + int(p)
        w0 = int(words[2 * node])
        if k + 1 < len(path):
            if not w0 & FLAG:
                raise ValueError("the path ends above the requested level")
            base = w0 & MASK
    return node, w0


def evict_subtree(words, path):
    """What svoslam_pool_evict_subtree must produce, restated on host words: breadth-first over the TILES below the node
    (level by level; within a level in parent order, then octant order), the stand-alone linear tree with RELATIVE child
    indices (tile k of the list holds nodes 8k .. 8k+7), the original tile indices; and the pool afterwards (the node
    childless: word0 = 0, colour kept; the tiles zeroed).  -> (tiles, blob_words, words_after, node_index)"""
    words = np.array(words, dtype=np.uint32)
    node, w0 = walk_path(words, path)
    if not w0 & FLAG:
        raise ValueError("nothing below this node")
    tiles = [w0 & MASK]
    level = [w0 & MASK]
    blob = []
    while level:
        nxt = []
        for t in level:
            for j in range(8):
                a, b = int(words[2 * (t + j)]), int(words[2 * (t + j) + 1])
                if a & FLAG:
                    rel = len(tiles) + len(nxt)
                    nxt.append(a & MASK)
                    a = FLAG | ((rel * 8) & MASK)
                blob.extend((a, b))
        tiles.extend(nxt)
        level = nxt
    after = words.copy()
    after[2 * node] = 0
    for t in tiles:
        after[2 * t:2 * t + 16] = 0
    return np.array(tiles, dtype=np.uint32), np.array(blob, dtype=np.uint32), after, node


def read_subtree_file(path):
    raw = open(path, "rb").read()
    magic, version, levels, opath, node_index, num_tiles, pool_size, _res, checksum, _pad = SUBTREE_HEADER.unpack(raw[:64])
    if magic != b"SVOSUBT1" or version != 1 or not 1 <= levels <= 16 or num_tiles == 0:
        raise ValueError("bad header")
    need = 64 + 4 * num_tiles + 64 * num_tiles
    if len(raw) < need:
        raise ValueError("truncated")
    tiles = np.frombuffer(raw[64:64 + 4 * num_tiles], dtype=np.uint32).copy()
    nodes = np.frombuffer(raw[64 + 4 * num_tiles:need], dtype=np.uint32).copy()
    if fnv1a_bytes(nodes.tobytes(), fnv1a_bytes(tiles.tobytes())) != checksum:
        raise ValueError("checksum")
    return {"levels": levels, "path": list(opath[:levels]), "node_index": node_index, "pool_size": pool_size, "tiles": tiles, "nodes": nodes}


def write_subtree_file(path, octant_path, node_index, pool_size, tiles, nodes):
    tiles = np.ascontiguousarray(tiles, dtype=np.uint32)
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    p = bytes(octant_path) + b"\0" * (16 - len(octant_path))
    hdr = SUBTREE_HEADER.pack(b"SVOSUBT1", 1, len(octant_path), p, int(node_index), tiles.size, int(pool_size), 0,
                              fnv1a_bytes(nodes.tobytes(), fnv1a_bytes(tiles.tobytes())), b"\0" * 8)
    with open(path, "wb") as f:
        f.write(hdr); f.write(tiles.tobytes()); f.write(nodes.tobytes())


def restore_subtree(words_after, sub):
    """the pool with the sub-tree of read_subtree_file back at the indices it came from (absolute child indices)"""
    out = np.array(words_after, dtype=np.uint32)
    node, w0 = walk_path(out, sub["path"])
    if node != sub["node_index"]:
        raise ValueError("not the pool the sub-tree came from")
    if w0 & FLAG:
        raise ValueError("the cube was fused into while it was paged out")
    tiles, nodes = sub["tiles"], sub["nodes"]
    for k, t in enumerate(tiles.tolist()):
        for j in range(8):
            a, b = int(nodes[2 * (8 * k + j)]), int(nodes[2 * (8 * k + j) + 1])
            if a & FLAG:
                a = FLAG | (int(tiles[(a & MASK) >> 3]) & MASK)
            out[2 * (t + j)] = a
            out[2 * (t + j) + 1] = b
    out[2 * node] = FLAG | (int(tiles[0]) & MASK)
    return out


# ------------------------------------------------------------------------------------------------ re-rooting
def average_children(kids):
    """averageChildren (svo.cu:384-441, Q5: all 8 always counted): per channel floor(sum / 8), alpha = max"""
    r = sum(k[0] & 0xFF for k in kids) >> 3
    g = sum((k[0] >> 8) & 0xFF for k in kids) >> 3
    b = sum((k[0] >> 16) & 0xFF for k in kids) >> 3
    a = max(k[0] >> 24 for k in kids)
    return r | (g << 8) | (b << 16) | (a << 24)


def expand_root(children, center, edge, toward):
    """A correct Octree::expandBySize for ONE doubling, in terms of the host tree: the new root cube has
    twice the edge and is centred one old half-edge towards `toward` on every axis; the old root becomes the child on the
    side AWAY from the growth (octant bit k = old centre > new centre on axis k), carrying the mip value of its children;
    the seven other children are empty nodes (value 0, as initOctree leaves them, svo.cu:24-31).
    -> (new children, new center, new edge)"""
    c = [np.float32(x) for x in center]
    e = np.float32(edge)
    octant, nc = 0, []
    for k in range(3):
        plus = np.float32(toward[k]) > c[k]
        n = np.float32(c[k] + e) if plus else np.float32(c[k] - e)
        nc.append(n)
        if c[k] > n:
            octant |= 1 << k
    old_root = (average_children(children), tuple(children))
    new = tuple(old_root if i == octant else (0, None) for i in range(8))
    return new, tuple(float(x) for x in nc), float(np.float32(e * np.float32(2.0)))
