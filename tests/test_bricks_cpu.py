"""CPU check of the occupancy bricks' RULES (csrc/pool_grid.hpp "occupancy bricks", decode() of cone_trace_brick_kernel):
brick entries and level-grid entries are built from an oracle pool exactly as brick_rebuild / grid_entry define them, and
the kernel's decode rule -- restated here in Python -- must give, for every sample it decides, the level and the
saturation of the node the reference's walk ends on (cone_tracing_kernels.cu:76-119, restated below as walk()).  A sample
it does not decide goes to the tree walk in the kernel; the test also checks that the samples the design promises to
decide (LOD 8..12 among nodes, empty space at or above level 8) are decided; likewise for the shape of deeper pools (shift 1:
bricks of level-10 nodes, level-12 cells, bits for level 13, inside the field's window).  No device code runs here."""
import numpy as np
import pytest

from util import surface_cloud

FLAG, MASK = 0x40000000, 0x3FFFFFFF


def walk(words, bits, lod):
    """the reference's descent for a sample whose octant bits per level are bits[l] (l = 1..): (level it ends on, colour word)"""
    node, child = 0, 0
    depth = lod
    for i in range(lod):
        node = child + bits[i]
        if not (int(words[2 * node]) & FLAG):
            depth = i + 1
            break
        child = int(words[2 * node]) & MASK
    return depth, int(words[2 * node + 1]) if lod >= 1 else int(words[1])


def grid_entry(words, bits8):
    """pool_grid.hip grid_entry(): x = flag | children tile of the level-8 node, or the level of the first childless node; y = colour"""
    base, out = 0, (0, 0)
    for l in range(1, 9):
        nd0, nd1 = int(words[2 * (base + bits8[l - 1])]), int(words[2 * (base + bits8[l - 1]) + 1])
        if not (nd0 & FLAG):
            return (l, nd1)
        base = nd0 & MASK
        out = (FLAG | base, nd1)
    return out


def brick_entry(words, g, bits_below_8, shift):
    """brick_rebuild<.., S>(): the 16-bit entry of the cell at level 11 + shift below the level-8 node whose grid entry is g
    (0 = no brick): the levels 9 .. 9 + shift are the same for the whole brick (a childless node there: code 5 for level 9 of
    shift 1, 1 for the brick node itself), then two levels per cell, then the tile of the bits level"""
    if not (g[0] & FLAG):
        return 0
    sat = lambda w: 1 if (w >> 24) >= 254 else 0
    nl = 9 + shift
    tile, v = g[0] & MASK, 0
    for l in range(9, nl + 3):                      # levels 9 .. cell level
        n = tile + bits_below_8[l - 9]
        w0, w1 = int(words[2 * n]), int(words[2 * n + 1])
        v |= sat(w1) << (l - 5)
        if not (w0 & FLAG):
            return v | ((4 + (l - 8)) if l < nl else (l - nl + 1))
        tile = w0 & MASK
    v |= 4
    for q in range(8):
        c0, c1 = int(words[2 * (tile + q)]), int(words[2 * (tile + q) + 1])
        v |= sat(c1) << (8 + q)
        if c0 & FLAG:
            v |= 8
    return v


def decode(e, g, lod, oct_bl, shift):
    """decode() of cone_trace_brick_kernel<.., S>: (decided, level, retired)"""
    bl = 12 + shift
    st = ((e & 7) | 8) if shift == 0 else (0x009DCBA8 >> ((e & 7) << 2)) & 15
    depth_b = min(lod, st)
    bit = depth_b - 5
    nl = 9 + shift
    # the brick node's level and its cells' two levels; a level above the brick node only where the path stops there
    by_brick = 0 <= depth_b - nl < 3 or (8 < st < nl and depth_b == st)
    if lod >= bl:
        deep = depth_b == bl and (lod == bl or not (e & 8))
        by_brick = by_brick or deep
        bit = 8 + oct_bl if deep else bit
    depth_g = min(g[0], 8)
    top_g = 127 if g[0] < FLAG else 8
    by_grid = depth_g <= lod <= top_g
    depth = depth_b if by_brick else depth_g
    retired = ((e >> bit) & 1) if by_brick else (1 if g[1] >= 0xFE000000 else 0)
    return by_brick or by_grid, depth, retired


@pytest.mark.parametrize("depth,passes,shift", [(10, 3, 0), (12, 130, 0), (13, 2, 0), (13, 130, 1), (14, 3, 1), (11, 2, 1)])
def test_brick_and_grid_entries_decide_what_the_reference_walk_finds(oracle, depth, passes, shift):
    rng = np.random.default_rng(depth)
    pool = oracle.Pool()
    pts, col = surface_cloud(rng, 1500, jitter=0.002)
    for _ in range(passes):   # 130 passes: leaves saturate (A += 2 per observation), Q4 gives octant-7 leaves children
        pool.insert_cloud(pts, col, depth, (0, 0, 0), 1.0)
    words = pool.words()
    # samples: on and around the inserted points, and uniformly in the cube; octant bits straight from coordinates (the
    # rule under test does not depend on how the kernel obtains the bits)
    near = pts[rng.integers(0, len(pts), 700)] + rng.normal(scale=0.004, size=(700, 3)).astype(np.float32)
    samples = np.concatenate([near, rng.uniform(-0.99, 0.99, size=(300, 3))]).clip(-0.999, 0.999)
    decided_among_nodes = total_among_nodes = decided_empty = total_empty = outside = 0
    cl = 11 + shift
    org, span = ((1 << cl) - 2048) // 2, 2048           # the window of the brick field, in cells of level cl (pool_grid.hpp)
    for p in samples:
        cell = np.floor((p + 1.0) * 0.5 * (1 << 16)).astype(np.int64)      # 16 levels of octant bits per axis
        bits = [int(((cell[0] >> (15 - l)) & 1) | (((cell[1] >> (15 - l)) & 1) << 1) | (((cell[2] >> (15 - l)) & 1) << 2)) for l in range(16)]
        g = grid_entry(words, bits[:8])
        inside = all(org <= int(c >> (16 - cl)) < org + span for c in cell)
        e = brick_entry(words, g, bits[8:], shift) if inside else 0
        outside += not inside
        for lod in range(1, 16):
            want_depth, want_word = walk(words, bits, lod)
            ok, got_depth, got_ret = decode(e, g, lod, bits[11 + shift], shift)
            if ok:
                assert got_depth == want_depth, (p, lod, hex(e), g, want_depth)
                assert got_ret == (1 if (want_word >> 24) >= 254 else 0), (p, lod, hex(e), g, hex(want_word))
            among = bool(g[0] & FLAG)
            if among and inside and 8 <= lod <= 12 + shift and not (shift == 1 and lod == 9):
                total_among_nodes += 1
                decided_among_nodes += ok
            if not among and lod >= 8:
                total_empty += 1
                decided_empty += ok
    # LOD 8..12 + shift among nodes inside the window: always the bricks' (or the grid's, at 8) -- except LOD 9 of shift 1 (a level-9
    # node's alpha is not kept current in the eight bricks below it)
    assert total_among_nodes > (500 if shift == 0 else 250) and decided_among_nodes == total_among_nodes
    assert total_empty > 100 and decided_empty == total_empty                       # empty space at or above level 8: always the grid's
    assert (outside > 50) == (shift == 1)                                           # shift 1: some samples lie outside the window and take the tree walk
