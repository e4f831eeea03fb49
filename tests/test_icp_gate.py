"""The ICP distance gate without its square root (csrc/icp_device.hpp, kDistThreshSq): the reference rejects a pixel pair when
length(v2 - v1) > DIST_THRESH = 0.1f (localization_kernels.cu:17,196-199); the kernels compare the SQUARED length with the largest
binary32 whose correctly rounded root is still <= 0.1f.  Pinned here on the CPU (numpy's float32 sqrt is IEEE-correctly rounded, as is
the oracle's sqrtf): the constant in the header is that float, and the two predicates agree on a dense window around it, on random
values over the whole range, and on the specials."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T01 = np.float32(0.1)


def header_threshold():
    src = open(os.path.join(ROOT, "octree-slam_amd", "csrc", "icp_device.hpp")).read()
    m = re.search(r"kDistThreshSq\s*=\s*(0x[0-9a-fA-F.]+p[-+]?\d+)f;", src)
    assert m, "kDistThreshSq not found"
    v = float.fromhex(m.group(1))
    f = np.float32(v)
    assert float(f) == v                      # the literal is exactly a binary32
    return f


def test_threshold_is_the_last_float_whose_root_rounds_to_at_most_a_tenth():
    t = header_threshold()
    assert np.array([t]).view(np.uint32)[0] == 0x3C23D70B
    nxt = np.nextafter(t, np.float32(1))
    assert np.sqrt(t) <= T01 and np.sqrt(nxt) > T01
    # the same in exact arithmetic: the root of t lies below the midpoint of 0.1f and its successor, the root of nxt above it
    mid = (float(T01) + float(np.nextafter(T01, np.float32(1)))) / 2
    assert float(t) < mid * mid < float(nxt)


def test_predicates_agree():
    t = header_threshold()
    bits = np.array([t]).view(np.uint32)[0]
    window = (bits + np.arange(-(1 << 20), 1 << 20, dtype=np.int64)).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(17)
    rand = rng.integers(0, 0x7F800000, size=1 << 22, dtype=np.uint32).view(np.float32)      # every non-negative finite float, subnormals included
    special = np.array([0.0, np.inf, np.nan, 1e-45, 0.01, 0.010000001, 0.0100000017, 3.4e38], np.float32)
    for x in (window, rand, special):
        with np.errstate(invalid="ignore"):
            assert np.array_equal(np.sqrt(x) > T01, x > t)
