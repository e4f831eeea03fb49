"""Pins the CPU oracle against the hand-derived known-answer vectors of
SURVEY.md Appendix C (tests/golden/kat_appendix_c.json).  The reference ships
no tests of its own, so these are the only pins."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat_appendix_c.json")))


def rgba(word1):
    return [int(word1 & 0xFF), int((word1 >> 8) & 0xFF), int((word1 >> 16) & 0xFF), int(word1 >> 24)]


def test_c1_keys(oracle):
    c1 = KAT["C1_keys"]
    for p, k in zip(c1["points"], c1["keys"]):
        assert oracle.compute_key(p, c1["center"], c1["depth"], c1["half_edge"]) == k
    for k, d in c1["depth_from_key"].items():
        assert oracle.depth_from_key(int(k)) == d
    v, out = oracle.get_first_value_and_shift_down(c1["shift_down"]["in"])
    assert (v, out) == (c1["shift_down"]["value"], c1["shift_down"]["out"])


def test_c1_invalid_point_key_is_1(oracle):
    # svo.cu:38 (Q1): x or z non-finite -> key 1; a non-finite y is NOT detected
    assert oracle.compute_key([np.inf, 0, 0], [0, 0, 0], 3, 1.0) == 1
    assert oracle.compute_key([0, 0, np.nan], [0, 0, 0], 3, 1.0) == 1
    assert oracle.compute_key([0.5, np.inf, 0.5], [0, 0, 0], 1, 1.0) == 0o17


def test_c2_insert(oracle):
    c1, c2 = KAT["C1_keys"], KAT["C2_insert"]
    pool = oracle.Pool()
    # split planning on the fresh pool
    pool.insert_cloud(np.zeros((0, 3)), np.zeros((0, 3)), 2, c1["center"], c1["half_edge"])  # creates 8 root children
    assert pool.size == 8
    total, sizes, codes = pool.prepass(np.array(c1["keys"], np.int64), 2)
    assert total == 2 and sizes == [2, 0] and codes.tolist() == c2["pass0_split_list"]

    pool.insert_cloud(c1["points"], c2["colors"], c1["depth"], c1["center"], c1["half_edge"])
    assert pool.size == c2["size_after_first"]
    w = pool.words()
    assert int(w[0]) == c2["node0_word0_after_first"]
    assert int(w[14]) == c2["node7_word0_after_first"]
    for node, val in c2["nodes_after_first"].items():
        assert rgba(int(w[2 * int(node) + 1])) == val, node
    # node 0: root-level pass (Q6) with snapshot semantics: mean of root children 0..7
    # children: node0=(0,1,1,129) [mean of 8..15], node7=(28,22,19,129), others 0
    assert rgba(int(w[1])) == [28 // 8, 23 // 8, 20 // 8, 129]

    pool.insert_cloud(c1["points"], c2["colors"], c1["depth"], c1["center"], c1["half_edge"])
    assert pool.size == c2["size_after_second"]
    w = pool.words()
    assert int(w[2 * 23]) == c2["node23_word0_after_second"]  # Q4: octant-7 leaf gains children once
    assert not (int(w[2 * 8]) & 0x40000000) and not (int(w[2 * 16]) & 0x40000000)
    for node, val in c2["nodes_after_second"].items():
        assert rgba(int(w[2 * int(node) + 1])) == val, node
    # a third insert must not split again
    pool.insert_cloud(c1["points"], c2["colors"], c1["depth"], c1["center"], c1["half_edge"])
    assert pool.size == c2["size_after_second"]


def test_c3_vertex_map(oracle):
    c3 = KAT["C3_vertex"]
    d = np.zeros((c3["h"], c3["w"]), np.uint16)
    d[240, 320] = 1000
    d[0, 0] = 1000
    v = oracle.vertex_map(d, c3["f"], c3["f"], c3["w"], c3["h"])
    assert v[240, 320].tolist() == [0.0, 0.0, 1.0]
    f = np.float32(c3["f"])
    exp = np.array([np.float32(-320) * np.float32(1000) / f * np.float32(0.001),
                    np.float32(240) * np.float32(1000) / f * np.float32(0.001), np.float32(1000) * np.float32(0.001)],
                   np.float32)
    assert v[0, 0].tolist() == exp.tolist()
    np.testing.assert_allclose(v[0, 0], [-320 / 570.3, 240 / 570.3, 1.0], rtol=1e-6)
    assert np.isinf(v[1, 1]).all()  # depth 0 -> (INF,INF,INF)


def test_c4_bilateral_constant(oracle):
    img = np.full((24, 32), 1234, np.uint16)
    out = oracle.bilateral(img)
    assert (out == 1234).all()  # windows never empty for W,H >= 2


def test_c5_icp_identical_frames(oracle):
    rng = np.random.default_rng(7)
    h, w = 48, 64
    d = (1500 + 200 * np.sin(np.arange(w) / 9.0)[None, :] + 100 * np.cos(np.arange(h) / 7.0)[:, None]).astype(np.uint16)
    v = oracle.vertex_map(d, 57.03, 57.03, w, h)
    n = oracle.normal_map(v)
    A, b = oracle.icp_cost2(v, n, v, n)
    assert (b == 0).all()
    assert np.array_equal(A, A.T)
    assert np.linalg.eigvalsh(A.astype(np.float64)).min() > -1e-3 * np.abs(A).max()
    x = oracle.solve_cholesky(A, b)
    assert (x == 0).all()
    T = oracle.icp_update_transform(x)
    assert T.tolist() == oracle.mat4_identity().tolist()
    del rng


def test_c6_single_step_render(oracle):
    c1, c2, c6 = KAT["C1_keys"], KAT["C2_insert"], KAT["C6_render"]
    for case in c6["cases"]:
        pool = oracle.Pool()
        pool.insert_cloud(c1["points"], c2["colors"], c1["depth"], c1["center"], c1["half_edge"])
        w = pool.words()
        r, g, b, a = case["rgba"]
        w[2 * c6["node"] + 1] = r | (g << 8) | (b << 16) | (a << 24)
        view = oracle.look_at(c6["origin"], [1.0, 0.3, 0.2], [0, 1, 0])
        img, steps, _ = oracle.cone_trace(w, 16, 12, 45.0, view, c6["center"], c6["size"])
        assert steps == 16 * 12  # every ray retires on its first sample
        assert (img.reshape(-1, 4) == np.array(case["pixel"], np.uint8)).all()
    # A <= 253 does not retire on step 1
    pool = oracle.Pool()
    pool.insert_cloud(c1["points"], c2["colors"], c1["depth"], c1["center"], c1["half_edge"])
    w = pool.words()
    w[2 * 16 + 1] = 200 | (100 << 8) | (50 << 16) | (253 << 24)
    view = oracle.look_at(c6["origin"], [1.0, 0.3, 0.2], [0, 1, 0])
    img, steps, _ = oracle.cone_trace(w, 4, 4, 45.0, view, c6["center"], c6["size"])
    assert steps > 16
