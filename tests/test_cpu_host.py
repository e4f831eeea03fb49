"""CPU-only tests: the C-ABI library loads and exports every symbol include/svoslam.h declares,
fails loudly without a GPU, and the oracle agrees with independent numpy float64 restatements."""
import ctypes as C
import os

import pytest
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    import svoslam_pkg
    return svoslam_pkg.load()


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    header = open(pkg.HEADER_PATH).read()
    declared = sorted(set(re.findall(r"\b(svoslam_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 45
    L = C.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "symbol %s declared in include/svoslam.h but not exported" % name
        assert name in pkg.SIGNATURES, "symbol %s has no ctypes signature" % name
    assert sorted(pkg.SIGNATURES) == declared
    assert pkg.lib().svoslam_abi_version() == 1


def test_no_gpu_means_loud_failure():
    import torch
    pkg = load_pkg()
    if torch.cuda.is_available():
        return
    L = pkg.lib()
    h = C.c_void_p()
    assert L.svoslam_workspace_create(C.byref(h)) == -2  # SVOSLAM_ERR_NO_DEVICE: no CPU fallback
    assert b"no CPU fallback" in L.svoslam_last_error()
    assert pkg.device_arch() is None
    try:
        pkg.Workspace()
    except pkg.SvoslamError:
        pass
    else:
        raise AssertionError("Workspace() must raise without a GPU")


def test_product_never_imports_oracle():
    pat = re.compile(r"oracle", re.I)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "octree-slam_amd")):
        if "_obj" in dirpath or "__pycache__" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                for ln, line in enumerate(open(os.path.join(dirpath, f), errors="ignore"), 1):
                    if pat.search(line) and ("import" in line or "#include" in line or "dlopen" in line or "CDLL" in line):
                        raise AssertionError("%s:%d references the oracle: %s" % (f, ln, line.strip()))


# ---- oracle vs independent float64 numpy restatements -----------------------
def test_oracle_bilateral_vs_float64(oracle):
    rng = np.random.default_rng(0)
    h, w = 40, 56
    d = (1500 + 300 * rng.random((h, w))).astype(np.uint16)
    d[5:9, 7:30] += 800
    out = oracle.bilateral(d).astype(np.int64)
    ref = np.zeros((h, w))
    dd = d.astype(np.float64)
    for y in range(h):
        for x in range(w):
            ys, xs = slice(max(y - 3, 0), min(y + 4, h - 1)), slice(max(x - 3, 0), min(x + 4, w - 1))
            yy, xx = np.mgrid[ys, xs]
            win = dd[ys, xs]
            wgt = np.exp(-(((x - xx) ** 2 + (y - yy) ** 2) * 0.5 / 4.5 ** 2 + (dd[y, x] - win) ** 2 * 0.5 / 40.0 ** 2))
            ref[y, x] = (win * wgt).sum() / wgt.sum() if wgt.sum() > 0 else 0
    assert np.abs(out - np.rint(ref)).max() <= 1          # float32 + <=1 ulp exp vs float64: at most 1 LSB
    assert (out != np.rint(ref)).mean() < 0.01


def test_oracle_icp_vs_float64(oracle):
    rng = np.random.default_rng(1)
    h, w = 60, 80
    yy, xx = np.mgrid[0:h, 0:w]
    d = (1500 + 200 * np.sin(xx / 9.0) + 100 * np.cos(yy / 7.0) + rng.normal(scale=1.0, size=(h, w))).astype(np.uint16)
    f = 570.3 * w / 640
    v1 = oracle.vertex_map(d, f, f, w, h); n1 = oracle.normal_map(v1)
    T = oracle.icp_update_transform(np.array([0.003, -0.002, 0.001, 0.003, -0.002, 0.001], np.float32))
    v2 = oracle.transform_vertex_map(v1, T); n2 = oracle.transform_normal_map(n1, T)
    A, b = oracle.icp_cost2(v1, n1, v2, n2)
    V1, N1, V2, N2 = (a.reshape(-1, 3).astype(np.float64) for a in (v1, n1, v2, n2))
    ok = np.isfinite(V1).all(1) & np.isfinite(V2).all(1) & np.isfinite(N1).all(1) & np.isfinite(N2).all(1)
    ok &= (V1[:, 2] >= 0.1) & (V2[:, 2] >= 0.1) & (V1[:, 2] <= 10) & (V2[:, 2] <= 10)
    ok &= np.linalg.norm(np.where(ok[:, None], V2 - V1, 0), axis=1) <= 0.1
    ok &= (np.where(ok[:, None], N2 * N1, 0)).sum(1) >= 0.87
    limit = (h * w // (20 * w // 640)) * (20 * w // 640)
    ok[limit:] = False
    x, y, z = V2[ok].T; nx, ny, nz = N1[ok].T
    J = np.stack([-x * ny - y * nz, -z * nx + x * nz, y * nx + z * ny, nx, ny, nz], 1)   # Q14 rows
    r = (N1[ok] * (V1[ok] - V2[ok])).sum(1)
    rA, rb = J.T @ J, J.T @ r
    assert ok.sum() > 1000
    np.testing.assert_allclose(A, rA, rtol=1e-4, atol=1e-4 * np.abs(rA).max())   # north_star tolerance 1e-4 rel
    np.testing.assert_allclose(b, rb, rtol=1e-4, atol=1e-4 * np.abs(rb).max())
    xs = oracle.solve_cholesky(A, b)
    np.testing.assert_allclose(xs, np.linalg.solve(rA, rb), rtol=2e-2, atol=2e-4)


def test_oracle_mat4_vs_numpy(oracle):
    rng = np.random.default_rng(2)
    a = rng.normal(size=(4, 4)).astype(np.float32); b = rng.normal(size=(4, 4)).astype(np.float32)
    # column-major flat arrays: flat = M.T.reshape(-1)
    fa, fb = a.T.reshape(-1).copy(), b.T.reshape(-1).copy()
    np.testing.assert_allclose(oracle.mat4_mul(fa, fb).reshape(4, 4).T, a @ b, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(oracle.mat4_inverse(fa).reshape(4, 4).T, np.linalg.inv(a.astype(np.float64)), rtol=1e-3, atol=1e-4)
    for ang in (0.0, 1e-3, 0.3, -2.0, 10.0, 1234.5):
        s, c = oracle.sincos(ang)
        assert abs(s - np.float32(np.sin(np.float64(np.float32(ang))))) <= 6e-8
        assert abs(c - np.float32(np.cos(np.float64(np.float32(ang))))) <= 6e-8
    R = oracle.mat4_rotate_deg(oracle.mat4_identity(), 30.0, (0, 0, 1)).reshape(4, 4).T
    np.testing.assert_allclose(R[:2, :2], [[np.cos(np.pi / 6), -np.sin(np.pi / 6)], [np.sin(np.pi / 6), np.cos(np.pi / 6)]], atol=1e-6)
    V = oracle.look_at((0, 0.1, -0.6), (0, 0.1, 0), (0, 1, 0)).reshape(4, 4).T
    np.testing.assert_allclose(V @ np.array([0, 0.1, -0.6, 1.0]), [0, 0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(V @ np.array([0, 0.1, 0.0, 1.0]), [0, 0, -0.6, 1], atol=1e-6)   # looks down -Z


def test_oracle_tree_invariants_and_extract_roundtrip(oracle):
    """extraction is the inverse of insertion for alpha > 127"""
    rng = np.random.default_rng(3)
    depth, center, edge = 6, (0.0, 0.0, 0.0), 1.0
    pts = (rng.random((4000, 3)) * 1.9 - 0.95).astype(np.float32)
    col = rng.integers(0, 256, (4000, 3), dtype=np.uint8)
    pool = oracle.Pool()
    pool.insert_cloud(pts, col, depth, center, edge)
    keys = np.unique(oracle.compute_keys(pts, depth, center, edge))
    ce, co = pool.extract(depth, center, edge)
    ek = oracle.compute_keys(ce[:, :3], depth, center, edge)
    assert np.array_equal(np.sort(ek), keys) and np.array_equal(ek, np.sort(ek))   # BFS order == key order
    assert (co[:, 3] * 255 > 127).all()
    w = pool.words()
    w0 = w[0::2]; fl = (w0 & 0x40000000) != 0
    ch = w0[fl] & 0x3FFFFFFF
    assert (ch % 8 == 0).all() and np.unique(ch).size == ch.size and ch.max() + 8 == pool.size


def test_synth_stream_is_deterministic_and_plausible():
    import importlib
    load_pkg()
    synth = importlib.import_module("octree_slam_amd.synth")
    d1, c1 = synth.render_frame(3, 160, 120)
    d2, c2 = synth.render_frame(3, 160, 120)
    assert (d1 == d2).all() and (c1 == c2).all()
    d = d1.numpy().view(np.uint16)
    assert 0.005 < (d == 0).mean() < 0.02                 # 1 % dropouts
    valid = d[d > 0]
    assert 400 < valid.min() and valid.max() < 6000       # inside the 6 x 3 x 6 m room
    assert c1.shape == (120, 160, 3)


def test_compat_header_compiles(tmp_path):
    """the reference-signature C++ shim compiles and links against the library with plain g++"""
    import subprocess
    pkg = load_pkg()
    src = tmp_path / "host.cpp"
    src.write_text('''
#include "octree_slam_compat.hpp"
using namespace octree_slam;
void fuse_and_render(const vec3* d_points, const Color256* d_colors, int n, uchar4_t* d_image) {
  static unsigned int* pool = nullptr; static int pool_size = 0;
  svo::svoFromPointCloud(d_points, d_colors, n, 12, pool, pool_size, vec3{0, 1.5f, 0}, 4.096f);
  mat4 view = {};
  rendering::coneTraceSVO(d_image, vec2{640, 480}, 45.0f, view, SVO{pool, vec3{0, 1.5f, 0}, 4.096f});
  VoxelGrid grid; svo::extractVoxelGridFromSVO(pool, pool_size, 12, vec3{0, 1.5f, 0}, 4.096f, grid);
  sensor::ICPFrame a(8, 8), b(8, 8); float A[36], bb[6]; sensor::computeICPCost2(&a, b, A, bb); sensor::computeICPCost(&a, b, A, bb);
  startTiming(); (void)stopTiming();
  Mesh m; VoxelGrid vg; voxelization::meshToVoxelGrid(m, nullptr, vg); (void)voxelization::log_N();
  Mesh cube, cubes; voxelization::voxelGridToMesh(vg, cube, cubes);
  BoundingBox outer, inner; outer.bbox1 = vec3{2, 2, 2}; inner.bbox0 = vec3{-1, 0.5f, 0.5f}; inner.bbox1 = vec3{1, 1, 3};
  if (outer.contains(inner) || outer.distanceOutside(inner) != -1.0f + 0.0f && outer.distanceOutside(inner) != 0.0f) {}
  RawFrame frame(8, 8);
  sensor::subsampleDepth<uint16_t>(frame.depth, 8, 8); sensor::subsample<Color256>(frame.color, 8, 8);
  float* fl = nullptr; sensor::subsample<float>(fl, 0, 0); sensor::subsampleDepth<float>(fl, 0, 0);
  sensor::RGBDFrame ra(8, 8), rb(8, 8); sensor::computeRGBDCost(&ra, rb, A, bb);
  sensor::RGBDCamera cam(8, 8, vec2{570.3f, 570.3f}); cam.update(&frame); (void)cam.position(); (void)cam.orientation();
  world::Scene scene; scene.loadObjFile("x.obj"); scene.voxelizeMeshes(true); (void)scene.svo(); (void)scene.voxel_grid();
  scene.addPointCloudToOctree(vec3{0, 0, 0}, d_points, d_colors, n, outer); scene.extractVoxelGridFromOctree();
}
int main() {
  try { fuse_and_render(nullptr, nullptr, 0, nullptr); } catch (const std::exception& e) { return 0; }
  return 0;
}
''')
    exe = tmp_path / "host"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lsvoslam_hip",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_bench_line_contract():
    """the committed bench line (profiles/r01_bench_cfg3.json, written by bench.py on the GPU box) carries every
    field of the driver's contract plus the roofline and cpu_baseline objects, with consistent values"""
    import json
    line = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_cfg3.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 1.0) < 1e-6        # frames/s x s/frame
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
    assert abs(r["alg_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 - r["achieved"]) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


@pytest.mark.parametrize("name", ["r04_bench_cfg3.json", "r04_bench_cfg3_driver_args_20frames.json", "r04_bench_cfg4.json"])
def test_bench_line_contract_round4(name):
    """the round-4 lines: `value` is the median of the listed windows, the measured-copy peak stands beside the spec peak, the CPU
    baseline ran on the timed frames, the pose error and the second, labelled measurement with the corrected tracker are there"""
    import json
    line = json.load(open(os.path.join(ROOT, "profiles", name)))
    runs = sorted(line["runs"])
    assert len(runs) >= 5 and abs(line["value"] - runs[(len(runs) - 1) // 2 if len(runs) % 2 else len(runs) // 2]) < 1e-6 * line["value"] or line["value"] in line["runs"]
    assert line["value_min"] <= line["value"] <= line["value_max"] and len(line["runs_steady_state"]) == len(line["runs"])
    assert line["pipeline_fill"]["pipeline_fill_ms"] > 0 and line["pipeline_fill"]["steady_frames_per_s"] >= line["value"] * 0.98
    for r in [line["roofline"]] + line["roofline_stages"]:
        assert r["peak"] == 8000.0 and r["peak_measured"] == 6290.0 and abs(r["achieved"] / 6290.0 - r["frac_of_measured"]) < 1e-9
    first = line["config"]["frames_in_map_at_end"] - line["steps"]
    assert "frames %d.." % (first + 1) in line["cpu_baseline"]["sample"]
    assert line["config"]["pose_error_deg_end"] > 0 and "saturated_nodes_end" in line["config"] and "Q14" in line["config"]["tracker"]
    ct = line["corrected_tracker"]
    assert ct["value"] > 0 and ct["pose_error_deg_end"] < 15.0 and "corrected" in ct["what"]
    assert ct["value"] > 0.7 * line["value"]   # (a second pipeline of the process runs at its own rate: the runner's streams are reused)
    lat = line["latency"]                        # first map kernel of a frame to the end of its march: several frame periods
    assert lat["frame_latency_ms"] > 1.5 * line["ms_per_step"] and lat["frame_period_ms_with_marks"] > 0
    assert line["metric"].startswith("SLAM frames/sec") and "CORRECTED" not in line["config"]["workload"]


def test_config_struct_defaults_set_get_and_environment():
    """svoslam_config (include/svoslam.h): defaults, set / get round trip, refused values, and the one environment variable the
    library reads (SVOSLAM_CONFIG, in a child process: it is parsed once)"""
    import json
    import subprocess
    import sys
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    c = pkg.get_config()
    assert c["march_bricks"] == 1 and c["track_stream"] == 1 and c["runner_replicas"] == 1 and c["graphs"] == 0
    before = pkg.configure(track_mode=1, runner_lead=3)
    try:
        now = pkg.get_config()
        assert now["track_mode"] == 1 and now["runner_lead"] == 3 and now["march_bricks"] == before["march_bricks"]
        with pytest.raises(pkg.SvoslamError):
            pkg.configure(runner_replicas=3)
        with pytest.raises(pkg.SvoslamError):
            pkg.configure(track_mode=7)
        with pytest.raises(KeyError):
            pkg.configure(no_such_field=1)
    finally:
        pkg.configure(**before)
    assert pkg.get_config() == before
    code = "import sys, json; sys.path.insert(0, %r); import svoslam_pkg; print('CFG' + json.dumps(svoslam_pkg.load().get_config()))" % ROOT
    env = dict(os.environ, **pkg.config_env(march_bricks=0, track_workers=17, runner_deferred=0))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("CFG")][0][3:])
    assert got["march_bricks"] == 0 and got["track_workers"] == 17 and got["runner_deferred"] == 0 and got["track_stream"] == 1
