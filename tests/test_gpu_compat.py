"""The drop-in claim end to end: a reference-style C++ caller (the code of main.cpp:39-44 / octree.cpp:290 /
cuda_renderer.cpp:163 / rgbd_camera.cpp:64-141, written against include/octree_slam_compat.hpp only, built with
plain g++ and no HIP headers) runs on the GPU and its outputs equal the CPU oracle's."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOST = r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "octree_slam_compat.hpp"
using namespace octree_slam;

template <class T> static std::vector<T> rd(FILE* f, size_t n) { std::vector<T> v(n); if (fread(v.data(), sizeof(T), n, f) != n) exit(3); return v; }
template <class T> static void wr(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }
template <class T> static T* up(const std::vector<T>& v) { void* d = nullptr; svoslam_malloc(&d, v.size() * sizeof(T) + 16); svoslam_memcpy_h2d(d, v.data(), v.size() * sizeof(T)); return (T*)d; }

int main(int argc, char** argv) {
  FILE* in = fopen(argv[1], "rb"); FILE* out = fopen(argv[2], "wb");
  int hdr[4]; if (fread(hdr, 4, 4, in) != 4) return 2;           // n points, depth, image w, h
  const int n = hdr[0], depth = hdr[1], w = hdr[2], h = hdr[3];
  auto pts = rd<float>(in, 3 * (size_t)n); auto col = rd<unsigned char>(in, 3 * (size_t)n);
  auto view = rd<float>(in, 16);
  auto dep0 = rd<uint16_t>(in, (size_t)w * h); auto dep1 = rd<uint16_t>(in, (size_t)w * h);
  const vec3 center{0.0f, 0.0f, 0.0f}; const float edge = 1.0f;
  // --- mapping: two insertions of the same cloud (octree.cpp:290), extraction (octree.cpp:336), render (cuda_renderer.cpp:163)
  unsigned int* pool = nullptr; int pool_size = 0;
  vec3* d_pts = (vec3*)up(pts); Color256* d_col = (Color256*)up(col);
  svo::svoFromPointCloud(d_pts, d_col, n, depth, pool, pool_size, center, edge);
  svo::svoFromPointCloud(d_pts, d_col, n, depth, pool, pool_size, center, edge);
  std::vector<unsigned int> words(2 * (size_t)pool_size);
  svoslam_memcpy_d2h(words.data(), pool, words.size() * 4);
  wr(out, &pool_size, 1); wr(out, words.data(), words.size());
  std::vector<uchar4_t> img((size_t)w * h); uchar4_t* d_img = up(img);
  mat4 vm; for (int i = 0; i < 16; i++) vm.m[i] = view[i];
  rendering::coneTraceSVO(d_img, vec2{(float)w, (float)h}, 45.0f, vm, SVO{pool, center, edge});
  svoslam_memcpy_d2h(img.data(), d_img, img.size() * 4);
  wr(out, img.data(), img.size());
  { VoxelGrid grid; svo::extractVoxelGridFromSVO(pool, pool_size, depth, center, edge, grid);
    std::vector<float> ce(4 * (size_t)grid.size), co(4 * (size_t)grid.size);
    svoslam_memcpy_d2h(ce.data(), grid.centers, ce.size() * 4); svoslam_memcpy_d2h(co.data(), grid.colors, co.size() * 4);
    wr(out, &grid.size, 1); wr(out, ce.data(), ce.size()); wr(out, co.data(), co.size()); }
  // --- front end: two frames -> filtered depth, vertex / normal maps, ICP system (rgbd_camera.cpp:64-141)
  sensor::ICPFrame f0(w, h), f1(w, h);
  const float fx = 570.3f * w / 640.0f;
  const uint16_t* dsrc[2] = {dep0.data(), dep1.data()}; sensor::ICPFrame* fr[2] = {&f0, &f1};
  for (int k = 0; k < 2; k++) {
    RawFrame raw(w, h); svoslam_memcpy_h2d(raw.depth, dsrc[k], (size_t)w * h * 2);
    uint16_t* filt = nullptr; svoslam_malloc((void**)&filt, (size_t)w * h * 2);
    sensor::bilateralFilter(raw.depth, filt, w, h);
    sensor::generateVertexMap(filt, fr[k]->vertex, w, h, vec2{fx, fx}, int2_t{w, h});
    sensor::generateNormalMap(fr[k]->vertex, fr[k]->normal, w, h);
    if (k == 1) {
      std::vector<uint16_t> hf((size_t)w * h); svoslam_memcpy_d2h(hf.data(), filt, hf.size() * 2); wr(out, hf.data(), hf.size());
      sensor::subsampleDepth<uint16_t>(filt, w, h);
      std::vector<uint16_t> hs((size_t)(w / 2) * (h / 2)); svoslam_memcpy_d2h(hs.data(), filt, hs.size() * 2); wr(out, hs.data(), hs.size());
    }
    svoslam_free(filt);
  }
  float A[36], b[6]; sensor::computeICPCost2(&f0, f1, A, b); wr(out, A, 36); wr(out, b, 6);
  BoundingBox bb; sensor::computePointCloudBoundingBox(f1.vertex, w * h, bb); wr(out, &bb.bbox0.x, 3); wr(out, &bb.bbox1.x, 3);
  // --- mainLoop (main.cpp:31-84) with the class mirrors: track, back-project, bbox, insert, view
  { sensor::RGBDCamera cam(w, h, vec2{fx, fx}); world::Scene scene;
    for (int k = 0; k < 2; k++) {
      RawFrame raw(w, h); raw.timestamp = k;
      svoslam_memcpy_h2d(raw.depth, dsrc[k], (size_t)w * h * 2);
      std::vector<unsigned char> rgb((size_t)w * h * 3, (unsigned char)(90 + 60 * k)); svoslam_memcpy_h2d(raw.color, rgb.data(), rgb.size());
      cam.update(&raw);
      vec3* cloud = nullptr; svoslam_malloc((void**)&cloud, (size_t)w * h * sizeof(vec3));
      sensor::generateVertexMap(raw.depth, cloud, w, h, vec2{fx, fx}, int2_t{w, h});
      const vec3 p = cam.position(); const mat3 o = cam.orientation();
      mat4 T; for (int i = 0; i < 16; i++) T.m[i] = 0.0f;                       // mat4(orientation) * translate(I, position)
      svoslam_memcpy_d2h(T.m, cam.fusionTransformDevice(), 64);
      sensor::transformVertexMap(cloud, T, w * h);
      BoundingBox bb2; sensor::computePointCloudBoundingBox(cloud, w * h, bb2);
      scene.addPointCloudToOctree(p, cloud, raw.color, w * h, bb2);
      wr(out, &p.x, 3); wr(out, o.m, 9); wr(out, T.m, 16); wr(out, &bb2.bbox0.x, 3); wr(out, &bb2.bbox1.x, 3);
      svoslam_free(cloud);
    }
    SVO s = scene.svo();
    wr(out, &s.center.x, 3); wr(out, &s.size, 1);
    rendering::coneTraceSVO(d_img, vec2{(float)w, (float)h}, 45.0f, vm, s);
    svoslam_memcpy_d2h(img.data(), d_img, img.size() * 4); wr(out, img.data(), img.size());
  }
  fclose(out); svoslam_free(d_pts); svoslam_free(d_col); svoslam_free(d_img); svoslam_free(pool);
  return 0;
}
'''


def test_reference_style_cpp_caller(tmp_path, oracle):
    import svoslam_pkg
    pkg = svoslam_pkg.load()
    synth = __import__("importlib").import_module("octree_slam_amd.synth")
    src, exe = tmp_path / "host.cpp", tmp_path / "host"
    src.write_text(HOST)
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lsvoslam_hip",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    rng = np.random.default_rng(5)
    n, depth, w, h = 20000, 8, 160, 120
    u = rng.random((n, 2))
    pts = np.stack([np.cos(2 * np.pi * u[:, 0]) * 0.6, u[:, 1] * 1.2 - 0.6, np.sin(2 * np.pi * u[:, 0]) * 0.6], -1).astype(np.float32)
    pts[::211] = np.nan
    col = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    view = oracle.look_at((0.2, 0.3, -2.2), (0, 0, 0), (0, 1, 0))
    d0 = synth.render_frame(0, w, h)[0].numpy().view(np.uint16)
    d1 = synth.render_frame(3, w, h)[0].numpy().view(np.uint16)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([n, depth, w, h], np.int32).tobytes()); f.write(pts.tobytes()); f.write(col.tobytes())
        f.write(np.asarray(view, np.float32).tobytes()); f.write(d0.tobytes()); f.write(d1.tobytes())
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = open(tmp_path / "out.bin", "rb").read()
    pos = 0

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=pos)
        pos += a.nbytes
        return a
    # oracle
    opool = oracle.Pool()
    opool.insert_cloud(pts, col, depth, (0, 0, 0), 1.0)
    opool.insert_cloud(pts, col, depth, (0, 0, 0), 1.0)
    size = int(take(np.int32, 1)[0])
    assert size == opool.size
    assert np.array_equal(take(np.uint32, 2 * size), opool.words())
    ref, _, _ = oracle.cone_trace(opool, w, h, 45.0, view, (0, 0, 0), 1.0, oracle.RENDER_REFERENCE)
    assert np.array_equal(take(np.uint8, w * h * 4).reshape(h, w, 4), ref)
    rce, rco = opool.extract(depth, (0, 0, 0), 1.0)
    k = int(take(np.int32, 1)[0])
    assert k == rce.shape[0] > 0
    assert np.array_equal(take(np.uint32, 4 * k), rce.view(np.uint32).reshape(-1)) and np.array_equal(take(np.uint32, 4 * k), rco.view(np.uint32).reshape(-1))
    f = np.float32(570.3) * np.float32(w) / np.float32(640.0)
    filt = [oracle.bilateral(d) for d in (d0, d1)]
    vm = [oracle.vertex_map(x, f, f, w, h) for x in filt]
    nm = [oracle.normal_map(v) for v in vm]
    assert np.array_equal(take(np.uint16, w * h).reshape(h, w), filt[1])
    assert np.array_equal(take(np.uint16, (w // 2) * (h // 2)).reshape(h // 2, w // 2), oracle.subsample_depth(filt[1]))
    rA, rb = oracle.icp_cost2(vm[0], nm[0], vm[1], nm[1])
    assert np.array_equal(take(np.float32, 36), rA.reshape(-1)) and np.array_equal(take(np.float32, 6), rb)
    b0, b1 = oracle.point_cloud_bbox(vm[1].reshape(-1, 3))
    assert np.array_equal(take(np.float32, 3), b0) and np.array_equal(take(np.float32, 3), b1)
    # main loop with RGBDCamera + Scene
    ocam = oracle.Camera(w, h, float(f), float(f))
    opool2, depth2, c2, size2 = None, None, None, None
    for k, d in enumerate((d0, d1)):
        rgb = np.full((h, w, 3), 90 + 60 * k, np.uint8)
        assert ocam.update(d, rgb, k) == 1
        rp, ro = ocam.pose()
        assert np.array_equal(take(np.float32, 3), rp) and np.array_equal(take(np.float32, 9), ro)
        T = ocam.fusion_transform()
        assert np.array_equal(take(np.float32, 16), T)
        cloud = oracle.transform_vertex_map(oracle.vertex_map(d, f, f, w, h), T).reshape(-1, 3)
        b0, b1 = oracle.point_cloud_bbox(cloud)
        assert np.array_equal(take(np.float32, 3), b0) and np.array_equal(take(np.float32, 3), b1)
        if k == 0:   # Scene::addPointCloudToOctree: tree from the first cloud's box, resolution 0.01 (scene.cpp:100-103)
            c2, size2 = (b1 + b0) / np.float32(2.0), float(b1[0])
            depth2 = int(np.ceil(np.log2(np.float64(np.float32(size2) / np.float32(0.01)))))
            opool2 = oracle.Pool()
        else:        # scene.cpp:104-108 + Octree::expandBySize (octree.cpp:362-378): the root is only rescaled (Q16)
            t0, t1 = c2 - np.float32(size2), c2 + np.float32(size2)
            if not ((t0 <= b0).all() and (t1 >= b1).all()):
                r = np.float32(max(0.0, *(t0 - b0), *(b1 - t1)))
                layers = int(np.log(np.ceil((np.float32(size2) + r) / np.float32(size2))) / np.log(np.float32(2.0)))
                assert layers >= 1      # this stream does leave the first frame's box
                size2 = float(np.float32(2.0) ** np.float32(layers) * np.float32(size2))
                depth2 = int(np.ceil(np.log2(np.float64(np.float32(size2) / np.float32(0.01)))))
        opool2.insert_cloud(cloud, rgb.reshape(-1, 3), depth2, c2, size2)
    assert np.array_equal(take(np.float32, 3), c2) and take(np.float32, 1)[0] == np.float32(size2)
    ref2, _, _ = oracle.cone_trace(opool2, w, h, 45.0, view, c2, size2, oracle.RENDER_REFERENCE)
    assert np.array_equal(take(np.uint8, w * h * 4).reshape(h, w, 4), ref2)
    assert pos == len(raw)
