// octree_slam_compat.hpp -- header-only C++ shim that re-exposes the reference's GPU kernel
// API (same names, argument order and meaning) on top of the C ABI of include/svoslam.h, so
// callers written like the reference's octree.cpp / rgbd_camera.cpp / cuda_renderer.cpp compile
// against libsvoslam_hip.so unchanged.  glm is not required: vec2/vec3/vec4/mat4 below are
// layout-compatible PODs (glm::vec3 = 12 packed bytes, glm::mat4 = 16 floats column-major).
//
// Reference declarations mirrored:
//   include/octree_slam/world/svo/svo.h:14-18
//   include/octree_slam/rendering/cone_tracing_kernels.h:16
//   include/octree_slam/sensor/image_kernels.h:21-55
//   include/octree_slam/sensor/localization_kernels.h:17-42
//   include/octree_slam/timing_utils.h:5-10
#pragma once

#include <stdint.h>

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "svoslam.h"

namespace octree_slam {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { float m[16]; };  // column-major, m[4*col + row]
struct mat3 { float m[9]; };   // column-major, m[3*col + row]
struct int2_t { int x, y; };
struct uchar4_t { unsigned char x, y, z, w; };

struct Color256 { uint8_t r, g, b; };                         // common_types.h:49-53
struct BoundingBox {                                          // common_types.h:8-18, common_types.cu:8-34
  vec3 bbox0{0, 0, 0}, bbox1{0, 0, 0};
  bool contains(const BoundingBox &o) const {
    return bbox0.x <= o.bbox0.x && bbox0.y <= o.bbox0.y && bbox0.z <= o.bbox0.z && bbox1.x >= o.bbox1.x && bbox1.y >= o.bbox1.y &&
           bbox1.z >= o.bbox1.z;
  }
  float distanceOutside(const BoundingBox &o) const {           // as written in the reference (:22-34), operand order kept
    float r = 0.0f;
    r = r > o.bbox0.x - bbox0.x ? r : o.bbox0.x - bbox0.x; r = r > o.bbox0.y - bbox0.y ? r : o.bbox0.y - bbox0.y;
    r = r > o.bbox0.z - bbox0.z ? r : o.bbox0.z - bbox0.z; r = r > bbox1.x - o.bbox1.x ? r : bbox1.x - o.bbox1.x;
    r = r > bbox1.y - o.bbox1.y ? r : bbox1.y - o.bbox1.y; r = r > bbox1.z - o.bbox1.z ? r : bbox1.z - o.bbox1.z;
    return r;
  }
};
struct VoxelGrid {                                            // common_types.h:55-63
  vec4 *centers = nullptr;
  vec4 *colors = nullptr;
  int size = 0;
  float scale = 0.0f;
  BoundingBox bbox;
  ~VoxelGrid() { if (size > 0) { svoslam_free(centers); svoslam_free(colors); } }
};
struct RawFrame {                                             // common_types.h:65-73, common_types.cu:36-45
  RawFrame(const int w, const int h) : height(h), width(w) {
    void *c = nullptr, *d = nullptr;
    if (svoslam_malloc(&c, (size_t)w * h * sizeof(Color256)) != 0 || svoslam_malloc(&d, (size_t)w * h * sizeof(uint16_t)) != 0) {
      svoslam_free(c);
      throw std::runtime_error("RawFrame: device allocation failed");
    }
    color = static_cast<Color256 *>(c); depth = static_cast<uint16_t *>(d);
  }
  ~RawFrame() { svoslam_free(color); svoslam_free(depth); }
  RawFrame(const RawFrame &) = delete;
  RawFrame &operator=(const RawFrame &) = delete;
  Color256 *color = nullptr;
  uint16_t *depth = nullptr;
  int height, width;
  long long timestamp = 0;
};
struct SVO { unsigned int *data; vec3 center; float size; };  // common_types.h:75-79

namespace detail {
inline void check(int status, const char *what) {
  if (status != SVOSLAM_OK)
    throw std::runtime_error(std::string(what) + ": " + svoslam_status_string(status) + " (" + svoslam_last_error() + ")");
}
// The reference passes the pool as `unsigned int*& octree, int& octree_size`; capacity is kept here.
struct Registry {
  std::mutex mu;
  std::map<unsigned int *, int> capacity;
  std::mutex ws_mu;                                           // the one workspace serves one call at a time
  svoslam_workspace *ws = nullptr;
  static Registry &get() { static Registry r; return r; }
  // the shared workspace, locked until the end of the full expression (the blocking call it is passed to): callers
  // on several host threads are serialised here instead of racing on its scratch buffers
  struct LockedWorkspace {
    std::unique_lock<std::mutex> held;
    svoslam_workspace *ws;
    operator svoslam_workspace *() const { return ws; }
  };
  LockedWorkspace workspace() {
    std::unique_lock<std::mutex> held(ws_mu);
    if (!ws) check(svoslam_workspace_create(&ws), "svoslam_workspace_create");
    return LockedWorkspace{std::move(held), ws};
  }
  svoslam_pool open(unsigned int *data, int size) {
    std::lock_guard<std::mutex> g(mu);
    svoslam_pool p{data, size, size, nullptr, 0, 0};
    auto it = capacity.find(data);
    if (it != capacity.end() && it->second >= size) p.capacity = it->second;
    return p;
  }
  void close(unsigned int *old_data, const svoslam_pool &p) {
    std::lock_guard<std::mutex> g(mu);
    if (old_data != p.d_data) capacity.erase(old_data);
    capacity[p.d_data] = p.capacity;
  }
};
}  // namespace detail

namespace svo {
// svo.h:16 / svo.cu:642-696.  `octree` may be replaced by a larger allocation (the old one is freed),
// `octree_size` is updated, exactly as the reference does; the caller frees with cudaFree/hipFree.
inline void svoFromPointCloud(const vec3 *points, const Color256 *colors, const int size, const int max_depth,
                              unsigned int *&octree, int &octree_size, vec3 octree_center, const float edge_length,
                              void * /*cudaArray* d_bricks*/ = nullptr) {
  auto &r = detail::Registry::get();
  svoslam_pool p = r.open(octree, octree_size);
  unsigned int *old = octree;
  const float c[3] = {octree_center.x, octree_center.y, octree_center.z};
  detail::check(svoslam_svo_from_point_cloud(r.workspace(), &points->x, &colors->r, size, max_depth, &p, c, edge_length,
                                             nullptr, nullptr), "svoFromPointCloud");
  r.close(old, p);
  octree = p.d_data;
  octree_size = p.size;
}
// svo.h:14 / svo.cu:584-640
inline void svoFromVoxelGrid(const VoxelGrid &grid, const int max_depth, unsigned int *&octree, int &octree_size,
                             vec3 octree_center, const float edge_length, void * = nullptr) {
  auto &r = detail::Registry::get();
  svoslam_pool p = r.open(octree, octree_size);
  unsigned int *old = octree;
  const float c[3] = {octree_center.x, octree_center.y, octree_center.z};
  detail::check(svoslam_svo_from_voxel_grid(r.workspace(), &grid.centers->x, &grid.colors->x, grid.size, max_depth, &p, c,
                                            edge_length, nullptr, nullptr), "svoFromVoxelGrid");
  r.close(old, p);
  octree = p.d_data;
  octree_size = p.size;
}
// svo.h:18 / svo.cu:699-745
inline void extractVoxelGridFromSVO(unsigned int *&octree, int &octree_size, const int max_depth, const vec3 center,
                                    float edge_length, VoxelGrid &grid) {
  auto &r = detail::Registry::get();
  svoslam_pool p = r.open(octree, octree_size);
  const float c[3] = {center.x, center.y, center.z};
  float *ce = nullptr, *co = nullptr;
  int32_t n = 0;
  detail::check(svoslam_extract_voxel_grid(r.workspace(), &p, max_depth, c, edge_length, &ce, &co, &n, nullptr),
                "extractVoxelGridFromSVO");
  grid.centers = reinterpret_cast<vec4 *>(ce);
  grid.colors = reinterpret_cast<vec4 *>(co);
  grid.size = n;
}
}  // namespace svo

namespace rendering {
// cone_tracing_kernels.h:16 / cone_tracing_kernels.cu:157-198 (pos = mapped PBO in the reference,
// any device uchar4 buffer here)
inline void coneTraceSVO(uchar4_t *pos, vec2 resolution, float fov, mat4 cameraPose, SVO octree) {
  const float c[3] = {octree.center.x, octree.center.y, octree.center.z};
  detail::check(svoslam_cone_trace_svo(&pos->x, (int)resolution.x, (int)resolution.y, fov, cameraPose.m, octree.data, c,
                                       octree.size, SVOSLAM_RENDER_REFERENCE, nullptr, nullptr), "coneTraceSVO");
}
}  // namespace rendering

namespace sensor {
struct ICPFrame {  // localization_kernels.h:17-24
  ICPFrame(const int w, const int h) : width(w), height(h) {
    detail::check(svoslam_malloc((void **)&vertex, sizeof(vec3) * (size_t)w * h), "ICPFrame");
    detail::check(svoslam_malloc((void **)&normal, sizeof(vec3) * (size_t)w * h), "ICPFrame");
  }
  ~ICPFrame() { svoslam_free(vertex); svoslam_free(normal); }
  vec3 *vertex; vec3 *normal; int width; int height;
};
struct RGBDFrame {  // localization_kernels.h:26-33
  RGBDFrame(const int w, const int h) : width(w), height(h) {
    detail::check(svoslam_malloc((void **)&intensity, sizeof(float) * (size_t)w * h), "RGBDFrame");
    detail::check(svoslam_malloc((void **)&vertex, sizeof(vec3) * (size_t)w * h), "RGBDFrame");
  }
  ~RGBDFrame() { svoslam_free(intensity); svoslam_free(vertex); }
  float *intensity; vec3 *vertex; int width; int height;
};
// localization_kernels.h:42 / localization_kernels.cu:328-331: an empty stub in the reference (A and b are left untouched);
// kept so that callers link
// localization_kernels.h:42: an empty body in the reference (localization_kernels.cu:328-331), kept so (the frames
// carry no intrinsics).  The overload below is this library's own photometric term (svoslam.h "Photometric RGB-D term"):
// last_gradient = gradient() of last_frame->intensity; fx, fy in pixels of the full img_width x img_height image.
inline void computeRGBDCost(const RGBDFrame *, const RGBDFrame &, float *, float *) {}
inline void computeRGBDCost(const RGBDFrame *last_frame, const vec2 *last_gradient, const RGBDFrame &this_frame, const vec2 &focal_length,
                            const int img_width, const int img_height, float *A, float *b) {
  detail::check(svoslam_rgbd_cost(last_frame->intensity, &last_gradient->x, &last_frame->vertex->x, this_frame.intensity,
                                  &this_frame.vertex->x, this_frame.width, this_frame.height, focal_length.x, focal_length.y, img_width,
                                  img_height, A, b, nullptr), "computeRGBDCost");
}
// image_kernels.h:24-55
inline void generateVertexMap(const uint16_t *depth_pixels, vec3 *vertex_map, const int width, const int height,
                              const vec2 focal_length, const int2_t img_size) {
  detail::check(svoslam_generate_vertex_map(depth_pixels, &vertex_map->x, width, height, focal_length.x, focal_length.y,
                                            img_size.x, img_size.y, nullptr), "generateVertexMap");
}
inline void generateNormalMap(const vec3 *vertex_map, vec3 *normal_map, const int width, const int height) {
  detail::check(svoslam_generate_normal_map(&vertex_map->x, &normal_map->x, width, height, nullptr), "generateNormalMap");
}
inline void bilateralFilter(const uint16_t *depth_in, uint16_t *filtered_out, const int width, const int height) {
  detail::check(svoslam_bilateral_filter(depth_in, filtered_out, width, height, nullptr), "bilateralFilter");
}
inline void computePointCloudBoundingBox(vec3 *points, const int num_points, BoundingBox &bbox) {
  detail::check(svoslam_point_cloud_bbox(&points->x, num_points, &bbox.bbox0.x, &bbox.bbox1.x, nullptr),
                "computePointCloudBoundingBox");
}
inline void transformVertexMap(vec3 *vertex_map, const mat4 &trans, const int size) {
  detail::check(svoslam_transform_vertex_map(&vertex_map->x, trans.m, size, nullptr), "transformVertexMap");
}
inline void transformNormalMap(vec3 *normal_map, const mat4 &trans, const int size) {
  detail::check(svoslam_transform_normal_map(&normal_map->x, trans.m, size, nullptr), "transformNormalMap");
}
// image_kernels.h:36-42: in place, the (width/2 x height/2) result overwrites the head of data
namespace sub_detail {
template <class T, class F> inline void subsample_with(F f, T *data, int width, int height, const char *what) {
  void *tmp = nullptr;
  octree_slam::detail::check(svoslam_malloc(&tmp, (size_t)(width / 2) * (height / 2) * sizeof(T) + 16), what);
  const int rc = f(data, static_cast<T *>(tmp), width, height, nullptr);
  svoslam_free(tmp);
  octree_slam::detail::check(rc, what);
}
}  // namespace sub_detail
template <class T> void subsample(T *data, const int width, const int height);
template <> inline void subsample<float>(float *data, const int width, const int height) {
  sub_detail::subsample_with<float>(svoslam_subsample_f32, data, width, height, "subsample<float>");
}
template <> inline void subsample<Color256>(Color256 *data, const int width, const int height) {
  void *tmp = nullptr;
  octree_slam::detail::check(svoslam_malloc(&tmp, (size_t)(width / 2) * (height / 2) * 3 + 16), "subsample<Color256>");
  const int rc = svoslam_subsample_rgb8(&data->r, static_cast<uint8_t *>(tmp), width, height, nullptr);
  svoslam_free(tmp);
  octree_slam::detail::check(rc, "subsample<Color256>");
}
template <class T> void subsampleDepth(T *data, const int width, const int height);
template <> inline void subsampleDepth<uint16_t>(uint16_t *data, const int width, const int height) {
  sub_detail::subsample_with<uint16_t>(svoslam_subsample_depth_u16, data, width, height, "subsampleDepth<uint16_t>");
}
template <> inline void subsampleDepth<float>(float *data, const int width, const int height) {
  sub_detail::subsample_with<float>(svoslam_subsample_depth_f32, data, width, height, "subsampleDepth<float>");
}
// image_kernels.h:45-49: declared by the reference, defined nowhere; own specification (Sobel / 8; in1 - in2)
inline void gradient(const float *intensity_in, vec2 *gradient_out, const int width, const int height) {
  detail::check(svoslam_gradient(intensity_in, &gradient_out->x, width, height, nullptr), "gradient");
}
inline void difference(const float *in1, const float *in2, float *out, const int size) {
  detail::check(svoslam_difference(in1, in2, out, size, nullptr), "difference");
}
inline void colorToIntensity(const Color256 *color_in, float *intensity_out, const int size) {
  detail::check(svoslam_color_to_intensity(&color_in->r, intensity_out, size, nullptr), "colorToIntensity");
}
// localization_kernels.h:38 : the correspondence variant; A and b untouched without correspondences
inline void computeICPCost(const ICPFrame *last_frame, const ICPFrame &this_frame, float *A, float *b) {
  detail::check(svoslam_icp_cost(&last_frame->vertex->x, &last_frame->normal->x, &this_frame.vertex->x, &this_frame.normal->x,
                                 this_frame.width, this_frame.height, A, b, nullptr, nullptr), "computeICPCost");
}
// localization_kernels.h:39 : A (36) and b (6) are host arrays
inline void computeICPCost2(const ICPFrame *last_frame, const ICPFrame &this_frame, float *A, float *b) {
  detail::check(svoslam_icp_cost2(&last_frame->vertex->x, &last_frame->normal->x, &this_frame.vertex->x, &this_frame.normal->x,
                                  this_frame.width, this_frame.height, A, b, nullptr), "computeICPCost2");
}
// rgbd_camera.h:17-84 / rgbd_camera.cpp:41-222: the same state machine, resident on the device (no host round trips)
class RGBDCamera {
public:
  RGBDCamera(const int width, const int height, const vec2 &focal_length) {
    detail::check(svoslam_camera_create(&cam_, width, height, focal_length.x, focal_length.y), "RGBDCamera");
  }
  ~RGBDCamera() { svoslam_camera_destroy(cam_); }
  RGBDCamera(const RGBDCamera &) = delete;
  RGBDCamera &operator=(const RGBDCamera &) = delete;
  const vec3 position() const { pose(); return position_; }
  const mat3 orientation() const { pose(); return orientation_; }
  // frames whose timestamp is not newer than the latest processed one are skipped (rgbd_camera.cpp:55-58)
  void update(const RawFrame *this_frame) {
    int32_t used = 0;
    detail::check(svoslam_camera_update(cam_, this_frame->depth, &this_frame->color->r, this_frame->timestamp, &used, nullptr),
                  "RGBDCamera::update");
  }
  // main.cpp:40: mat4(orientation) * translate(mat4(1), position), as a device pointer (stays valid; updated by update())
  const float *fusionTransformDevice() const { return svoslam_camera_fusion_transform_device(cam_); }
  svoslam_camera *handle() const { return cam_; }

private:
  void pose() const { detail::check(svoslam_camera_pose(cam_, &position_.x, orientation_.m, nullptr), "RGBDCamera::pose"); }
  svoslam_camera *cam_ = nullptr;
  mutable vec3 position_{0, 0, 0};
  mutable mat3 orientation_{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
};
}  // namespace sensor


// common_types.h:20-38 (host arrays; non-indexed triangles as Scene::loadObjFile fills them)
struct Mesh {
  int vbosize = 0, nbosize = 0, cbosize = 0, ibosize = 0, tbosize = 0;
  float *vbo = nullptr, *nbo = nullptr, *cbo = nullptr;
  int *ibo = nullptr;
  float *tbo = nullptr;
  BoundingBox bbox;
};
struct bmp_texture { vec3 *data; int width, height; };

namespace voxelization {
// voxelization.h:16-17 / voxelization.cu:24-25
inline int log_N() { return 8; }
inline int log_T() { return 3; }
// voxelization.h:21 / voxelization.cu:381-405
inline void meshToVoxelGrid(const Mesh &m_in, const bmp_texture *tex, VoxelGrid &grid_out) {
  svoslam_mesh m;
  m.vbo = m_in.vbo; m.tbo = m_in.tbo; m.n_tris = m_in.vbosize / 9; m.tbosize = m_in.tbosize;
  m.bbox0[0] = m_in.bbox.bbox0.x; m.bbox0[1] = m_in.bbox.bbox0.y; m.bbox0[2] = m_in.bbox.bbox0.z;
  m.bbox1[0] = m_in.bbox.bbox1.x; m.bbox1[1] = m_in.bbox.bbox1.y; m.bbox1[2] = m_in.bbox.bbox1.z;
  svoslam_texture t{nullptr, 0, 0};
  if (tex) { t.data = &tex->data->x; t.width = tex->width; t.height = tex->height; }
  float *ce = nullptr, *co = nullptr, scale = 0.0f;
  int32_t n = 0;
  detail::check(svoslam_mesh_to_voxel_grid(detail::Registry::get().workspace(), &m, tex ? &t : nullptr, log_N(), log_T(), &ce, &co,
                                           nullptr, &n, &scale, nullptr), "meshToVoxelGrid");
  grid_out.centers = reinterpret_cast<vec4 *>(ce);
  grid_out.colors = reinterpret_cast<vec4 *>(co);
  grid_out.size = n;
  grid_out.scale = scale;
  grid_out.bbox = m_in.bbox;
}
// voxelization.h:13,19 / voxelization.cu:325-379: m_out's arrays are malloc'ed host memory, as in the reference
const float CUBE_MESH_SCALE = 0.1f;
inline void voxelGridToMesh(const VoxelGrid &grid, const Mesh &m_cube, Mesh &m_out) {
  if (m_cube.vbosize != m_cube.nbosize) throw std::runtime_error("voxelGridToMesh: cube vbo and nbo have different sizes");
  const size_t nv = (size_t)grid.size * m_cube.vbosize, ni = (size_t)grid.size * m_cube.ibosize;
  void *dv = nullptr, *di = nullptr, *dn = nullptr, *dc = nullptr;
  detail::check(svoslam_malloc(&dv, (nv ? nv : 1) * 4), "voxelGridToMesh"); detail::check(svoslam_malloc(&dn, (nv ? nv : 1) * 4), "voxelGridToMesh");
  detail::check(svoslam_malloc(&dc, (nv ? nv : 1) * 4), "voxelGridToMesh"); detail::check(svoslam_malloc(&di, (ni ? ni : 1) * 4), "voxelGridToMesh");
  const float scale = (grid.bbox.bbox1.x - grid.bbox.bbox0.x) / float(1 << log_N()) / 2.0f / CUBE_MESH_SCALE;  // computeScale / CUBE_MESH_SCALE
  detail::check(svoslam_voxel_grid_to_mesh(detail::Registry::get().workspace(), &grid.centers->x, &grid.colors->x, grid.size, scale,
                                           m_cube.vbo, m_cube.vbosize, m_cube.ibo, m_cube.ibosize, m_cube.nbo, (float *)dv, (int32_t *)di,
                                           (float *)dn, (float *)dc, nullptr), "voxelGridToMesh");
  m_out.vbosize = (int)nv; m_out.ibosize = (int)ni; m_out.nbosize = (int)nv; m_out.cbosize = (int)nv;
  m_out.vbo = (float *)malloc((nv ? nv : 1) * 4); m_out.ibo = (int *)malloc((ni ? ni : 1) * 4);
  m_out.nbo = (float *)malloc((nv ? nv : 1) * 4); m_out.cbo = (float *)malloc((nv ? nv : 1) * 4);
  detail::check(svoslam_memcpy_d2h(m_out.vbo, dv, nv * 4), "voxelGridToMesh"); detail::check(svoslam_memcpy_d2h(m_out.ibo, di, ni * 4), "voxelGridToMesh");
  detail::check(svoslam_memcpy_d2h(m_out.nbo, dn, nv * 4), "voxelGridToMesh"); detail::check(svoslam_memcpy_d2h(m_out.cbo, dc, nv * 4), "voxelGridToMesh");
  svoslam_free(dv); svoslam_free(di); svoslam_free(dn); svoslam_free(dc);
}
}  // namespace voxelization
namespace world {
// scene.h:22-79 / scene.cpp: OBJ / BMP loading, mesh voxelization into the octree, point-cloud insertion, SVO view
class Scene {
public:
  Scene() { detail::check(svoslam_scene_create(&scene_), "Scene"); }
  ~Scene() { svoslam_scene_destroy(scene_); }
  Scene(const Scene &) = delete;
  Scene &operator=(const Scene &) = delete;
  void loadObjFile(const std::string &filename) { detail::check(svoslam_scene_load_obj(scene_, filename.c_str()), "Scene::loadObjFile"); }
  void loadBMP(const std::string &filename) { detail::check(svoslam_scene_load_bmp(scene_, filename.c_str()), "Scene::loadBMP"); }
  void voxelizeMeshes(const bool octree = false) {
    detail::check(svoslam_scene_voxelize_meshes(scene_, octree ? 1 : 0, 0, nullptr), "Scene::voxelizeMeshes");
  }
  void extractVoxelGridFromOctree() { detail::check(svoslam_scene_extract_voxel_grid(scene_, nullptr), "Scene::extractVoxelGridFromOctree"); }
  void addPointCloudToOctree(const vec3 &origin, const vec3 *points, const Color256 *colors, const int size, const BoundingBox &bbox) {
    detail::check(svoslam_scene_add_point_cloud(scene_, &origin.x, &points->x, &colors->r, size, &bbox.bbox0.x, &bbox.bbox1.x, nullptr),
                  "Scene::addPointCloudToOctree");
  }
  // non-owning view of the scene's voxel grid (the reference returns a reference to its own VoxelGrid)
  struct VoxelGridView { const vec4 *centers; const vec4 *colors; int size; float scale; };
  VoxelGridView voxel_grid() const {
    const float *ce = nullptr, *co = nullptr; int32_t n = 0; float scale = 0.0f;
    detail::check(svoslam_scene_voxel_grid(scene_, &ce, &co, &n, &scale), "Scene::voxel_grid");
    return VoxelGridView{reinterpret_cast<const vec4 *>(ce), reinterpret_cast<const vec4 *>(co), n, scale};
  }
  // Octree::extractSVO (octree.cpp:339-360) ignores its bbox argument as well
  SVO svo(const BoundingBox & = BoundingBox()) const {
    const uint32_t *data = nullptr; float c[3] = {0, 0, 0}, size = 0.0f; int32_t nodes = 0, depth = 0;
    detail::check(svoslam_scene_svo(scene_, &data, c, &size, &nodes, &depth), "Scene::svo");
    return SVO{const_cast<unsigned int *>(data), vec3{c[0], c[1], c[2]}, size};
  }
  svoslam_scene *handle() const { return scene_; }

private:
  svoslam_scene *scene_ = nullptr;
};
}  // namespace world

// timing_utils.h:5-10
inline void startTiming() { detail::check(svoslam_timer_start(nullptr), "startTiming"); }
inline float stopTiming() { float ms = 0; detail::check(svoslam_timer_stop(nullptr, &ms), "stopTiming"); return ms; }

}  // namespace octree_slam
