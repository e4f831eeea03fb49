// host_scene.hip -- host mirror of world::Octree and world::Scene above the C ABI
// (include/octree_slam/world/octree.h:80-125, src/world/octree.cpp:251-385;
//  include/octree_slam/world/scene.h:20-81, src/world/scene.cpp:11-133).
//
// Same entry methods, same arithmetic for the root parameters and the tree depth, same state
// machine (first cloud creates the tree from its bounding box, later clouds expand it).  The
// OctreeNode CPU<->GPU paging machinery (octree.cpp:41-247) is not mirrored: in the reference
// the root is always the GPU node, so getNodeContainingBoundingBox() always returns it
// (octree.cpp:208-212) and every call lands on the single device pool kept here.
//
// Tree depth: max_depth = ceil(log(edge/resolution)/log 2) (octree.cpp:284,306,330) is evaluated
// exactly from the binary32 exponent of the rounded quotient (the reference's float log can be one
// level off at powers of two); an explicit depth can be set instead.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "host_scene.hpp"
#include "pool_grid.hpp"
#include "mesh.hpp"
#include "svo_build.hpp"

namespace octree_slam {
namespace world {

static int ceil_log2_of(float q) {
  uint32_t u;
  memcpy(&u, &q, 4);
  if ((int32_t)u <= 0) return 0;
  const int ex = (int)(u >> 23);
  const uint32_t man = u & 0x7FFFFFu;
  if (ex == 255) return 128;
  if (ex == 0) return -126;
  return (ex - 127) + (man != 0);
}

// ---- Octree -------------------------------------------------------------------------------
Octree::Octree(const float resolution, const float center[3], const float size) : size_(size), resolution_(resolution) {
  for (int k = 0; k < 3; k++) center_[k] = center[k];
  pool_.d_data = nullptr; pool_.size = 0; pool_.capacity = 0;
  pool_.d_size = nullptr; pool_.pending = 0; pool_.pending_bound = 0;
}

Octree::~Octree() {
  svoslam::pool_accel_unregister(&pool_);
  if (pool_.d_data) (void)hipFree(pool_.d_data);
  if (pool_.d_size) (void)hipFree(pool_.d_size);
}

int Octree::maxDepth(float edge_length, float resolution) const {
  if (depth_override_ > 0) return depth_override_;
  return ceil_log2_of(edge_length / resolution);
}

void Octree::boundingBox(float bbox0[3], float bbox1[3]) const {  // octree.cpp:380-385
  for (int k = 0; k < 3; k++) { bbox0[k] = center_[k] - size_; bbox1[k] = center_[k] + size_; }
}

// octree.cpp:269-291 (the subtree is always the root: edge_length = size_ / pow(2, 0))
int Octree::addCloud(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int size, hipStream_t s) {
  const float edge_length = size_ / powf(2.0f, 0.0f);
  const int max_depth = maxDepth(edge_length, resolution_);
  last_depth_ = max_depth;
  return svoslam::svo_from_point_cloud(ws, d_points, d_colors, size, max_depth, &pool_, center_, edge_length, nullptr, s);
}

// octree.cpp:293-313
int Octree::addVoxelGrid(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int n, hipStream_t s) {
  const float edge_length = size_ / powf(2.0f, 0.0f);
  const int max_depth = maxDepth(edge_length, resolution_);
  last_depth_ = max_depth;
  return svoslam::svo_from_voxel_grid(ws, d_centers, d_colors, n, max_depth, &pool_, center_, edge_length, nullptr, s);
}

// octree.cpp:315-337 : depth from grid.scale instead of the resolution
int Octree::extractVoxelGrid(svoslam_workspace *ws, float grid_scale, float **d_centers, float **d_colors, int32_t *n, hipStream_t s) {
  const float edge_length = size_ / powf(2.0f, 0.0f);
  const int max_depth = maxDepth(edge_length, grid_scale);
  return svoslam::extract_voxel_grid(ws, &pool_, max_depth, center_, edge_length, d_centers, d_colors, n, s);
}

// octree.cpp:362-378.  On a GPU-backed root OctreeNode::expand() has nothing to re-root, so the net
// effect in the reference is the rescale of size_ (Q16); mirrored as is.
void Octree::expandBySize(const float add_size) {
  const int add_layers = (int)(logf(ceilf((size_ + add_size) / size_)) / logf(2.0f));
  if (add_layers < 1) return;
  size_ = powf(2.0f, (float)add_layers) * size_;
}

// ---- Scene --------------------------------------------------------------------------------
Scene::Scene() : tree_(nullptr) {
  memset(&grid_, 0, sizeof(grid_));
  svoslam_workspace_create(&ws_);
}

Scene::~Scene() {
  delete tree_;
  freeGrid();
  for (auto &m : meshes_) svoslam::mesh_free(&m);
  for (auto &t : textures_) svoslam::texture_free(&t);
  svoslam_workspace_destroy(ws_);
}

void Scene::freeGrid() {
  if (grid_.size > 0) { (void)hipFree(grid_.d_centers); (void)hipFree(grid_.d_colors); }
  grid_.d_centers = grid_.d_colors = nullptr;
  grid_.size = 0;
}

int Scene::loadObjFile(const char *filename) {  // scene.cpp:26-33
  svoslam_mesh m;
  SVO_TRY(svoslam::mesh_load_obj(filename, &m));
  meshes_.push_back(m);
  return SVOSLAM_OK;
}

int Scene::loadBMP(const char *filename) {  // scene.cpp:35-62
  svoslam_texture t;
  SVO_TRY(svoslam::texture_load_bmp(filename, &t));
  textures_.push_back(t);
  return SVOSLAM_OK;
}

// scene.cpp:64-85 ; log_N <= 0 -> 8 (GRID_RES, voxelization.cu:24)
int Scene::voxelizeMeshes(const bool octree, int log_N, hipStream_t s) {
  if (meshes_.empty()) return SVOSLAM_OK;
  if (log_N <= 0) log_N = 8;
  const svoslam_mesh &m = meshes_[0];
  const svoslam_texture *tex = textures_.empty() ? nullptr : &textures_[0];
  float *ce = nullptr, *co = nullptr;
  int32_t n = 0;
  float mesh_scale = 0;
  SVO_TRY(svoslam::mesh_to_voxel_grid(ws_, &m, tex, log_N, 3 < log_N ? 3 : log_N, &ce, &co, nullptr, &n, &mesh_scale, s));
  const float scale = m.bbox1[0] / (float)(1 << log_N);  // scene.cpp:72,76 (overrides computeScale)
  if (!octree) {
    freeGrid();
    grid_.d_centers = ce; grid_.d_colors = co; grid_.size = n; grid_.scale = scale;
    memcpy(grid_.bbox0, m.bbox0, 12); memcpy(grid_.bbox1, m.bbox1, 12);
    return SVOSLAM_OK;
  }
  grid_.scale = scale;
  if (!tree_) {
    const float center[3] = {(m.bbox1[0] + m.bbox0[0]) / 2.0f, (m.bbox1[1] + m.bbox0[1]) / 2.0f, (m.bbox1[2] + m.bbox0[2]) / 2.0f};
    tree_ = new Octree(grid_.scale, center, m.bbox1[0]);  // scene.cpp:77-79
    tree_->setDepthOverride(depth_override_);
  }
  int st = tree_->addVoxelGrid(ws_, ce, co, n, s);
  if (n > 0) { (void)hipFree(ce); (void)hipFree(co); }
  SVO_TRY(st);
  tree_->boundingBox(grid_.bbox0, grid_.bbox1);
  grid_.scale *= (float)1;
  const float keep_scale = grid_.scale;
  freeGrid();
  grid_.scale = keep_scale;
  float *ec = nullptr, *ek = nullptr;
  int32_t en = 0;
  SVO_TRY(tree_->extractVoxelGrid(ws_, grid_.scale, &ec, &ek, &en, s));  // scene.cpp:83
  grid_.d_centers = ec; grid_.d_colors = ek; grid_.size = en;
  return SVOSLAM_OK;
}

int Scene::extractVoxelGridFromOctree(hipStream_t s) {  // scene.cpp:87-96
  if (!tree_) return SVOSLAM_ERR_INVALID_ARG;
  freeGrid();
  tree_->boundingBox(grid_.bbox0, grid_.bbox1);
  grid_.scale = 0.01f;
  float *ec = nullptr, *ek = nullptr;
  int32_t en = 0;
  SVO_TRY(tree_->extractVoxelGrid(ws_, grid_.scale, &ec, &ek, &en, s));
  grid_.d_centers = ec; grid_.d_colors = ek; grid_.size = en;
  return SVOSLAM_OK;
}

// scene.cpp:98-113
int Scene::addPointCloudToOctree(const float * /*origin*/, const float *d_points, const uint8_t *d_colors, const int size,
                                 const float bbox0[3], const float bbox1[3], hipStream_t s) {
  if (!tree_) {
    const float center[3] = {(bbox1[0] + bbox0[0]) / 2.0f, (bbox1[1] + bbox0[1]) / 2.0f, (bbox1[2] + bbox0[2]) / 2.0f};
    tree_ = new Octree(0.01f, center, bbox1[0]);  // Q16: size = max-x coordinate, not a half extent
    tree_->setDepthOverride(depth_override_);
  } else {
    float t0[3], t1[3];
    tree_->boundingBox(t0, t1);
    const bool contains = t0[0] <= bbox0[0] && t0[1] <= bbox0[1] && t0[2] <= bbox0[2] && t1[0] >= bbox1[0] && t1[1] >= bbox1[1] &&
                          t1[2] >= bbox1[2];  // BoundingBox::contains, common_types.cu:8-21
    if (!contains) {
      // BoundingBox::distanceOutside(tree box), common_types.cu:23-36, called on the CLOUD box
      float r = 0.0f;
      r = fmaxf(r, t0[0] - bbox0[0]); r = fmaxf(r, t0[1] - bbox0[1]); r = fmaxf(r, t0[2] - bbox0[2]);
      r = fmaxf(r, bbox1[0] - t1[0]); r = fmaxf(r, bbox1[1] - t1[1]); r = fmaxf(r, bbox1[2] - t1[2]);
      tree_->expandBySize(r);
    }
  }
  return tree_->addCloud(ws_, d_points, d_colors, size, s);
}

int Scene::setOctree(float resolution, const float center[3], float size, int depth_override) {
  if (tree_) return SVOSLAM_ERR_INVALID_ARG;
  tree_ = new Octree(resolution, center, size);
  depth_override_ = depth_override;
  tree_->setDepthOverride(depth_override);
  return SVOSLAM_OK;
}

}  // namespace world
}  // namespace octree_slam

// ---- C API -----------------------------------------------------------------------------------
using octree_slam::world::Scene;
struct svoslam_scene { Scene impl; };

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

int svoslam_scene_create(svoslam_scene **scene) {
  if (!scene) return SVOSLAM_ERR_INVALID_ARG;
  SVO_TRY(svoslam::ensure_device());
  *scene = new svoslam_scene();
  return SVOSLAM_OK;
}
int svoslam_scene_destroy(svoslam_scene *scene) { delete scene; return SVOSLAM_OK; }
int svoslam_scene_load_obj(svoslam_scene *scene, const char *path) { return scene ? scene->impl.loadObjFile(path) : SVOSLAM_ERR_INVALID_ARG; }
int svoslam_scene_load_bmp(svoslam_scene *scene, const char *path) { return scene ? scene->impl.loadBMP(path) : SVOSLAM_ERR_INVALID_ARG; }
int svoslam_scene_set_octree(svoslam_scene *scene, float resolution, const float center[3], float size, int32_t depth_override) {
  return scene && center ? scene->impl.setOctree(resolution, center, size, depth_override) : SVOSLAM_ERR_INVALID_ARG;
}
int svoslam_scene_voxelize_meshes(svoslam_scene *scene, int32_t octree, int32_t log_N, void *stream) {
  return scene ? scene->impl.voxelizeMeshes(octree != 0, log_N, STREAM(stream)) : SVOSLAM_ERR_INVALID_ARG;
}
int svoslam_scene_extract_voxel_grid(svoslam_scene *scene, void *stream) {
  return scene ? scene->impl.extractVoxelGridFromOctree(STREAM(stream)) : SVOSLAM_ERR_INVALID_ARG;
}
int svoslam_scene_add_point_cloud(svoslam_scene *scene, const float origin[3], const float *d_points, const uint8_t *d_colors,
                                  int32_t n, const float bbox0[3], const float bbox1[3], void *stream) {
  if (!scene || !bbox0 || !bbox1) return SVOSLAM_ERR_INVALID_ARG;
  return scene->impl.addPointCloudToOctree(origin, d_points, d_colors, n, bbox0, bbox1, STREAM(stream));
}
int svoslam_scene_voxel_grid(svoslam_scene *scene, const float **d_centers, const float **d_colors, int32_t *n, float *scale) {
  if (!scene) return SVOSLAM_ERR_INVALID_ARG;
  const auto &g = scene->impl.voxel_grid();
  if (d_centers) *d_centers = g.d_centers;
  if (d_colors) *d_colors = g.d_colors;
  if (n) *n = g.size;
  if (scale) *scale = g.scale;
  return SVOSLAM_OK;
}
int svoslam_scene_svo(svoslam_scene *scene, const uint32_t **d_data, float center[3], float *size, int32_t *num_nodes,
                      int32_t *max_depth) {
  if (!scene || !scene->impl.tree()) return SVOSLAM_ERR_INVALID_ARG;
  const auto *t = scene->impl.tree();  // Octree::extractSVO, octree.cpp:339-360 (node = root)
  if (d_data) *d_data = t->pool().d_data;
  if (center) for (int k = 0; k < 3; k++) center[k] = t->center()[k];
  if (size) *size = t->size() / (float)pow(2, 0);
  if (num_nodes) *num_nodes = t->pool().size;
  if (max_depth) *max_depth = t->lastDepth();
  return SVOSLAM_OK;
}

}  // extern "C"
