// host_scene.hpp -- host mirror of world::Octree / world::Scene (see host_scene.hip)
#pragma once
#include <vector>

#include "common.hpp"
#include "workspace.hpp"

namespace octree_slam {
namespace world {

// VoxelGrid of common_types.h:55-63 with device arrays
struct DeviceVoxelGrid {
  float *d_centers, *d_colors;  // size x vec4
  int32_t size;
  float scale;
  float bbox0[3], bbox1[3];
};

class Octree {  // include/octree_slam/world/octree.h:80-125
 public:
  Octree(const float resolution, const float center[3], const float size);
  ~Octree();
  int addCloud(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int size, hipStream_t s);
  int addVoxelGrid(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int n, hipStream_t s);
  int extractVoxelGrid(svoslam_workspace *ws, float grid_scale, float **d_centers, float **d_colors, int32_t *n, hipStream_t s);
  void boundingBox(float bbox0[3], float bbox1[3]) const;
  void expandBySize(const float add_size);
  void setDepthOverride(int d) { depth_override_ = d; }
  const svoslam_pool &pool() const { return pool_; }
  const float *center() const { return center_; }
  float size() const { return size_; }
  int lastDepth() const { return last_depth_; }

 private:
  int maxDepth(float edge_length, float resolution) const;
  svoslam_pool pool_;      // root_->gpu_data_ / gpu_size_
  float center_[3];
  float size_;             // half edge length of the root
  float resolution_;
  int depth_override_ = 0;
  int last_depth_ = 0;
};

class Scene {  // include/octree_slam/world/scene.h:20-81
 public:
  Scene();
  ~Scene();
  int loadObjFile(const char *filename);
  int loadBMP(const char *filename);
  int voxelizeMeshes(const bool octree, int log_N, hipStream_t s);
  int extractVoxelGridFromOctree(hipStream_t s);
  int addPointCloudToOctree(const float *origin, const float *d_points, const uint8_t *d_colors, const int size,
                            const float bbox0[3], const float bbox1[3], hipStream_t s);
  int setOctree(float resolution, const float center[3], float size, int depth_override);
  const DeviceVoxelGrid &voxel_grid() const { return grid_; }
  const Octree *tree() const { return tree_; }

 private:
  void freeGrid();
  std::vector<svoslam_mesh> meshes_;
  std::vector<svoslam_texture> textures_;
  DeviceVoxelGrid grid_;
  Octree *tree_;
  svoslam_workspace *ws_ = nullptr;
  int depth_override_ = 0;
};

}  // namespace world
}  // namespace octree_slam
