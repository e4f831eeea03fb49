// abi.hip -- the extern "C" surface of libsvoslam_hip.so (include/svoslam.h).
// Thin: argument checks, device check, dispatch into the kernel translation units.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.hpp"
#include "cone_trace.hpp"
#include "icp.hpp"
#include "image_kernels.hpp"
#include "mesh.hpp"
#include "model_depth.hpp"
#include "frame_io.hpp"
#include "pool_grid.hpp"
#include "stage_timing.hpp"
#include "svo_build.hpp"
#include "workspace.hpp"

namespace svoslam {

static thread_local char g_last_error[512] = "";
static char g_arch[256] = "";
static int g_device_state = 0;  // 0 = unknown, 1 = ok, -1 = none

void set_last_error_text(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}
void set_last_error(const char *what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}

// Fail loudly when there is no gfx950 device: the library has no CPU path.
int ensure_device() {
  if (g_device_state == 1) return SVOSLAM_OK;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    snprintf(g_last_error, sizeof(g_last_error), "no HIP device (hipGetDeviceCount: %s); libsvoslam_hip has no CPU fallback",
             hipGetErrorString(e));
    g_device_state = -1;
    return SVOSLAM_ERR_NO_DEVICE;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) snprintf(g_arch, sizeof(g_arch), "%s", prop.gcnArchName);
  if (strncmp(g_arch, "gfx950", 6) != 0) {
    snprintf(g_last_error, sizeof(g_last_error), "device arch '%s' is not gfx950; this library is built for MI355X only", g_arch);
    g_device_state = -1;
    return SVOSLAM_ERR_NO_DEVICE;
  }
  g_device_state = 1;
  return SVOSLAM_OK;
}

static thread_local hipEvent_t g_t0 = nullptr, g_t1 = nullptr;

}  // namespace svoslam

using namespace svoslam;

#define S(stream) reinterpret_cast<hipStream_t>(stream)
#define NEED_DEVICE() SVO_TRY(ensure_device())

extern "C" {

int svoslam_abi_version(void) { return SVOSLAM_ABI_VERSION; }

const char *svoslam_status_string(int status) {
  switch (status) {
    case SVOSLAM_OK: return "ok";
    case SVOSLAM_ERR_INVALID_ARG: return "invalid argument";
    case SVOSLAM_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
    case SVOSLAM_ERR_HIP: return "HIP runtime error";
    case SVOSLAM_ERR_OOM: return "out of device memory";
    case SVOSLAM_ERR_DEPTH: return "max_depth outside [1,16]";
    case SVOSLAM_ERR_POOL_LIMIT: return "node pool would exceed 2^30 nodes";
    case SVOSLAM_ERR_TRACKING_LOST: return "camera tracking lost";
    case SVOSLAM_ERR_IO: return "file could not be opened, read or written";
    case SVOSLAM_ERR_FORMAT: return "file format error (magic, size, checksum or structure)";
    default: return "unknown status";
  }
}

const char *svoslam_last_error(void) { return g_last_error; }

const char *svoslam_device_arch(void) { return ensure_device() == SVOSLAM_OK ? g_arch : nullptr; }

int svoslam_pool_init(svoslam_pool *pool, int32_t capacity_nodes, void *stream) {
  NEED_DEVICE();
  return pool_init(pool, capacity_nodes, S(stream));
}
int svoslam_pool_reserve(svoslam_pool *pool, int32_t capacity_nodes, void *stream) {
  NEED_DEVICE();
  return pool_reserve(pool, capacity_nodes, S(stream));
}
int svoslam_pool_free(svoslam_pool *pool) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  pool_accel_unregister(pool);
  if (pool->d_data) SVO_HIP(hipFree(pool->d_data));
  if (pool->d_size) SVO_HIP(hipFree(pool->d_size));
  pool_tracker_destroy(pool);
  pool->d_data = nullptr; pool->size = 0; pool->capacity = 0;
  pool->d_size = nullptr; pool->pending = 0; pool->pending_bound = 0;
  return SVOSLAM_OK;
}
int svoslam_pool_reset(svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return pool_reset(pool, S(stream));
}
int svoslam_camera_reset(svoslam_camera *cam) {
  NEED_DEVICE();
  return camera_reset(cam);
}
int svoslam_pool_expand(svoslam_pool *pool, float center[3], float *edge_length, const float toward[3], void *stream) {
  NEED_DEVICE();
  return pool_expand(pool, center, edge_length, toward, S(stream));
}
int svoslam_pool_touch(svoslam_pool *pool) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  pool_accel_invalidate(pool);
  return SVOSLAM_OK;
}
int svoslam_pool_march_accel(const svoslam_pool *pool, int32_t *has_grid, int32_t *brick_state, int32_t *brick_shift) {
  if (!pool) return SVOSLAM_ERR_INVALID_ARG;
  const std::shared_ptr<PoolAccel> pa = pool_accel_find(pool->d_data);
  if (has_grid) *has_grid = pa && pa->grid.ptr ? 1 : 0;
  if (brick_state) *brick_state = !pa ? 0 : (pa->bricks_failed ? -1 : (pa->bricks ? 1 : 0));
  if (brick_shift) *brick_shift = pa && pa->bricks ? pa->brick_shift : -1;
  return SVOSLAM_OK;
}
int svoslam_pool_sync(svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return pool_sync(pool, S(stream));
}
int svoslam_pool_save(svoslam_pool *pool, const char *path, const float center[3], float edge_length, int32_t max_depth,
                      void *stream) {
  NEED_DEVICE();
  return pool_save(pool, path, center, edge_length, max_depth, S(stream));
}
int svoslam_pool_set_nodes(svoslam_pool *pool, const uint32_t *h_words, int32_t num_nodes, void *stream) {
  NEED_DEVICE();
  return pool_set_nodes(pool, h_words, num_nodes, S(stream));
}
int svoslam_pool_evict_subtree(svoslam_pool *pool, const uint8_t *path, int32_t levels, const char *file, void *stream) {
  NEED_DEVICE();
  return pool_evict_subtree(pool, path, levels, file, S(stream));
}
int svoslam_pool_restore_subtree(svoslam_pool *pool, const char *file, void *stream) {
  NEED_DEVICE();
  return pool_restore_subtree(pool, file, S(stream));
}
int svoslam_subtree_file_nodes(const char *file, uint32_t **h_words, int32_t *num_nodes) { return subtree_file_nodes(file, h_words, num_nodes); }
int svoslam_pool_copy(svoslam_pool *dst, svoslam_pool *src, void *stream) {
  NEED_DEVICE();
  return pool_copy(dst, src, S(stream));
}
int svoslam_pool_load(svoslam_pool *pool, const char *path, float center[3], float *edge_length, int32_t *max_depth,
                      void *stream) {
  NEED_DEVICE();
  int d = 0;
  const int rc = pool_load(pool, path, center, edge_length, &d, S(stream));
  if (max_depth) *max_depth = d;
  return rc;
}
int svoslam_svo_from_point_cloud_async(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int32_t n,
                                       int32_t max_depth, svoslam_pool *pool, const float center[3], float edge_length,
                                       void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return svo_from_point_cloud_async(ws, d_points, d_colors, n, max_depth, pool, center, edge_length, S(stream));
}

int svoslam_svo_fuse_sort(svoslam_workspace *ws, const float *d_points, int32_t n, int32_t max_depth, const float center[3],
                          float edge_length, void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return svo_fuse_sort(ws, d_points, n, max_depth, center, edge_length, S(stream));
}
int svoslam_svo_fuse_sort_frame(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int32_t width, int32_t height,
                                float fx, float fy, int32_t max_depth, const float center[3], float edge_length, float *d_bbox7,
                                void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return svo_fuse_sort_frame(ws, d_depth, d_pose, width, height, fx, fy, max_depth, center, edge_length, d_bbox7, S(stream));
}
int svoslam_svo_fuse_plan(svoslam_workspace *ws, int32_t n, int32_t max_depth, svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return svo_fuse_plan(ws, n, max_depth, pool, S(stream));
}
int svoslam_pool_structure_begin(svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return pool_structure_begin(pool, S(stream));
}
int svoslam_svo_fuse_sort_frame_band(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int32_t width, int32_t height,
                                     float fx, float fy, int32_t max_depth, const float center[3], float edge_length, int32_t first_row,
                                     int32_t rows, void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return svo_fuse_sort_frame_band(ws, d_depth, d_pose, width, height, fx, fy, max_depth, center, edge_length, first_row, rows, S(stream));
}
int svoslam_svo_fuse_merge_sorted(const unsigned long long *const *d_keys, const uint32_t *const *d_idx, const int32_t *counts, int32_t lists,
                                  unsigned long long *d_keys_out, uint32_t *d_idx_out, void *stream) {
  NEED_DEVICE();
  return svo_fuse_merge_sorted(d_keys, d_idx, counts, lists, d_keys_out, d_idx_out, S(stream));
}
int svoslam_svo_fuse_adopt_sorted(svoslam_workspace *ws, const unsigned long long *d_keys, const uint32_t *d_idx, int32_t n, int32_t max_depth) {
  NEED_DEVICE();
  return svo_fuse_adopt_sorted(ws, d_keys, d_idx, n, max_depth);
}
int svoslam_svo_fuse_export_sorted(svoslam_workspace *ws, int32_t n, unsigned long long *d_keys_out, uint32_t *d_idx_out, void *stream) {
  NEED_DEVICE();
  return svo_fuse_export_sorted(ws, n, d_keys_out, d_idx_out, S(stream));
}
int svoslam_svo_fuse_plan_structure(svoslam_workspace *ws, int32_t n, int32_t max_depth, svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return svo_fuse_plan_structure(ws, n, max_depth, pool, S(stream));
}
int svoslam_svo_fuse_split_early(svoslam_workspace *ws, int32_t n, int32_t max_depth, svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return svo_fuse_split_early(ws, n, max_depth, pool, S(stream));
}
int svoslam_svo_fuse_commit(svoslam_workspace *ws, const uint8_t *d_colors, int32_t n, int32_t max_depth, svoslam_pool *pool,
                            void *stream) {
  NEED_DEVICE();
  return svo_fuse_commit(ws, d_colors, n, max_depth, pool, S(stream));
}
int svoslam_svo_fuse_commit_to(svoslam_workspace *ws, const uint8_t *d_colors, int32_t n, int32_t max_depth, svoslam_pool *pool,
                               int32_t slot, int32_t keep_plan, void *stream) {
  NEED_DEVICE();
  return svo_fuse_commit_to(ws, d_colors, n, max_depth, pool, slot, keep_plan != 0, S(stream));
}
int svoslam_svo_fuse_commit_deferred(svoslam_workspace *ws, const uint8_t *d_colors, int32_t n, int32_t max_depth, svoslam_pool *pool,
                                     void *stream) {
  NEED_DEVICE();
  return svo_fuse_commit_deferred(ws, d_colors, n, max_depth, pool, S(stream));
}
int svoslam_svo_fuse_apply(svoslam_workspace *ws, svoslam_pool *pool, void *stream) {
  NEED_DEVICE();
  return svo_fuse_apply(ws, pool, S(stream));
}
int svoslam_svo_fuse_keyrange_commit(svoslam_workspace *ws, const unsigned long long *d_sorted_keys, const uint32_t *d_sorted_idx,
                                     const uint8_t *d_colors, int32_t n, int32_t max_depth, svoslam_pool *pool, int32_t rank, int32_t world,
                                     uint32_t *d_delta, int64_t delta_bytes, void *stream) {
  NEED_DEVICE();
  return svo_fuse_keyrange_commit(ws, d_sorted_keys, d_sorted_idx, d_colors, n, max_depth, pool, rank, world, d_delta, (long long)delta_bytes, S(stream));
}
int svoslam_svo_fuse_keyrange_apply(svoslam_workspace *ws, const unsigned long long *d_sorted_keys, int32_t n, int32_t max_depth, svoslam_pool *pool,
                                    const uint32_t *const *d_deltas, int32_t world, void *stream) {
  NEED_DEVICE();
  return svo_fuse_keyrange_apply(ws, d_sorted_keys, n, max_depth, pool, d_deltas, world, S(stream));
}
int svoslam_svo_fuse_keyrange_discard(svoslam_workspace *ws, svoslam_pool *pool) {
  NEED_DEVICE();
  return svo_fuse_keyrange_discard(ws, pool);
}
int svoslam_svo_fuse_keyrange_status(svoslam_workspace *ws, int32_t *flags, void *stream) {
  NEED_DEVICE();
  int f = 0;
  const int rc = svo_fuse_keyrange_status(ws, &f, S(stream));
  if (flags) *flags = f;
  return rc;
}

int svoslam_frame_reader_open(svoslam_frame_reader **reader, const char *association_file, float depth_units_per_metre) {
  return frame_reader_open(reader, association_file, depth_units_per_metre);
}
int svoslam_frame_reader_close(svoslam_frame_reader *reader) { return frame_reader_close(reader); }
int svoslam_frame_reader_info(const svoslam_frame_reader *reader, int32_t *width, int32_t *height, int32_t *num_frames) {
  int w = 0, h = 0, n = 0;
  const int rc = frame_reader_info(reader, &w, &h, &n);
  if (width) *width = w;
  if (height) *height = h;
  if (num_frames) *num_frames = n;
  return rc;
}
int svoslam_frame_reader_rewind(svoslam_frame_reader *reader) { return frame_reader_rewind(reader); }
int svoslam_frame_reader_next_host(svoslam_frame_reader *reader, uint16_t *h_depth, uint8_t *h_color, long long *timestamp,
                                   int32_t *got) {
  int g = 0;
  const int rc = frame_reader_next_host(reader, h_depth, h_color, timestamp, &g);
  if (got) *got = g;
  return rc;
}
int svoslam_frame_reader_next(svoslam_frame_reader *reader, uint16_t *d_depth, uint8_t *d_color, long long *timestamp,
                              int32_t *got, void *stream) {
  NEED_DEVICE();
  int g = 0;
  const int rc = frame_reader_next(reader, d_depth, d_color, timestamp, &g, S(stream));
  if (got) *got = g;
  return rc;
}
int svoslam_focal_from_fov(int32_t width, int32_t height, float hfov_rad, float vfov_rad, float *fx, float *fy) {
  return focal_from_fov(width, height, hfov_rad, vfov_rad, fx, fy);
}
int svoslam_image_load(const char *path, void **h_data, int32_t *width, int32_t *height, int32_t *channels, int32_t *bits) {
  if (!h_data) return SVOSLAM_ERR_INVALID_ARG;
  HostImage img;
  const int rc = image_load(path, img);
  if (rc != SVOSLAM_OK) return rc;
  void *p = malloc(img.data.size() ? img.data.size() : 1);
  if (!p) return SVOSLAM_ERR_OOM;
  memcpy(p, img.data.data(), img.data.size());
  *h_data = p;
  if (width) *width = img.width;
  if (height) *height = img.height;
  if (channels) *channels = img.channels;
  if (bits) *bits = img.bits;
  return SVOSLAM_OK;
}

int svoslam_workspace_create(svoslam_workspace **ws) {
  if (!ws) return SVOSLAM_ERR_INVALID_ARG;
  NEED_DEVICE();
  *ws = new svoslam_workspace();
  return SVOSLAM_OK;
}
int svoslam_workspace_destroy(svoslam_workspace *ws) {
  if (!ws) return SVOSLAM_OK;
  ws->release_all();
  delete ws;
  return SVOSLAM_OK;
}

int svoslam_svo_from_point_cloud(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int32_t n,
                                 int32_t max_depth, svoslam_pool *pool, const float center[3], float edge_length,
                                 svoslam_fuse_stats *stats, void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return svo_from_point_cloud(ws, d_points, d_colors, n, max_depth, pool, center, edge_length, stats, S(stream));
}

int svoslam_svo_from_voxel_grid(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int32_t n,
                                int32_t max_depth, svoslam_pool *pool, const float center[3], float edge_length,
                                svoslam_fuse_stats *stats, void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return svo_from_voxel_grid(ws, d_centers, d_colors, n, max_depth, pool, center, edge_length, stats, S(stream));
}

int svoslam_extract_voxel_grid(svoslam_workspace *ws, const svoslam_pool *pool, int32_t max_depth, const float center[3],
                               float edge_length, float **d_centers, float **d_colors, int32_t *n_out, void *stream) {
  NEED_DEVICE();
  if (!center) return SVOSLAM_ERR_INVALID_ARG;
  return extract_voxel_grid(ws, pool, max_depth, center, edge_length, d_centers, d_colors, n_out, S(stream));
}

int svoslam_free(void *d_ptr) {
  if (d_ptr) SVO_HIP(hipFree(d_ptr));
  return SVOSLAM_OK;
}

int svoslam_mesh_load_obj(const char *path, svoslam_mesh *out) { return mesh_load_obj(path, out); }
int svoslam_mesh_free(svoslam_mesh *mesh) { return mesh_free(mesh); }
int svoslam_texture_load_bmp(const char *path, svoslam_texture *out) { return texture_load_bmp(path, out); }
int svoslam_texture_free(svoslam_texture *tex) { return texture_free(tex); }
int svoslam_mesh_to_voxel_grid(svoslam_workspace *ws, const svoslam_mesh *mesh, const svoslam_texture *tex, int32_t log_N,
                               int32_t log_T, float **d_centers, float **d_colors, unsigned long long **d_indices,
                               int32_t *n_out, float *scale_out, void *stream) {
  NEED_DEVICE();
  return mesh_to_voxel_grid(ws, mesh, tex, log_N, log_T, d_centers, d_colors, d_indices, n_out, scale_out, S(stream));
}

int svoslam_mesh_last_fragments(const svoslam_workspace *ws, int64_t *fragments_out) {
  if (!ws || !fragments_out) return SVOSLAM_ERR_INVALID_ARG;
  *fragments_out = ws->mesh_fragments;
  return SVOSLAM_OK;
}

int svoslam_voxel_grid_to_mesh(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int32_t n, float scale_factor,
                               const float *cube_vbo, int32_t cube_vbosize, const int32_t *cube_ibo, int32_t cube_ibosize,
                               const float *cube_nbo, float *d_vbo, int32_t *d_ibo, float *d_nbo, float *d_cbo, void *stream) {
  NEED_DEVICE();
  return voxel_grid_to_mesh(ws, d_centers, d_colors, n, scale_factor, cube_vbo, cube_vbosize, cube_ibo, cube_ibosize, cube_nbo, d_vbo,
                            d_ibo, d_nbo, d_cbo, S(stream));
}

int svoslam_malloc(void **d_ptr, size_t bytes) {
  NEED_DEVICE();
  if (!d_ptr) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipMalloc(d_ptr, bytes));
  return SVOSLAM_OK;
}

int svoslam_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes) {
  NEED_DEVICE();
  if (bytes == 0) return SVOSLAM_OK;
  if (!d_dst || !h_src) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return SVOSLAM_OK;
}

int svoslam_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes) {
  NEED_DEVICE();
  if (bytes == 0) return SVOSLAM_OK;
  if (!h_dst || !d_src) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return SVOSLAM_OK;
}

int svoslam_cone_trace_svo(uint8_t *d_pos, int32_t width, int32_t height, float fov, const float view[16],
                           const uint32_t *d_octree, const float center[3], float size, int32_t mode,
                           unsigned long long *d_steps, void *stream) {
  NEED_DEVICE();
  return cone_trace_svo(d_pos, width, height, 0, height, fov, view, d_octree, center, size, mode, d_steps, S(stream));
}

int svoslam_cone_trace_svo_band(uint8_t *d_pos, int32_t width, int32_t height, int32_t row_first, int32_t rows, float fov,
                                const float view[16], const uint32_t *d_octree, const float center[3], float size,
                                int32_t mode, unsigned long long *d_steps, void *stream) {
  NEED_DEVICE();
  return cone_trace_svo(d_pos, width, height, row_first, rows, fov, view, d_octree, center, size, mode, d_steps, S(stream));
}

int svoslam_cone_trace_release(void *stream, int32_t all_streams) { return cone_trace_release(S(stream), all_streams != 0); }
int svoslam_cone_trace_timing(int32_t enable) {
  const unsigned bit = 1u << kStageMarch, m = stage_timing_mask();
  return stage_timing(enable ? (m | bit) : (m & ~bit));
}
int svoslam_cone_trace_timing_read(float *h_ms_sum, int32_t *h_launches) {
  int n = 0;
  const int rc = stage_timing_read(kStageMarch, h_ms_sum, &n);
  if (h_launches) *h_launches = n;
  return rc;
}
int svoslam_stage_timing(uint32_t mask) { return stage_timing(mask); }
int svoslam_stage_timing_read(int32_t stage, float *h_ms_sum, int32_t *h_pairs) {
  int n = 0;
  const int rc = stage_timing_read(stage, h_ms_sum, &n);
  if (h_pairs) *h_pairs = n;
  return rc;
}

int svoslam_generate_vertex_map(const uint16_t *d_depth, float *d_vertex, int32_t width, int32_t height, float fx, float fy,
                                int32_t img_w, int32_t img_h, void *stream) {
  NEED_DEVICE();
  return generate_vertex_map(d_depth, d_vertex, width, height, fx, fy, img_w, img_h, S(stream));
}
int svoslam_generate_vertex_map_rows(const uint16_t *d_depth, float *d_vertex, int32_t width, int32_t height, int32_t first_row,
                                     int32_t rows, float fx, float fy, int32_t img_w, int32_t img_h, void *stream) {
  NEED_DEVICE();
  return generate_vertex_map_rows(d_depth, d_vertex, width, height, first_row, rows, fx, fy, img_w, img_h, S(stream));
}
int svoslam_generate_normal_map(const float *d_vertex, float *d_normal, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return generate_normal_map(d_vertex, d_normal, width, height, S(stream));
}
int svoslam_bilateral_filter(const uint16_t *d_in, uint16_t *d_out, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return bilateral_filter(d_in, d_out, width, height, S(stream));
}
int svoslam_subsample_depth_u16(uint16_t *d_data, uint16_t *d_tmp, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return subsample_depth_u16(d_data, d_tmp, width, height, S(stream));
}
int svoslam_subsample_depth_f32(float *d_data, float *d_tmp, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return subsample_depth_f32(d_data, d_tmp, width, height, S(stream));
}
int svoslam_subsample_f32(float *d_data, float *d_tmp, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return subsample_f32(d_data, d_tmp, width, height, S(stream));
}
int svoslam_subsample_rgb8(uint8_t *d_data, uint8_t *d_tmp, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return subsample_rgb8(d_data, d_tmp, width, height, S(stream));
}
int svoslam_color_to_intensity(const uint8_t *d_rgb, float *d_out, int32_t n, void *stream) {
  NEED_DEVICE();
  return color_to_intensity(d_rgb, d_out, n, S(stream));
}
int svoslam_gradient(const float *d_intensity, float *d_gradient, int32_t width, int32_t height, void *stream) {
  NEED_DEVICE();
  return gradient(d_intensity, d_gradient, width, height, S(stream));
}
int svoslam_difference(const float *d_in1, const float *d_in2, float *d_out, int32_t n, void *stream) {
  NEED_DEVICE();
  return difference(d_in1, d_in2, d_out, n, S(stream));
}
int svoslam_transform_vertex_map(float *d_vertex, const float trans[16], int32_t n, void *stream) {
  NEED_DEVICE();
  return transform_vertex_map(d_vertex, trans, n, S(stream));
}
int svoslam_transform_normal_map(float *d_normal, const float trans[16], int32_t n, void *stream) {
  NEED_DEVICE();
  return transform_normal_map(d_normal, trans, n, S(stream));
}
int svoslam_transform_vertex_map_dmat(float *d_vertex, const float *d_trans, int32_t n, void *stream) {
  NEED_DEVICE();
  return transform_vertex_map_dmat(d_vertex, d_trans, n, S(stream));
}

static svoslam::DeviceBuffer g_misc;   // bbox partials / icp_cost2 accumulators for the stateless entry points
static svoslam::DeviceBuffer g_misc2;  // per-workgroup rows of svoslam_icp_accumulate

int svoslam_point_cloud_bbox(const float *d_points, int32_t n, float h_bbox0[3], float h_bbox1[3], void *stream) {
  NEED_DEVICE();
  return point_cloud_bbox(g_misc, d_points, n, h_bbox0, h_bbox1, S(stream));
}

int svoslam_point_cloud_bbox_device(svoslam_workspace *ws, const float *d_points, int32_t n, float *d_out7, void *stream) {
  NEED_DEVICE();
  if (!ws) return SVOSLAM_ERR_INVALID_ARG;
  return point_cloud_bbox_device(ws->misc, d_points, n, d_out7, S(stream));
}

int svoslam_icp_cost2(const float *d_last_vertex, const float *d_last_normal, const float *d_cur_vertex,
                      const float *d_cur_normal, int32_t width, int32_t height, float h_A[36], float h_b[6], void *stream) {
  NEED_DEVICE();
  return icp_cost2(g_misc, d_last_vertex, d_last_normal, d_cur_vertex, d_cur_normal, width, height, h_A, h_b, S(stream));
}

int svoslam_rgbd_cost(const float *d_last_intensity, const float *d_last_gradient, const float *d_last_vertex,
                      const float *d_cur_intensity, const float *d_cur_vertex, int32_t width, int32_t height, float fx, float fy,
                      int32_t img_width, int32_t img_height, float h_A[36], float h_b[6], void *stream) {
  NEED_DEVICE();
  return rgbd_cost(g_misc, d_last_intensity, d_last_gradient, d_last_vertex, d_cur_intensity, d_cur_vertex, width, height, fx, fy,
                   img_width, img_height, h_A, h_b, S(stream));
}
int svoslam_camera_set_strict_reference(svoslam_camera *cam, int32_t strict) { return camera_set_strict_reference(cam, strict); }
int svoslam_camera_set_rgbd(svoslam_camera *cam, int32_t enable) {
  NEED_DEVICE();
  return camera_set_rgbd(cam, enable);
}

int svoslam_raycast_model_depth(uint16_t *d_depth, int32_t width, int32_t height, float fx, float fy, const float *cam_to_world,
                                const float *d_cam_to_world, const uint32_t *d_octree, const float center[3], float size,
                                unsigned long long *d_steps, void *stream) {
  NEED_DEVICE();
  return raycast_model_depth(d_depth, width, height, fx, fy, cam_to_world, d_cam_to_world, d_octree, center, size, d_steps, S(stream));
}
int svoslam_camera_set_model_depth(svoslam_camera *cam, const uint16_t *d_depth, void *stream) {
  NEED_DEVICE();
  return camera_set_model_depth(cam, d_depth, S(stream));
}
int svoslam_camera_set_frame_to_model(svoslam_camera *cam, int32_t enable) {
  NEED_DEVICE();
  return camera_set_frame_to_model(cam, enable);
}

int svoslam_icp_cost(const float *d_last_vertex, const float *d_last_normal, const float *d_cur_vertex,
                     const float *d_cur_normal, int32_t width, int32_t height, float h_A[36], float h_b[6],
                     int32_t *num_correspondences, void *stream) {
  NEED_DEVICE();
  int m = 0;
  const int rc = icp_cost(g_misc, d_last_vertex, d_last_normal, d_cur_vertex, d_cur_normal, width, height, h_A, h_b, &m, S(stream));
  if (num_correspondences) *num_correspondences = m;
  return rc;
}

int svoslam_icp_accumulate(const float *d_last_vertex, const float *d_last_normal, const float *d_cur_vertex,
                           const float *d_cur_normal, int32_t width, int32_t height, int32_t first_pixel,
                           int32_t num_pixels, double *d_acc, void *stream) {
  NEED_DEVICE();
  return icp_accumulate(g_misc2, d_last_vertex, d_last_normal, d_cur_vertex, d_cur_normal, width, height, first_pixel,
                        num_pixels, d_acc, S(stream));
}

int svoslam_camera_create(svoslam_camera **cam, int32_t width, int32_t height, float fx, float fy) {
  return camera_create(cam, width, height, fx, fy);
}
int svoslam_camera_destroy(svoslam_camera *cam) { return camera_destroy(cam); }
int svoslam_camera_set_band(svoslam_camera *cam, int32_t first_row, int32_t rows) { return camera_set_band(cam, first_row, rows); }
int svoslam_camera_set_acc(svoslam_camera *cam, double *d_acc) { return camera_set_acc(cam, d_acc); }
int svoslam_camera_update(svoslam_camera *cam, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp,
                          int32_t *processed, void *stream) {
  return camera_update(cam, d_depth, d_rgb, timestamp, processed, S(stream));
}
int svoslam_camera_prepare(svoslam_camera *cam, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp,
                           int32_t *processed, void *stream) {
  return camera_prepare(cam, d_depth, d_rgb, timestamp, processed, S(stream));
}
int svoslam_camera_track(svoslam_camera *cam, void *stream) { return camera_track(cam, S(stream)); }
int svoslam_camera_pair_delta(svoslam_camera *cam, const uint16_t *d_depth_prev, const uint8_t *d_rgb_prev, const uint16_t *d_depth_cur,
                              const uint8_t *d_rgb_cur, float *d_delta, void *stream) {
  return camera_pair_delta(cam, d_depth_prev, d_rgb_prev, d_depth_cur, d_rgb_cur, d_delta, S(stream));
}
int svoslam_camera_apply_delta(svoslam_camera *cam, const float *d_delta, long long timestamp, int32_t *processed, void *stream) {
  return camera_apply_delta(cam, d_delta, timestamp, processed, S(stream));
}
int svoslam_camera_begin(svoslam_camera *cam, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp,
                         int32_t *processed, void *stream) {
  return camera_begin(cam, d_depth, d_rgb, timestamp, processed, S(stream));
}
int svoslam_camera_icp_iters(int32_t level) { return camera_icp_iters(level); }
int svoslam_camera_icp_accumulate(svoslam_camera *cam, int32_t level, int32_t iter, void *stream) {
  return camera_icp_accumulate(cam, level, iter, S(stream));
}
double *svoslam_camera_acc(svoslam_camera *cam) { return camera_acc(cam); }
int svoslam_camera_icp_solve(svoslam_camera *cam, int32_t level, int32_t iter, void *stream) {
  return camera_icp_solve(cam, level, iter, S(stream));
}
int svoslam_camera_end(svoslam_camera *cam, void *stream) { return camera_end(cam, S(stream)); }
int svoslam_camera_pose(svoslam_camera *cam, float h_position[3], float h_orientation[9], void *stream) {
  return camera_pose(cam, h_position, h_orientation, S(stream));
}
const float *svoslam_camera_fusion_transform_device(svoslam_camera *cam) { return camera_fusion_transform_device(cam); }
int svoslam_camera_last_system(svoslam_camera *cam, float h_A[36], float h_b[6], float h_x[6], void *stream) {
  return camera_last_system(cam, h_A, h_b, h_x, S(stream));
}
const float *svoslam_camera_last_vertex(svoslam_camera *cam, int32_t level) { return camera_last_vertex(cam, level); }
const float *svoslam_camera_last_normal(svoslam_camera *cam, int32_t level) { return camera_last_normal(cam, level); }
int svoslam_camera_tracking_lost_count(svoslam_camera *cam, int32_t *count, void *stream) {
  return camera_tracking_lost_count(cam, count, S(stream));
}

int svoslam_camera_latest_timestamp(svoslam_camera *cam, int32_t *have, long long *timestamp) {
  return camera_latest_timestamp(cam, have, timestamp);
}
int svoslam_camera_track_profile(svoslam_camera *cam, unsigned long long *h_stamps, void *stream) {
  return camera_track_profile(cam, h_stamps, S(stream));
}

// startTiming / stopTiming (src/timing_utils.cu:11-32) on hipEvents
int svoslam_timer_start(void *stream) {
  NEED_DEVICE();
  if (!g_t0) { SVO_HIP(hipEventCreate(&g_t0)); SVO_HIP(hipEventCreate(&g_t1)); }
  SVO_HIP(hipEventRecord(g_t0, S(stream)));
  return SVOSLAM_OK;
}
int svoslam_timer_stop(void *stream, float *h_ms) {
  NEED_DEVICE();
  if (!g_t0 || !h_ms) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipEventRecord(g_t1, S(stream)));
  SVO_HIP(hipEventSynchronize(g_t1));
  SVO_HIP(hipEventElapsedTime(h_ms, g_t0, g_t1));
  return SVOSLAM_OK;
}

}  // extern "C"
