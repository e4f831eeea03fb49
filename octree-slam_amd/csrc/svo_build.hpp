// svo_build.hpp -- see svo_build.hip
#pragma once
#include "common.hpp"
#include "workspace.hpp"

namespace svoslam {
int pool_init(svoslam_pool *pool, int32_t capacity_nodes, hipStream_t stream);
int pool_reserve(svoslam_pool *pool, int32_t capacity_nodes, hipStream_t stream);
int pool_sync(svoslam_pool *pool, hipStream_t stream);
int pool_reset(svoslam_pool *pool, hipStream_t stream);
int pool_expand(svoslam_pool *pool, float center[3], float *edge, const float toward[3], hipStream_t stream);
void pool_tracker_destroy(svoslam_pool *pool);
int pool_save(svoslam_pool *pool, const char *path, const float center[3], float edge, int depth, hipStream_t stream);
int pool_set_nodes(svoslam_pool *pool, const uint32_t *h_words, int32_t num_nodes, hipStream_t stream);
int pool_copy(svoslam_pool *dst, svoslam_pool *src, hipStream_t stream);
// out-of-core paging of sub-trees (pool_paging.hip)
int pool_evict_subtree(svoslam_pool *pool, const uint8_t *path, int levels, const char *file, hipStream_t stream);
int pool_restore_subtree(svoslam_pool *pool, const char *file, hipStream_t stream);
int subtree_file_nodes(const char *file, uint32_t **h_words, int32_t *num_nodes);
int pool_load(svoslam_pool *pool, const char *path, float center[3], float *edge, int *depth, hipStream_t stream);
int svo_from_point_cloud_async(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int n, int depth,
                               svoslam_pool *pool, const float center[3], float edge, hipStream_t stream);
int svo_fuse_sort(svoslam_workspace *ws, const float *d_points, int n, int depth, const float center[3], float edge,
                  hipStream_t stream);
int svo_fuse_sort_frame(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int w, int h, float fx, float fy, int depth,
                        const float center[3], float edge, float *d_bbox7, hipStream_t stream);
int svo_fuse_plan(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, hipStream_t stream);
int pool_structure_begin(svoslam_pool *pool, hipStream_t stream);
int svo_fuse_sort_frame_band(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int w, int h, float fx, float fy, int depth,
                             const float center[3], float edge, int first_row, int rows, hipStream_t stream);
int svo_fuse_merge_sorted(const unsigned long long *const *d_keys, const uint32_t *const *d_idx, const int32_t *counts, int lists,
                          unsigned long long *d_keys_out, uint32_t *d_idx_out, hipStream_t stream);
int svo_fuse_adopt_sorted(svoslam_workspace *ws, const unsigned long long *d_keys, const uint32_t *d_idx, int n, int depth);
int svo_fuse_export_sorted(svoslam_workspace *ws, int n, unsigned long long *d_keys_out, uint32_t *d_idx_out, hipStream_t stream);
int svo_fuse_plan_structure(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, hipStream_t stream);
int svo_fuse_split_early(svoslam_workspace *ws, int n, int depth, svoslam_pool *pool, hipStream_t stream);
int svo_fuse_commit_to(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, int slot, bool keep_plan,
                       hipStream_t stream);
int svo_fuse_commit(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, hipStream_t stream);
int svo_fuse_commit_deferred(svoslam_workspace *ws, const uint8_t *d_colors, int n, int depth, svoslam_pool *pool, hipStream_t stream);
int svo_fuse_apply(svoslam_workspace *ws, svoslam_pool *pool, hipStream_t stream);
int svo_fuse_keyrange_commit(svoslam_workspace *ws, const unsigned long long *d_keys, const uint32_t *d_idx, const uint8_t *d_colors, int n,
                             int depth, svoslam_pool *pool, int rank, int world, uint32_t *d_delta, long long delta_bytes, hipStream_t stream);
int svo_fuse_keyrange_apply(svoslam_workspace *ws, const unsigned long long *d_keys, int n, int depth, svoslam_pool *pool,
                            const uint32_t *const *d_deltas, int world, hipStream_t stream);
int svo_fuse_keyrange_status(svoslam_workspace *ws, int *flags, hipStream_t stream);
int svo_fuse_keyrange_discard(svoslam_workspace *ws, svoslam_pool *pool);
int svo_from_point_cloud(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors, int n, int depth,
                         svoslam_pool *pool, const float center[3], float edge, svoslam_fuse_stats *stats,
                         hipStream_t stream);
int svo_from_voxel_grid(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int n, int depth,
                        svoslam_pool *pool, const float center[3], float edge, svoslam_fuse_stats *stats,
                        hipStream_t stream);
int extract_voxel_grid(svoslam_workspace *ws, const svoslam_pool *pool, int depth, const float center[3], float edge,
                       float **d_centers, float **d_colors, int32_t *n_out, hipStream_t stream);
}  // namespace svoslam
