// icp.hpp -- see icp.hip
#pragma once
#include "common.hpp"
#include "workspace.hpp"

struct svoslam_camera;

namespace svoslam {
int icp_accumulate(svoslam::DeviceBuffer &scratch, const float *lv, const float *ln, const float *cv, const float *cn,
                   int w, int h, int first, int num, double *d_acc, hipStream_t s);
int icp_cost2(svoslam::DeviceBuffer &scratch, const float *lv, const float *ln, const float *cv, const float *cn, int w,
              int h, float A[36], float b[6], hipStream_t s);
int icp_cost(svoslam::DeviceBuffer &scratch, const float *lv, const float *ln, const float *cv, const float *cn, int w, int h,
             float A[36], float b[6], int *num_corr, hipStream_t s);
int camera_icp_iters(int level);
int camera_create(svoslam_camera **out, int w, int h, float fx, float fy);
int camera_destroy(svoslam_camera *c);
int camera_reset(svoslam_camera *c);
int camera_begin(svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp, int32_t *processed, hipStream_t s);
int camera_prepare(svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp, int32_t *processed, hipStream_t s);
int camera_track(svoslam_camera *c, hipStream_t s);
int camera_icp_accumulate(svoslam_camera *c, int level, int iter, hipStream_t s);
int camera_icp_solve(svoslam_camera *c, int level, int iter, hipStream_t s);
int camera_end(svoslam_camera *c, hipStream_t s);
int camera_update(svoslam_camera *c, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp, int32_t *processed, hipStream_t s);
int camera_pair_delta(svoslam_camera *c, const uint16_t *d_depth_prev, const uint8_t *d_rgb_prev, const uint16_t *d_depth_cur,
                      const uint8_t *d_rgb_cur, float *d_delta, hipStream_t s);
int camera_apply_delta(svoslam_camera *c, const float *d_delta, long long timestamp, int32_t *processed, hipStream_t s);
int camera_set_rgbd(svoslam_camera *c, int enable);
int camera_set_strict_reference(svoslam_camera *c, int strict);
int camera_set_model_depth(svoslam_camera *c, const uint16_t *d_depth, hipStream_t s);
int camera_set_frame_to_model(svoslam_camera *c, int enable);
int rgbd_cost(svoslam::DeviceBuffer &scratch, const float *last_i, const float *last_g, const float *last_v, const float *cur_i,
              const float *cur_v, int w, int h, float fx, float fy, int img_w, int img_h, float A[36], float b[6], hipStream_t s);
int camera_set_band(svoslam_camera *c, int first_row, int rows);
int camera_set_acc(svoslam_camera *c, double *d_acc);
double *camera_acc(svoslam_camera *c);
int camera_pose(svoslam_camera *c, float pos[3], float ori[9], hipStream_t s);
int camera_last_system(svoslam_camera *c, float A[36], float b[6], float x[6], hipStream_t s);
const float *camera_fusion_transform_device(svoslam_camera *c);
const float *camera_last_vertex(svoslam_camera *c, int level);
const float *camera_last_normal(svoslam_camera *c, int level);
int camera_tracking_lost_count(svoslam_camera *c, int *count, hipStream_t s);
int camera_latest_timestamp(svoslam_camera *c, int32_t *have, long long *timestamp);
int camera_track_profile(svoslam_camera *c, unsigned long long *h_stamps, hipStream_t s);
}  // namespace svoslam
