// image_kernels.hpp -- see image_kernels.hip
#pragma once
#include "common.hpp"
#include "workspace.hpp"

namespace svoslam {
int generate_vertex_map(const uint16_t *d_depth, float *d_vertex, int w, int h, float fx, float fy, int img_w, int img_h, hipStream_t s);
int generate_vertex_map_rows(const uint16_t *d_depth, float *d_vertex, int w, int h, int first_row, int rows, float fx, float fy,
                             int img_w, int img_h, hipStream_t s);
int generate_normal_map(const float *d_vertex, float *d_normal, int w, int h, hipStream_t s);
// fused vertex + normal map straight from the depth image (same values as the two calls above)
int generate_vertex_normal_maps(const uint16_t *d_depth, float *d_vertex, float *d_normal, int w, int h, float fx, float fy,
                                int img_w, int img_h, hipStream_t s);
int bilateral_filter(const uint16_t *d_in, uint16_t *d_out, int w, int h, hipStream_t s);
int subsample_depth_u16(uint16_t *d_data, uint16_t *d_tmp, int w, int h, hipStream_t s);
int subsample_depth_f32(float *d_data, float *d_tmp, int w, int h, hipStream_t s);
// out-of-place variant used by the tracker (no copy back)
int subsample_depth_u16_to(const uint16_t *d_in, uint16_t *d_out, int w, int h, hipStream_t s);
int subsample_f32(float *d_data, float *d_tmp, int w, int h, hipStream_t s);
int subsample_rgb8(uint8_t *d_data, uint8_t *d_tmp, int w, int h, hipStream_t s);
int gradient(const float *d_in, float *d_grad2, int w, int h, hipStream_t s);       // Sobel / 8, (gx, gy) per pixel
int difference(const float *d_in1, const float *d_in2, float *d_out, int n, hipStream_t s);
int color_to_intensity(const uint8_t *d_rgb, float *d_out, int n, hipStream_t s);
int transform_vertex_map(float *d_v, const float trans[16], int n, hipStream_t s);
int transform_normal_map(float *d_v, const float trans[16], int n, hipStream_t s);
int transform_vertex_map_dmat(float *d_v, const float *d_trans, int n, hipStream_t s);
// non-blocking: d_out7 = {min x,y,z, max x,y,z, any_valid} on the device
int point_cloud_bbox_device(svoslam::DeviceBuffer &scratch, const float *d_points, int n, float *d_out7, hipStream_t s);
int point_cloud_bbox(svoslam::DeviceBuffer &scratch, const float *d_points, int n, float h_bbox0[3], float h_bbox1[3], hipStream_t s);
}  // namespace svoslam
