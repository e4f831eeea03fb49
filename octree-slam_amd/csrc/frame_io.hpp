// frame_io.hpp -- see frame_io.hip
#pragma once
#include <vector>

#include "common.hpp"

struct svoslam_frame_reader;

namespace svoslam {
struct HostImage {
  int width = 0, height = 0, channels = 0, bits = 0;  // bits per sample: 8 or 16 (16-bit samples in host byte order)
  std::vector<uint8_t> data;
};
int image_load(const char *path, HostImage &img);  // PNG / PGM / PPM
int frame_reader_open(svoslam_frame_reader **out, const char *association_file, float depth_units_per_metre);
int frame_reader_close(svoslam_frame_reader *r);
int frame_reader_info(const svoslam_frame_reader *r, int *width, int *height, int *num_frames);
int frame_reader_rewind(svoslam_frame_reader *r);
int frame_reader_next_host(svoslam_frame_reader *r, uint16_t *h_depth, uint8_t *h_color, long long *timestamp, int *got);
int frame_reader_next(svoslam_frame_reader *r, uint16_t *d_depth, uint8_t *d_color, long long *timestamp, int *got, hipStream_t stream);
int focal_from_fov(int width, int height, float hfov_rad, float vfov_rad, float *fx, float *fy);
}  // namespace svoslam
