// mesh.hpp -- see mesh.hip
#pragma once
#include "common.hpp"
#include "workspace.hpp"

namespace svoslam {
int mesh_load_obj(const char *path, svoslam_mesh *out);
int mesh_free(svoslam_mesh *m);
int texture_load_bmp(const char *path, svoslam_texture *out);
int texture_free(svoslam_texture *t);
int mesh_to_voxel_grid(svoslam_workspace *ws, const svoslam_mesh *mesh, const svoslam_texture *tex, int log_N, int log_T,
                       float **d_centers, float **d_colors, unsigned long long **d_indices, int32_t *n_out, float *scale_out,
                       hipStream_t stream);
int voxel_grid_to_mesh(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int n, float scale_factor,
                       const float *cube_vbo, int cube_vbosize, const int *cube_ibo, int cube_ibosize, const float *cube_nbo,
                       float *d_vbo, int *d_ibo, float *d_nbo, float *d_cbo, hipStream_t stream);
}  // namespace svoslam
