/*
 * svoslam_oracle.h -- CPU restatement of the dkotfis/Octree-SLAM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (octree-slam_amd/, include/,
 * bench.py's timed GPU region) may link, import or execute this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned by the reference".  The reference ships no
 * tests, golden vectors or fixtures for this path and cannot be compiled in
 * this image (CUDA 6.5 + Thrust + VoxelPipe + OpenNI + GL).  The
 * oracle is pinned instead against the hand-derived known-answer vectors of
 * SURVEY.md Appendix C (tests/golden/kat_*.json, tests/test_oracle_kat.py).
 * One exception is pinned by the reference itself: the OBJ loader
 * (ora_mesh_load_obj) is checked bit for bit against the reference's own
 * objUtil sources, compiled where they lie into oracle/_ref/libobjref.so
 * (`make ref`, ref_obj_shim.cpp; tests/test_ref_obj_loader.py and the digests
 * of its output in tests/golden/ref_obj_loader.json).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  Plain single-thread C, no SIMD intrinsics.
 *
 * Floating point: compiled with -ffp-contract=off; every float expression is
 * evaluated operation by operation in IEEE binary32 in the order the reference
 * source writes it.  Where the reference result depends on things its source
 * does not fix (thread races, thrust::reduce order, fast-math intrinsics,
 * log()/pow() rounding) the oracle fixes ONE deterministic resolution; each is
 * listed in DESIGN.md section "Deterministic resolutions".
 */
#ifndef SVOSLAM_ORACLE_H_
#define SVOSLAM_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t octkey; /* src/world/svo/svo.cu:22 */

#define ORA_FLAG_CHILDREN 0x40000000u
#define ORA_CHILD_MASK    0x3FFFFFFFu

/* ---------------------------------------------------------------- keys */
octkey ora_compute_key(const float p[3], const float center[3], int tree_depth, float edge_length); /* svo.cu:33-66 */
int    ora_depth_from_key(octkey key);                  /* svo.cu:68-78 (64-bit generalisation) */
int    ora_get_first_value_and_shift_down(octkey *key); /* svo.cu:84-90 */
void   ora_compute_keys(const float *pts, int stride_floats, int n, int max_depth,
                        const float center[3], float edge_length, octkey *keys); /* svo.cu:92-106 */

/* ---------------------------------------------------------------- pool */
typedef struct {
  uint32_t *data; /* 2 words per node, host memory (malloc) */
  int size;       /* nodes */
} ora_pool;

void ora_pool_free(ora_pool *p);
int ora_pool_load_words(ora_pool *p, const uint32_t *words, int num_nodes);

/* split planning, exposed for tests: returns number of split nodes, fills
 * pass_sizes[max_depth] and (if codes_out != NULL) the concatenated sorted
 * unique codes of every pass (caller frees *codes_out). svo.cu:108-237 */
int ora_prepass(const octkey *keys, int n, int max_depth, const uint32_t *octree,
                int *pass_sizes, octkey **codes_out);

/* svo.cu:642-696 ; colors = n x 3 bytes (Color256) */
void ora_svo_from_point_cloud(const float *points, const uint8_t *colors, int n, int max_depth,
                              ora_pool *pool, const float center[3], float edge_length);
/* svo.cu:584-640 ; centers/colors = n x 4 floats (glm::vec4) */
void ora_svo_from_voxel_grid(const float *centers, const float *colors, int n, int max_depth,
                             ora_pool *pool, const float center[3], float edge_length);
/* svo.cu:699-745 ; returns count; *centers,*colors malloc'ed n x 4 floats */
int ora_extract_voxel_grid(const ora_pool *pool, int max_depth, const float center[3],
                           float edge_length, float **centers, float **colors);

/* ---------------------------------------------------------------- render */
#define ORA_RENDER_REFERENCE 0 /* pixel written only on retirement */
#define ORA_RENDER_CARRY     1 /* local pixel carried across steps (intended behaviour) */
/* cone_tracing_kernels.cu:24-198 ; view = column-major mat4 ; pos = w*h*4 bytes.
 * Returns total number of march steps over all rays (for the bytes model). */
int64_t ora_cone_trace_svo(uint8_t *pos, int w, int h, float fov, const float view[16],
                           const uint32_t *octree, const float center[3], float size, int mode,
                           int64_t *levels_descended);

/* ---------------------------------------------------------------- sensor */
void ora_generate_vertex_map(const uint16_t *depth, float *vmap, int w, int h, float fx, float fy,
                             int img_w, int img_h);                          /* image_kernels.cu:24-58 */
void ora_generate_normal_map(const float *vmap, float *nmap, int w, int h); /* image_kernels.cu:104-139 */
void ora_bilateral_filter(const uint16_t *in, uint16_t *out, int w, int h); /* image_kernels.cu:142-186 */
void ora_subsample_depth_u16(uint16_t *data, int w, int h);                 /* image_kernels.cu:236-289 */
void ora_subsample_depth_f32(float *data, int w, int h);
void ora_subsample_f32(float *data, int w, int h);                          /* image_kernels.cu:291-326 */
void ora_subsample_rgb8(uint8_t *data, int w, int h);
void ora_color_to_intensity(const uint8_t *rgb, float *out, int n);         /* image_kernels.cu:188-203 */
void ora_transform_vertex_map(float *v, const float m[16], int n);          /* image_kernels.cu:206-219 */
void ora_transform_normal_map(float *v, const float m[16], int n);          /* image_kernels.cu:221-234 */
void ora_point_cloud_bbox(const float *pts, int n, float bbox0[3], float bbox1[3]); /* image_kernels.cu:60-102 */

/* localization_kernels.cu:154-229,303-326.  Fixed-point accumulation spec, see .c */
void ora_icp_cost2(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n,
                   int w, int h, float A[36], float b[6]);
/* the 27 raw accumulators (21 upper-triangle A terms row-major, then 6 b terms) */
int ora_icp_cost(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n, int w, int h,
                 float A[36], float b[6]);
int ora_icp_cost_raw(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n, int w, int h,
                     int64_t acc[27]);
void ora_icp_cost2_raw(const float *last_v, const float *last_n, const float *cur_v,
                       const float *cur_n, int first_pixel, int num_pixels, int w, int h,
                       int64_t acc[27]);
void ora_icp_finish(const int64_t acc[27], float A[36], float b[6]);
/* photometric RGB-D term: this build's own specification (the reference declares, never defines: image_kernels.h:45-49,
 * localization_kernels.cu:328-331, rgbd_camera.cpp:126-141) */
void ora_gradient(const float *in, float *grad2, int w, int h);
void ora_difference(const float *in1, const float *in2, float *out, int n);
void ora_rgbd_cost_raw(const float *last_i, const float *last_g, const float *last_v, const float *cur_i, const float *cur_v,
                       int w, int h, float fx, float fy, int img_w, int img_h, int64_t acc[27]);
void ora_rgbd_finish(const int64_t acc[27], float A[36], float b[6]);
void ora_rgbd_cost(const float *last_i, const float *last_g, const float *last_v, const float *cur_i, const float *cur_v, int w, int h,
                   float fx, float fy, int img_w, int img_h, float A[36], float b[6]);
/* rgbd_camera.cpp:194-222 */
void ora_solve_cholesky(int dim, const float *A, const float *b, float *x);

/* ---------------------------------------------------------------- mat4 (glm 0.9.5.4 semantics, column-major) */
void ora_mat4_identity(float m[16]);
void ora_mat4_mul(const float a[16], const float b[16], float out[16]);
void ora_mat4_inverse(const float m[16], float out[16]);
void ora_mat4_translate(const float m[16], const float v[3], float out[16]);
void ora_mat4_rotate_deg(const float m[16], float angle_deg, const float axis[3], float out[16]);
void ora_mat4_look_at(const float eye[3], const float center[3], const float up[3], float out[16]);
void ora_sincos(float a, float *s, float *c); /* deterministic sin/cos used by rotate */
/* this_trans of rgbd_camera.cpp:154-158 from the 6-vector x */
void ora_icp_update_transform(const float x[6], float out[16]);

/* ---------------------------------------------------------------- tracker (rgbd_camera.cpp:22-191) */
typedef struct ora_camera ora_camera;
ora_camera *ora_camera_create(int w, int h, float fx, float fy);
void ora_camera_destroy(ora_camera *c);
/* returns 1 if the frame was processed, 0 if skipped (stale timestamp) */
int ora_camera_update(ora_camera *c, const uint16_t *depth, const uint8_t *rgb, long long timestamp);
void ora_camera_pose(const ora_camera *c, float position[3], float orientation[9]);
void ora_camera_set_pose(ora_camera *c, const float position[3], const float orientation[9]);
/* frame-parallel tracking: a tracked frame's update_trans, and the pose step for an update_trans from anywhere */
void ora_camera_last_update(const ora_camera *c, float out[16]);
void ora_camera_set_strict_reference(ora_camera *c, int strict);
void ora_icp_cost2_raw_corrected(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n,
                                 int first_pixel, int num_pixels, int w, int h, int64_t acc[27]);
void ora_icp_update_transform_corrected(const float x[6], float out[16]);
int ora_camera_apply_delta(ora_camera *c, const float *update_trans, int levels_lost, long long timestamp);
int ora_camera_tracking_lost_count(const ora_camera *c);
void ora_camera_set_rgbd(ora_camera *c, int enable); /* adds W_RGBD x the photometric system to every ICP iteration */
/* frame-to-model tracking (own specification: see svoslam_oracle.c) */
int64_t ora_raycast_model_depth(uint16_t *depth_out, int w, int h, float fx, float fy, const float cam_to_world[16],
                                const uint32_t *octree, const float center[3], float size);
int ora_camera_set_model_depth(ora_camera *c, const uint16_t *depth);
int ora_camera_set_frame_to_model(ora_camera *c, int enable);
/* model matrix used by main.cpp:40 : mat4(orientation) * translate(I, position) */
void ora_camera_fusion_transform(const ora_camera *c, float out[16]);
/* last A,b,x of the last processed frame, for tests */
void ora_camera_last_system(const ora_camera *c, float A[36], float b[6], float x[6]);

/* ---------------------------------------------------------------- mesh path (svoslam_oracle_mesh.c) */
/* objloader.cpp:14-122 + obj.cpp:33-135,227-238 + scene.cpp:115-133: returns #triangles, -1 on open failure */
int ora_mesh_load_obj(const char *path, float **vbo, float **tbo, int *tbosize, float bbox0[3], float bbox1[3]);
/* scene.cpp:35-62 */
int ora_load_bmp(const char *path, float **data, int *width, int *height);
/* voxelization.cu:381-405 with the VoxelPipe THIN / NO_BLENDING rule; N = 2^log_N cells per axis */
int ora_mesh_to_voxel_grid(const float *vbo, int n_tris, const float *tbo, int tbosize, const float *tex, int tex_w,
                           int tex_h, const float bbox0[3], const float bbox1[3], int log_N, int log_T,
                           float **centers, float **colors, int64_t **indices);

/* voxelization::voxelGridToMesh / createCubeMesh (src/world/voxelization/voxelization.cu:184-217, :325-379):
 * one copy of the cube mesh per voxel, scaled by scale_factor = computeScale(bbox) / CUBE_MESH_SCALE and
 * moved to the voxel centre; colours replicated per vertex component; indices offset by idx * cube_ibosize
 * (the reference's own offset, kept).  Outputs hold n * cube_vbosize floats (vbo, nbo, cbo) and
 * n * cube_ibosize ints (ibo), allocated by the caller. */
void ora_voxel_grid_to_mesh(const float *centers, const float *colors, int n, float scale_factor, const float *cube_vbo,
                            int cube_vbosize, const int *cube_ibo, int cube_ibosize, const float *cube_nbo, float *out_vbo,
                            int *out_ibo, float *out_nbo, float *out_cbo);

#ifdef __cplusplus
}
#endif
#endif
