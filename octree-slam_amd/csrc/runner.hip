// runner.hip -- native frame scheduler of the SLAM loop (main.cpp:31-84 per frame: track -> back-project ->
// fuse -> raycast), software-pipelined over four HIP streams:
//
//   P  bilateral filter + vertex/normal pyramids of frame k+2 (three rotating map sets in the camera)
//   T  19 ICP iterations of frame k+1; touches only the camera state
//   S  back-projection, keys + sort and split planning of frame k+1 (reads the pool's tree)
//   M  commit of frame k (splits, leaf blend, mip levels -- the only writer of the pool), raycast of frame k
//
// Cross-stream order (events): S waits for the pose of its frame (from T) and, before planning, for the commit
// of the previous frame (from M); M waits for the plan of its frame; T waits, before it overwrites a slot of the
// 4-deep pose ring, for the back-projection that read that slot; P waits for the pose of frame k-2 (whose "last"
// map set it overwrites).  The same schedule as octree-slam_amd/pipeline.py::SlamPipeline.run_stream, which it
// replaces on the hot path: the Python loop needed 0.45 ms of host time per frame -- the whole frame time --
// for ~25 calls; here a frame costs the host about ten graph launches and a dozen event operations.
//
// Built on the public C ABI (include/svoslam.h) only: every stage is the call a reference-style host would make.
#include <hip/hip_runtime.h>

#include <vector>

#include "common.hpp"
#include "../../include/svoslam.h"

struct svoslam_runner {
  svoslam_camera *cam = nullptr;
  svoslam_pool *pool = nullptr;
  int w = 0, h = 0, depth = 0, mode = 0;
  float center[3] = {0, 0, 0}, edge = 0, fx = 0, fy = 0, fov = 45.0f;
  hipStream_t s_maps = nullptr, s_track = nullptr, s_prep = nullptr, s_map = nullptr;
  svoslam_workspace *ws[2] = {nullptr, nullptr};
  float *points[2] = {nullptr, nullptr};  // back-projected clouds of the two frames in flight
  float *bbox = nullptr;                  // 7 floats (main.cpp:44)
  // fixed input addresses per stream: the library replays its launch sequences as HIP graphs keyed on the
  // pointers it is given (graph_cache.hpp), so every frame is copied into a staging buffer first
  uint16_t *in_track = nullptr, *in_prep = nullptr;
  uint8_t *in_rgb = nullptr;
  std::vector<hipEvent_t> events;  // pool, grown on demand
  hipEvent_t ev_begin = nullptr, ev_end[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace {

int ensure_events(svoslam_runner *r, size_t n) {
  while (r->events.size() < n) {
    hipEvent_t e;
    SVO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    r->events.push_back(e);
  }
  return SVOSLAM_OK;
}

}  // namespace

extern "C" {

int svoslam_runner_create(svoslam_runner **out, svoslam_camera *cam, svoslam_pool *pool, int32_t width, int32_t height,
                          int32_t max_depth, const float center[3], float edge_length, float fx, float fy, int32_t render_mode) {
  if (!out || !cam || !pool || !center || width <= 0 || height <= 0) return SVOSLAM_ERR_INVALID_ARG;
  if (max_depth < 1 || max_depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  svoslam_runner *r = new svoslam_runner();
  r->cam = cam; r->pool = pool; r->w = width; r->h = height; r->depth = max_depth; r->mode = render_mode;
  for (int k = 0; k < 3; k++) r->center[k] = center[k];
  r->edge = edge_length; r->fx = fx; r->fy = fy;
  *out = r;
  const size_t n = (size_t)width * height;
  for (hipStream_t *s : {&r->s_maps, &r->s_track, &r->s_prep, &r->s_map}) SVO_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    SVO_TRY(svoslam_workspace_create(&r->ws[k]));
    SVO_HIP(hipMalloc((void **)&r->points[k], n * 12));
  }
  SVO_HIP(hipMalloc((void **)&r->bbox, 7 * 4));
  SVO_HIP(hipMemset(r->bbox, 0, 7 * 4));
  SVO_HIP(hipMalloc((void **)&r->in_track, n * 2));
  SVO_HIP(hipMalloc((void **)&r->in_prep, n * 2));
  SVO_HIP(hipMalloc((void **)&r->in_rgb, n * 3));
  SVO_HIP(hipEventCreateWithFlags(&r->ev_begin, hipEventDisableTiming));
  for (int k = 0; k < 4; k++) SVO_HIP(hipEventCreateWithFlags(&r->ev_end[k], hipEventDisableTiming));
  return SVOSLAM_OK;
}

int svoslam_runner_destroy(svoslam_runner *r) {
  if (!r) return SVOSLAM_OK;
  (void)hipDeviceSynchronize();
  for (hipEvent_t e : r->events) (void)hipEventDestroy(e);
  if (r->ev_begin) (void)hipEventDestroy(r->ev_begin);
  for (int k = 0; k < 4; k++) if (r->ev_end[k]) (void)hipEventDestroy(r->ev_end[k]);
  for (int k = 0; k < 2; k++) {
    if (r->ws[k]) svoslam_workspace_destroy(r->ws[k]);
    (void)hipFree(r->points[k]);
  }
  (void)hipFree(r->bbox); (void)hipFree(r->in_track); (void)hipFree(r->in_prep); (void)hipFree(r->in_rgb);
  for (hipStream_t s : {r->s_maps, r->s_track, r->s_prep, r->s_map}) if (s) (void)hipStreamDestroy(s);
  delete r;
  return SVOSLAM_OK;
}

// Enqueues n frames (device-resident depth u16 / RGB888 images, strictly increasing timestamps, one view matrix
// per frame for the raycast) and returns without waiting.  Work starts after everything already queued on
// caller_stream and caller_stream is made to wait for all of it: the caller synchronises that stream (or the
// device) to read d_image -- rows [row_first, row_first + rows) of the LAST frame's raycast -- the pool and the pose.
// d_steps (optional): 2 x u64 step / level counters accumulated over all raycasts.
int svoslam_runner_run(svoslam_runner *r, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs, const long long *timestamps,
                       const float *views, int32_t n, uint8_t *d_image, int32_t row_first, int32_t rows,
                       unsigned long long *d_steps, void *caller_stream) {
  if (!r || n < 0 || (n > 0 && (!d_depths || !d_rgbs || !timestamps || !views || !d_image))) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  hipStream_t cur = reinterpret_cast<hipStream_t>(caller_stream);
  const size_t px = (size_t)r->w * r->h;
  const int npts = (int)px;
  // events of this call: pose, back-projection, plan, commit, maps of every frame
  SVO_TRY(ensure_events(r, 5 * (size_t)n));
  hipEvent_t *ev_pose = r->events.data(), *ev_bp = ev_pose + n, *ev_plan = ev_bp + n, *ev_commit = ev_plan + n, *ev_maps = ev_commit + n;
  std::vector<const float *> fusion_ptr((size_t)n, nullptr);
  SVO_HIP(hipEventRecord(r->ev_begin, cur));
  for (hipStream_t s : {r->s_maps, r->s_track, r->s_prep, r->s_map}) SVO_HIP(hipStreamWaitEvent(s, r->ev_begin, 0));

  auto enqueue_maps = [&](int i) -> int {  // bilateral filter + pyramids of frame i (no dependence on earlier poses)
    if (i >= 2) SVO_HIP(hipStreamWaitEvent(r->s_maps, ev_pose[i - 2], 0));  // its map set was the "last" set of frame i-2
    SVO_HIP(hipMemcpyAsync(r->in_track, d_depths[i], px * 2, hipMemcpyDeviceToDevice, r->s_maps));
    int32_t used = 0;
    SVO_TRY(svoslam_camera_prepare(r->cam, r->in_track, d_rgbs[i], timestamps[i], &used, r->s_maps));
    if (!used) return SVOSLAM_ERR_INVALID_ARG;  // timestamps must increase strictly
    SVO_HIP(hipEventRecord(ev_maps[i], r->s_maps));
    return SVOSLAM_OK;
  };
  auto enqueue_track = [&](int i) -> int {
    if (i >= 4) SVO_HIP(hipStreamWaitEvent(r->s_track, ev_bp[i - 4], 0));  // ring slot i % 4 has been consumed
    SVO_HIP(hipStreamWaitEvent(r->s_track, ev_maps[i], 0));
    SVO_TRY(svoslam_camera_track(r->cam, r->s_track));
    fusion_ptr[i] = svoslam_camera_fusion_transform_device(r->cam);  // ring slot of frame i
    SVO_HIP(hipEventRecord(ev_pose[i], r->s_track));
    return SVOSLAM_OK;
  };
  auto enqueue_prepare = [&](int i) -> int {
    svoslam_workspace *ws = r->ws[i & 1];
    float *pts = r->points[i & 1];
    SVO_HIP(hipStreamWaitEvent(r->s_prep, ev_pose[i], 0));
    SVO_HIP(hipMemcpyAsync(r->in_prep, d_depths[i], px * 2, hipMemcpyDeviceToDevice, r->s_prep));
    SVO_TRY(svoslam_generate_vertex_map(r->in_prep, pts, r->w, r->h, r->fx, r->fy, r->w, r->h, r->s_prep));  // main.cpp:39
    SVO_TRY(svoslam_transform_vertex_map_dmat(pts, fusion_ptr[i], npts, r->s_prep));                         // main.cpp:40-41
    SVO_TRY(svoslam_point_cloud_bbox_device(r->ws[0], pts, npts, r->bbox, r->s_prep));                       // main.cpp:44
    SVO_HIP(hipEventRecord(ev_bp[i], r->s_prep));
    SVO_TRY(svoslam_svo_fuse_sort(ws, pts, npts, r->depth, r->center, r->edge, r->s_prep));
    if (i > 0) SVO_HIP(hipStreamWaitEvent(r->s_prep, ev_commit[i - 1], 0));  // the tree the plan reads
    SVO_TRY(svoslam_svo_fuse_plan(ws, npts, r->depth, r->pool, r->s_prep));
    SVO_HIP(hipEventRecord(ev_plan[i], r->s_prep));
    return SVOSLAM_OK;
  };

  SVO_TRY(enqueue_maps(0));
  SVO_TRY(enqueue_track(0));
  if (n > 1) SVO_TRY(enqueue_maps(1));
  SVO_TRY(enqueue_prepare(0));
  for (int i = 0; i < n; i++) {
    if (i + 1 < n) SVO_TRY(enqueue_track(i + 1));
    if (i + 2 < n) SVO_TRY(enqueue_maps(i + 2));
    SVO_HIP(hipStreamWaitEvent(r->s_map, ev_plan[i], 0));
    SVO_HIP(hipMemcpyAsync(r->in_rgb, d_rgbs[i], px * 3, hipMemcpyDeviceToDevice, r->s_map));
    SVO_TRY(svoslam_svo_fuse_commit(r->ws[i & 1], r->in_rgb, npts, r->depth, r->pool, r->s_map));
    SVO_HIP(hipEventRecord(ev_commit[i], r->s_map));
    if (i + 1 < n) SVO_TRY(enqueue_prepare(i + 1));  // host order: after ev_commit[i] has been recorded
    SVO_TRY(svoslam_cone_trace_svo_band(d_image, r->w, r->h, row_first, rows, r->fov, views + 16 * (size_t)i, r->pool->d_data,
                                        r->center, r->edge, r->mode, d_steps, r->s_map));
  }
  hipStream_t ss[4] = {r->s_maps, r->s_track, r->s_prep, r->s_map};
  for (int k = 0; k < 4; k++) {
    SVO_HIP(hipEventRecord(r->ev_end[k], ss[k]));
    SVO_HIP(hipStreamWaitEvent(cur, r->ev_end[k], 0));
  }
  return SVOSLAM_OK;
}

}  // extern "C"
