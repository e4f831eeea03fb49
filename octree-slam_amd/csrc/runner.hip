// runner.hip -- native frame scheduler of the SLAM loop (main.cpp:31-84 per frame: track -> back-project ->
// fuse -> raycast), software-pipelined over HIP streams:
//
//   P   bilateral filter + vertex/normal pyramids of frame k+2 (three rotating map sets in the camera)
//   T   the ICP iterations of frame k+1 (one launch); touches only the camera state
//   S   back-projection, keys + sort and split planning of frame k+1 (reads a pool's tree)
//   M0  commits to replica 0 of the map, raycasts of the even frames
//   M1  commits to replica 1 of the map, raycasts of the odd frames
//
// Two replicas of the map.  A frame's raycast (~0.3 ms: bound by the dependent-load latency of its longest rays, not by
// CUs) and the next frame's commit cannot touch one pool at the same time, and with one pool their sum is the frame
// period.  The scheduler therefore keeps a second, byte-identical replica of the caller's pool: every plan is applied
// to both (svoslam_svo_fuse_commit_to -- the commit is a deterministic function of plan + pool, so the replicas stay
// identical), and the frames are ray-marched on them alternately.  While frame k is marched on replica k & 1 the
// commits of frames k+1 and k+2 go to the other one, so each replica carries two commits and one raycast per TWO
// frames.  Cost: the commit kernels run twice and the map takes twice the HBM (a few GB of 288).
// MEASURED (round 2, cfg3, profiles/r02_runner_timeline_*.txt): with ONE replica the M stream bounds the frame at
// commit 0.133 ms + build/march 0.272 ms = 0.405 ms; with TWO the replica streams do overlap but every kernel gets
// slower -- march 0.27 -> 0.41 ms, commit 0.13 -> 0.22-0.29 ms, maps 0.10 -> 0.37 ms -- because the march keeps
// ~4800 wavefronts (66 % of the VGPR file) resident for its whole duration and the tracker's 151 workgroups want
// the other half: 0.52-0.55 ms per frame against 0.41-0.43.  The schedule is therefore OPT-IN (svoslam_config.runner_replicas = 2;
// it is also the single-GPU form of pipelining the stages over several GPUs, where each replica has a GPU to itself);
// the default is one pool.
//
// Cross-stream order (events): S waits for the pose of its frame (from T) and, before planning, for commit k-1 on the
// replica it reads; an M stream waits for the plan of the frame it commits; T waits, before it overwrites a slot of
// the 4-deep pose ring, for the back-projection that read that slot; P waits for the pose of frame k-2 (whose "last"
// map set it overwrites); S waits, before it reuses a workspace / point buffer / colour staging buffer (rings of three),
// for both commits of the frame that used them.  Results are those of calling the stages one after the other.
//
// Built on the public C ABI (include/svoslam.h) only: every stage is the call a reference-style host would make.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "config.hpp"
#include "../../include/svoslam.h"

namespace {
constexpr int kRing = 3;  // frames whose fusion may be in flight: workspaces, point clouds, colour staging
}

// stream sets of destroyed runners, by kind (see svoslam_runner_create)
static std::mutex g_streams_mu;
static std::map<int, std::vector<std::array<hipStream_t, 5>>> g_free_streams;

struct svoslam_runner {
  svoslam_camera *cam = nullptr;
  svoslam_pool *pool = nullptr;     // replica 0: the caller's pool
  svoslam_pool replica1;            // replica 1 (owned); d_data == nullptr while unused
  int replicas = 1;
  int w = 0, h = 0, depth = 0, mode = 0;
  float center[3] = {0, 0, 0}, edge = 0, fx = 0, fy = 0, fov = 45.0f;
  hipStream_t s_maps = nullptr, s_track = nullptr, s_prep = nullptr, s_map[2] = {nullptr, nullptr};
  svoslam_workspace *ws[kRing] = {nullptr, nullptr, nullptr};
  float *points[kRing] = {nullptr, nullptr, nullptr};  // back-projected clouds of the frames in flight
  uint8_t *in_rgb[kRing] = {nullptr, nullptr, nullptr};
  float *bbox = nullptr;                   // 7 floats (main.cpp:44)
  uint8_t *scratch_image[2] = {nullptr, nullptr};  // raycasts of all but the last frame of a call
  // fixed input addresses per stream: the library replays its launch sequences as HIP graphs keyed on the
  // pointers it is given (graph_cache.hpp), so every frame is copied into a staging buffer first
  uint16_t *in_track = nullptr, *in_prep = nullptr;
  uint16_t *model_depth = nullptr;         // svoslam_runner_run_model: the map ray-cast from the pose just tracked
  unsigned *model_count = nullptr;         // ... and the number of its pixels that met the map
  std::vector<hipEvent_t> events;  // pool, grown on demand
  hipEvent_t ev_begin = nullptr, ev_end[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t last_caller = nullptr;
  bool model_pending = false;   // svoslam_runner_run_model left its model in the camera (cleared by the next svoslam_runner_run)
  int lead = -1;  // commits the host may run ahead of the device (see svoslam_runner_run); < 0: the schedule's default (1 deferred, 2 in place)
  bool fused_front = false;  // back-projection + bounding box + keys in one launch (keys that do not fit the packed word: the stand-alone calls)
  bool early_split = true;  // split_all_kernel right behind the plan, beside the previous frame's march
  // one replica: the commit of frame k+1 is computed during the march of frame k (svoslam_svo_fuse_commit_deferred).  Default
  // since round 3 for images up to 640x480-class: the march over occupancy bricks is bound by instruction issue and no
  // longer by the loads the commit competes for (cfg3, 300-frame map: 1862 -> 2055 frames/s; the driver's 20 frames 1565 ->
  // 1707; cfg4, where the launch-chain tracker bounds the frame, 815 -> 694: off there).  svoslam_config.runner_deferred = 0 / 1 overrides.
  int stream_kind = 0;  // device x priorities: which free list the five streams return to
  bool deferred = false, deferred_explicit = false;
  bool ran = false;
  // svoslam_config.runner_timeline = 1: timing events at the stage boundaries of the last call (svoslam_runner_timeline)
  bool timeline = false;
  std::vector<hipEvent_t> tl_events;
  int tl_frames = 0;
};

namespace { constexpr int kTlStages = 10; }
  // maps0 maps1 track0 track1 prep0 plan0 plan1 commit0 commit1(first replica) ray1

namespace {

int ensure_events(svoslam_runner *r, size_t n) {
  while (r->events.size() < n) {
    hipEvent_t e;
    SVO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    r->events.push_back(e);
  }
  return SVOSLAM_OK;
}

svoslam_pool *replica(svoslam_runner *r, int k) { return k == 0 ? r->pool : &r->replica1; }

}  // namespace

extern "C" {

int svoslam_runner_create(svoslam_runner **out, svoslam_camera *cam, svoslam_pool *pool, int32_t width, int32_t height,
                          int32_t max_depth, const float center[3], float edge_length, float fx, float fy, int32_t render_mode) {
  if (!out || !cam || !pool || !center || width <= 0 || height <= 0) return SVOSLAM_ERR_INVALID_ARG;
  if (max_depth < 1 || max_depth > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_DEPTH;
  svoslam_runner *r = new svoslam_runner();
  memset(&r->replica1, 0, sizeof(r->replica1));
  r->cam = cam; r->pool = pool; r->w = width; r->h = height; r->depth = max_depth; r->mode = render_mode;
  for (int k = 0; k < 3; k++) r->center[k] = center[k];
  r->edge = edge_length; r->fx = fx; r->fy = fy;
  const svoslam_config cfg = svoslam::config();  // (settings are taken when the runner is created)
  r->replicas = cfg.runner_replicas == 2 ? 2 : 1;
  r->timeline = cfg.runner_timeline != 0;
  r->deferred_explicit = cfg.runner_deferred >= 0;
  r->deferred = cfg.runner_deferred >= 0 ? cfg.runner_deferred == 1 : ((long long)width * height <= 400000ll && r->replicas == 1);
  {
    int idx_bits = 1;
    while ((1ll << idx_bits) < (long long)width * height) idx_bits++;
    r->fused_front = 3 * max_depth + 1 + idx_bits <= 64 && cfg.sort_pairs == 0;
  }
  if (cfg.runner_lead >= 0) r->lead = cfg.runner_lead;
  *out = r;
  const size_t n = (size_t)width * height;
  {
    // Stream priorities: the map stream -- the chain that bounds the frame -- on the highest, the tracker in the middle, maps
    // and sort on the lowest.  Measured (means of 6 / 3 runs): cfg3 over the driver's 20 frames 2690 -> 2795 frames/s with
    // half the run-to-run spread, 100 and 300 frames and a rank of 8 unchanged, cfg4 748 -> 733 (there the launch-chain
    // tracker is level with the map stream and loses what the map stream gains).  Default: on for images up to 640x480-class
    // (the one-launch tracker's domain); svoslam_config.runner_prio = 0 / 1 overrides.  (Measured and not kept: the tracker
    // first -- cfg4 769 -> 774, within the noise --, the front-end stream raised: the slow steady state more often.)
    int least = 0, greatest = 0;
    const bool small = (long long)width * height <= 400000ll;
    const bool want = cfg.runner_prio >= 0 ? cfg.runner_prio != 0 : small;
    const bool prio = want && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
    hipStream_t *ss[5] = {&r->s_maps, &r->s_track, &r->s_prep, &r->s_map[0], &r->s_map[1]};
    const int mid = (least + greatest) / 2;
    // (round 6, re-measured with the shorter march, profiles/r06_priority_experiments.txt: the sort / plan stream at the middle priority
    // and / or s_setprio(3) in its kernels: 640x480 -1..0 %, 1080p +1..2 %: within the noise, not kept)
    const int pr[5] = {least, mid, least, greatest, greatest};
    // A destroyed runner's five streams are kept for the next runner of the same kind (device, priorities) instead of being
    // destroyed: streams created after others were destroyed can come to share hardware queues -- a runner created after
    // another one's destruction ran its sort behind its own march, 2700 -> 1300 frames/s at 640x480 (bench.py's second
    // pipeline; scratch measurement in DESIGN.md section 6).
    int dev = 0;
    SVO_HIP(hipGetDevice(&dev));
    r->stream_kind = dev * 2 + (prio ? 1 : 0);
    bool reused = false;
    {
      std::lock_guard<std::mutex> lock(g_streams_mu);
      auto &free_sets = g_free_streams[r->stream_kind];
      if (!free_sets.empty()) {
        for (int k = 0; k < 5; k++) *ss[k] = free_sets.back()[k];
        free_sets.pop_back();
        reused = true;
      }
    }
    for (int k = 0; k < 5 && !reused; k++) {
      if (prio) SVO_HIP(hipStreamCreateWithPriority(ss[k], hipStreamNonBlocking, pr[k]));
      else SVO_HIP(hipStreamCreateWithFlags(ss[k], hipStreamNonBlocking));
    }
  }
  for (int k = 0; k < kRing; k++) {
    SVO_TRY(svoslam_workspace_create(&r->ws[k]));
    SVO_HIP(hipMalloc((void **)&r->points[k], n * 12));
    SVO_HIP(hipMalloc((void **)&r->in_rgb[k], n * 3));
  }
  for (int k = 0; k < 2; k++) SVO_HIP(hipMalloc((void **)&r->scratch_image[k], n * 4));
  SVO_HIP(hipMalloc((void **)&r->bbox, 7 * 4));
  SVO_HIP(svoslam::memset_sync(r->bbox, 0, 7 * 4));
  SVO_HIP(hipMalloc((void **)&r->in_track, n * 2));
  SVO_HIP(hipMalloc((void **)&r->in_prep, n * 2));
  SVO_HIP(hipEventCreateWithFlags(&r->ev_begin, hipEventDisableTiming));
  for (int k = 0; k < 5; k++) SVO_HIP(hipEventCreateWithFlags(&r->ev_end[k], hipEventDisableTiming));
  return SVOSLAM_OK;
}

int svoslam_runner_destroy(svoslam_runner *r) {
  if (!r) return SVOSLAM_OK;
  (void)hipDeviceSynchronize();
  for (hipEvent_t e : r->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : r->tl_events) (void)hipEventDestroy(e);
  if (r->ev_begin) (void)hipEventDestroy(r->ev_begin);
  for (int k = 0; k < 5; k++) if (r->ev_end[k]) (void)hipEventDestroy(r->ev_end[k]);
  for (int k = 0; k < kRing; k++) {
    if (r->ws[k]) svoslam_workspace_destroy(r->ws[k]);
    (void)hipFree(r->points[k]);
    (void)hipFree(r->in_rgb[k]);
  }
  for (int k = 0; k < 2; k++) (void)hipFree(r->scratch_image[k]);
  (void)hipFree(r->bbox); (void)hipFree(r->in_track); (void)hipFree(r->in_prep);
  if (r->model_depth) (void)hipFree(r->model_depth);
  if (r->model_count) (void)hipFree(r->model_count);
  {
    // (the streams' render tables are released; the streams themselves wait for the next runner: see svoslam_runner_create)
    std::array<hipStream_t, 5> set = {r->s_maps, r->s_track, r->s_prep, r->s_map[0], r->s_map[1]};
    bool whole = true;
    for (hipStream_t s : set) {
      if (s) (void)svoslam_cone_trace_release(s, 0);
      else whole = false;
    }
    if (whole) {
      std::lock_guard<std::mutex> lock(g_streams_mu);
      g_free_streams[r->stream_kind].push_back(set);
    } else {
      for (hipStream_t s : set) if (s) (void)hipStreamDestroy(s);
    }
  }
  if (r->replica1.d_data) (void)svoslam_pool_free(&r->replica1);
  delete r;
  return SVOSLAM_OK;
}

// Enqueues n frames (device-resident depth u16 / RGB888 images, strictly increasing timestamps newer than any the
// camera has seen, one view matrix per frame for the raycast) and returns without waiting for them.  Work starts after
// everything already queued on caller_stream and caller_stream is made to wait for all of it: the caller synchronises
// that stream (or the device) to read d_image -- rows [row_first, row_first + rows) of the LAST frame's raycast (the
// earlier frames of the call are marched into internal buffers) -- the pool and the pose.
// d_steps (optional): 2 x u64 step / level counters accumulated over all raycasts.
// All arguments are validated BEFORE anything is enqueued; if a stage fails later, the streams are still joined to
// caller_stream before the error is returned.  Calls on one runner must use one caller_stream (checked).  With two
// replicas the call first brings replica 1 up to date with the caller's pool (blocking device copy).
//
// Sharded form (svoslam_runner_run_sharded): d_deltas != nullptr.  The poses come from
// svoslam_camera_apply_delta(d_deltas[i]) instead of the tracker -- the frames were tracked elsewhere (other ranks, other
// streams: svoslam_camera_pair_delta), delta_events[i] (optional) is what stream T waits for before it reads d_deltas[i] --
// the camera is never handed a depth image, every commit is applied, and only the frames with march[i] != 0 are ray-marched,
// into d_images[i].
static int runner_run_impl(svoslam_runner *r, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs, const long long *timestamps,
                           const float *views, int32_t n, uint8_t *d_image, int32_t row_first, int32_t rows,
                           unsigned long long *d_steps, void *caller_stream, const float *const *d_deltas,
                           void *const *delta_events, const uint8_t *march, uint8_t *const *d_images,
                           const unsigned long long *const *d_sorted_keys = nullptr, const uint32_t *const *d_sorted_idx = nullptr,
                           void *const *sorted_events = nullptr) {
  const bool sharded = d_deltas != nullptr;
  // presorted (frame-sharded sessions with a sharded SORT): frame i's sorted keys / point indices come
  // from whichever rank owns the frame (all-gathered); back-projection + keys + sort are skipped here, the plan adopts them
  const bool presorted = d_sorted_keys != nullptr;
  if (presorted && (!sharded || !d_sorted_idx)) return SVOSLAM_ERR_INVALID_ARG;
  if (presorted)
    for (int i = 0; i < n; i++)
      if (!d_sorted_keys[i] || !d_sorted_idx[i]) return SVOSLAM_ERR_INVALID_ARG;
  if (!r || n < 0 || (n > 0 && (!d_depths || !d_rgbs || !timestamps || !views || (!d_image && !d_images)))) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  if (row_first < 0 || rows < 0 || row_first + rows > r->h) return SVOSLAM_ERR_INVALID_ARG;
  if (sharded && (r->replicas != 1 || (r->deferred && r->deferred_explicit))) return SVOSLAM_ERR_INVALID_ARG;
  const bool deferred = r->deferred && !sharded;  // (frame-sharded sessions keep the in-place commit)
  if (d_images)
    for (int i = 0; i < n; i++)
      if ((!march || march[i]) && !d_images[i]) return SVOSLAM_ERR_INVALID_ARG;
  hipStream_t cur = reinterpret_cast<hipStream_t>(caller_stream);
  if (r->ran && cur != r->last_caller) return SVOSLAM_ERR_INVALID_ARG;  // the join below orders calls on ONE caller stream only
  {  // timestamps: strictly increasing and newer than the camera's latest (a stale frame would be skipped mid-pipeline)
    int32_t have = 0; long long latest = 0;
    SVO_TRY(svoslam_camera_latest_timestamp(r->cam, &have, &latest));
    for (int i = 0; i < n; i++) {
      if (!d_depths[i] || !d_rgbs[i]) return SVOSLAM_ERR_INVALID_ARG;
      if ((i == 0 && have && timestamps[0] <= latest) || (i > 0 && timestamps[i] <= timestamps[i - 1])) return SVOSLAM_ERR_INVALID_ARG;
    }
  }
  if (r->model_pending) {
    // a model left by svoslam_runner_run_model: this loop never refreshes it, and would track every frame against a map view that
    // ages with every frame it fuses -- back to frame-to-frame tracking, as the loop is specified
    SVO_TRY(svoslam_camera_set_model_depth(r->cam, nullptr, cur));
    SVO_TRY(svoslam_camera_set_frame_to_model(r->cam, 0));
    r->model_pending = false;
  }
  const size_t px = (size_t)r->w * r->h;
  const int npts = (int)px;
  const int R = r->replicas;
  if (R == 2) {
    // replica 1 := the caller's pool as it is now (the caller may have fused, loaded or reset it since the last call)
    SVO_HIP(hipStreamSynchronize(cur));
    const int rc = svoslam_pool_copy(&r->replica1, r->pool, cur);
    if (rc != SVOSLAM_OK) return rc;
  }
  // events of this call: maps, pose, back-projection, plan of every frame; commit of every frame on every replica
  SVO_TRY(ensure_events(r, 8 * (size_t)n));
  hipEvent_t *ev_maps = r->events.data(), *ev_pose = ev_maps + n, *ev_bp = ev_pose + n, *ev_plan = ev_bp + n;
  hipEvent_t *ev_commit[2] = {ev_plan + n, ev_plan + 2 * (size_t)n};
  hipEvent_t *ev_ray = ev_plan + 3 * (size_t)n;
  hipEvent_t *ev_sorted = ev_ray + n;
  // Large images with in-place commits (1920x1080): back-projection + sort alone fill the S stream (0.83 of the 1.05 ms period in the
  // scheduler's timeline, the plan the other 0.22), so the plan runs on the maps stream, which has 0.8 ms to spare: 932 -> 963 frames/s
  const bool plan_on_maps = !sharded && !deferred && R == 1 && (long long)r->w * r->h > 400000ll;
  const bool serial_marches = true;  // (two replicas: their marches one after the other)
  std::vector<const float *> fusion_ptr((size_t)n, nullptr);
  r->ran = true; r->last_caller = cur;
  if (r->timeline) {
    while (r->tl_events.size() < (size_t)kTlStages * n) {
      hipEvent_t e;
      SVO_HIP(hipEventCreate(&e));
      r->tl_events.push_back(e);
    }
    r->tl_frames = n;
  }
  auto mark = [&](int i, int stage, hipStream_t s) { if (r->timeline) (void)hipEventRecord(r->tl_events[(size_t)i * kTlStages + stage], s); };
  SVO_HIP(hipEventRecord(r->ev_begin, cur));
  hipStream_t all[5] = {r->s_maps, r->s_track, r->s_prep, r->s_map[0], r->s_map[1]};
  for (hipStream_t s : all) SVO_HIP(hipStreamWaitEvent(s, r->ev_begin, 0));

  const bool staged = svoslam::config().graphs != 0;  // graphs are keyed on pointers: frames are staged through fixed buffers
  // frame-sharded ranks that march at most one frame in three plan on the map stream (see enqueue_commit)
  bool plan_on_map = false;
  // ... or, by default, run the STRUCTURE CHAIN (svoslam_svo_fuse_plan_structure): a plan reads structure words only, so
  // plan + splits of frame i+1 need the splits of frame i, not its leaf blend / mip levels.  Three chains side by side --
  // front end + sort on the (otherwise idle) maps stream, plan + splits on S, leaf kernel + straddlers (+ this rank's
  // marches) on M -- and a frame costs the longest of them instead of plan + commit in sequence.  A march of frame k must not
  // see the structure of frame k+1: S waits for the rank's latest march before the next plan.
  bool chain = false;
  int last_march = -1;  // latest frame of this call whose march has been enqueued (chain: ev_ray[last_march] is recorded)
  if (sharded && march) {
    int marched = 0;
    for (int i = 0; i < n; i++) marched += march[i] ? 1 : 0;
    plan_on_map = 3 * marched <= n;
    if (plan_on_map && R == 1 && !deferred) { chain = true; plan_on_map = false; }
  }
  if (chain) SVO_TRY(svoslam_pool_structure_begin(r->pool, r->s_prep));
  hipStream_t s_maps = r->s_maps;
  auto enqueue_maps = [&](int i) -> int {  // bilateral filter + pyramids of frame i (no dependence on earlier poses)
    if (sharded) return SVOSLAM_OK;  // tracked elsewhere: this camera only composes poses
    if (i >= 2) SVO_HIP(hipStreamWaitEvent(s_maps, ev_pose[i - 2], 0));  // its map set was the "last" set of frame i-2
    mark(i, 0, s_maps);
    // fixed input addresses only where the library replays recorded launch sequences (graphs are keyed on pointers);
    // the caller's frames stay valid for the whole call (its stream is joined at the end)
    if (staged) SVO_HIP(hipMemcpyAsync(r->in_track, d_depths[i], px * 2, hipMemcpyDeviceToDevice, s_maps));
    int32_t used = 0;
    SVO_TRY(svoslam_camera_prepare(r->cam, staged ? r->in_track : d_depths[i], d_rgbs[i], timestamps[i], &used, s_maps));
    if (!used) return SVOSLAM_ERR_INVALID_ARG;  // cannot happen after the validation above
    SVO_HIP(hipEventRecord(ev_maps[i], s_maps));
    mark(i, 1, s_maps);
    return SVOSLAM_OK;
  };
  auto enqueue_track = [&](int i) -> int {
    if (i >= 4) SVO_HIP(hipStreamWaitEvent(r->s_track, ev_bp[i - 4], 0));  // ring slot i % 4 has been consumed
    if (!sharded) SVO_HIP(hipStreamWaitEvent(r->s_track, ev_maps[i], 0));
    if (sharded) { mark(i, 0, r->s_track); mark(i, 1, r->s_track); }  // (no maps here; keeps the timeline's origin)
    mark(i, 2, r->s_track);
    if (sharded) {
      if (delta_events && delta_events[i]) SVO_HIP(hipStreamWaitEvent(r->s_track, reinterpret_cast<hipEvent_t>(delta_events[i]), 0));
      int32_t used = 0;
      SVO_TRY(svoslam_camera_apply_delta(r->cam, d_deltas[i], timestamps[i], &used, r->s_track));
      if (!used) return SVOSLAM_ERR_INVALID_ARG;  // cannot happen after the validation above
    } else {
      SVO_TRY(svoslam_camera_track(r->cam, r->s_track));
    }
    fusion_ptr[i] = svoslam_camera_fusion_transform_device(r->cam);  // ring slot of frame i
    SVO_HIP(hipEventRecord(ev_pose[i], r->s_track));
    mark(i, 3, r->s_track);
    return SVOSLAM_OK;
  };
  auto enqueue_prepare = [&](int i) -> int {
    svoslam_workspace *ws = r->ws[i % kRing];
    float *pts = r->points[i % kRing];
    hipStream_t s_plan = r->s_prep;  // where the plan runs
    // chain: front end + sort of this frame on the maps stream, idle in a sharded call.  (Alternating the sorts of
    // consecutive frames over TWO idle streams was measured: each sort then takes 0.5 ms instead of 0.14 and a rank of 8
    // drops from 5450 to 3500 frames/s -- six busy streams on the runtime's hardware queues.)
    hipStream_t s_sort = chain ? r->s_maps : r->s_prep;
    SVO_HIP(hipStreamWaitEvent(s_sort, ev_pose[i], 0));
    if (i >= kRing)  // the ring slot's previous user: both of its commits are done with workspace, points and colours
      for (int k = 0; k < R; k++) SVO_HIP(hipStreamWaitEvent(s_sort, ev_commit[k][i - kRing], 0));
    mark(i, 4, s_sort);
    if (staged) {
      SVO_HIP(hipMemcpyAsync(r->in_prep, d_depths[i], px * 2, hipMemcpyDeviceToDevice, s_sort));
      SVO_HIP(hipMemcpyAsync(r->in_rgb[i % kRing], d_rgbs[i], px * 3, hipMemcpyDeviceToDevice, s_sort));
    }
    if (presorted) {
      if (sorted_events && sorted_events[i]) SVO_HIP(hipStreamWaitEvent(s_sort, reinterpret_cast<hipEvent_t>(sorted_events[i]), 0));
      SVO_TRY(svoslam_svo_fuse_adopt_sorted(ws, d_sorted_keys[i], d_sorted_idx[i], npts, r->depth));
      SVO_HIP(hipEventRecord(ev_bp[i], s_sort));
    } else if (r->fused_front) {
      // main.cpp:39-44 + computeKeys in one launch, no point cloud in memory (svoslam_svo_fuse_sort_frame), then the sort
      SVO_TRY(svoslam_svo_fuse_sort_frame(ws, staged ? r->in_prep : d_depths[i], fusion_ptr[i], r->w, r->h, r->fx, r->fy, r->depth,
                                          r->center, r->edge, r->bbox, s_sort));
      SVO_HIP(hipEventRecord(ev_bp[i], s_sort));
    } else {
      SVO_TRY(svoslam_generate_vertex_map(staged ? r->in_prep : d_depths[i], pts, r->w, r->h, r->fx, r->fy, r->w, r->h, s_sort));  // main.cpp:39
      SVO_TRY(svoslam_transform_vertex_map_dmat(pts, fusion_ptr[i], npts, s_sort));                         // main.cpp:40-41
      SVO_TRY(svoslam_point_cloud_bbox_device(r->ws[0], pts, npts, r->bbox, s_sort));                       // main.cpp:44
      SVO_HIP(hipEventRecord(ev_bp[i], s_sort));
      SVO_TRY(svoslam_svo_fuse_sort(ws, pts, npts, r->depth, r->center, r->edge, s_sort));
    }
    if (plan_on_map) {  // the plan moves to the map stream (enqueue_commit): see there
      mark(i, 5, s_sort);
      SVO_HIP(hipEventRecord(ev_plan[i], s_sort));
      return SVOSLAM_OK;
    }
    if (chain) {
      SVO_HIP(hipEventRecord(ev_maps[i], s_sort));        // "sorted" (the maps events are free in a sharded call)
      SVO_HIP(hipStreamWaitEvent(s_plan, ev_maps[i], 0));
      if (last_march >= 0) SVO_HIP(hipStreamWaitEvent(s_plan, ev_ray[last_march], 0));  // no structure of frame i under that march
      mark(i, 5, s_plan);
      SVO_TRY(svoslam_svo_fuse_plan_structure(ws, npts, r->depth, r->pool, s_plan));
      SVO_HIP(hipEventRecord(ev_plan[i], s_plan));
      mark(i, 6, s_plan);
      return SVOSLAM_OK;
    }
    // the plan reads the replica that receives commit i-1 FIRST (the one frame i-1 is marched on); the march only reads
    const int src = i > 0 ? ((i - 1) & (R - 1)) : 0;
    if (plan_on_maps) {  // (see above)
      SVO_HIP(hipEventRecord(ev_sorted[i], r->s_prep));
      s_plan = r->s_maps;
      SVO_HIP(hipStreamWaitEvent(s_plan, ev_sorted[i], 0));
    }
    if (i > 0) SVO_HIP(hipStreamWaitEvent(s_plan, ev_commit[src][i - 1], 0));
    mark(i, 5, s_plan);
    svoslam_pool *planned = replica(r, src);
    const int32_t cap_before = planned->capacity;
    SVO_TRY(svoslam_svo_fuse_plan(ws, npts, r->depth, planned, s_plan));
    if (R == 2 && planned->capacity != cap_before) {
      // the plan had to grow its replica (it waited for the whole device to do so): the other one follows
      svoslam_pool *other = replica(r, src ^ 1);
      SVO_HIP(hipDeviceSynchronize());
      SVO_TRY(svoslam_pool_reserve(other, planned->capacity, r->s_prep));
    }
    // the child tiles of this frame's splits, beyond the pool's size, while the previous frame is still being marched:
    // the commit on the map stream -- the stream that bounds the frame -- is then two launches instead of three
    if (R == 1 && !deferred && r->early_split) SVO_TRY(svoslam_svo_fuse_split_early(ws, npts, r->depth, planned, s_plan));
    SVO_HIP(hipEventRecord(ev_plan[i], s_plan));
    mark(i, 6, s_plan);
    return SVOSLAM_OK;
  };
  auto enqueue_commit = [&](int i, int k, bool last) -> int {
    SVO_HIP(hipStreamWaitEvent(r->s_map[k], ev_plan[i], 0));
    if (plan_on_map) {
      // Frame-sharded ranks march one frame in N: the map stream has room, and the stream that back-projects and sorts
      // is what bounds the frame (0.145 + 0.045 ms at cfg3).  The plan -- which needs commit i-1 anyway -- runs HERE, in
      // order behind that commit, and S goes straight on to the next frame's sort: 4440 -> 5240 frames/s for a rank of 8,
      // 3870 -> 4280 for a rank of 4 (a rank of 2 marches every other frame and loses 2 %: it keeps the plan on S).
      mark(i, 6, r->s_map[k]);  // (plan begin; mark 5 = sort end)
      SVO_TRY(svoslam_svo_fuse_plan(r->ws[i % kRing], npts, r->depth, r->pool, r->s_map[k]));
    }
    if (k == (i & (R - 1))) mark(i, 7, r->s_map[k]);
    SVO_TRY(svoslam_svo_fuse_commit_to(r->ws[i % kRing], staged ? r->in_rgb[i % kRing] : d_rgbs[i], npts, r->depth, replica(r, k), k,
                                       last ? 0 : 1, r->s_map[k]));
    SVO_HIP(hipEventRecord(ev_commit[k][i], r->s_map[k]));
    if (k == (i & (R - 1))) mark(i, 8, r->s_map[k]);
    return SVOSLAM_OK;
  };
  // One replica, deferred commits (the default for images up to 640x480-class, svoslam_runner::deferred): stream C computes the commit of frame k+1 -- splits, leaf
  // blends, mip levels, into memory the march cannot see (svoslam_svo_fuse_commit_deferred) -- WHILE stream M ray-marches
  // frame k; M then publishes it with one short launch (svoslam_svo_fuse_apply) and marches frame k+1.  The map stream
  // carries apply + grid / brick refresh + march instead of commit + refresh + march.  Round 2, beside the tree march (a
  // latency chain through the same L2 / HBM as the commit): the march took 0.32 ms instead of 0.28, the commit 0.27 instead of
  // 0.11, the period barely moved.  Round 3, beside the issue-bound brick march: cfg3 1862 -> 2055 frames/s; the period is
  // now apply(k) -> plan(k+1) -> this commit -> apply(k+1) (0.42 ms) level with the map stream (0.40).  Round 6 took the plan out of
  // that cycle (its splits' links written as pending links no march follows, the plans a chain of their own on S or on C): same pool,
  // same images, 2350-2500 frames/s either way -- the streams share one machine, the cycle was not what held the frame; not kept
  // (profiles/r06_plan_ahead_pending_links_experiment.txt).
  hipStream_t s_compute = r->s_map[1];
  auto enqueue_compute = [&](int i) -> int {
    SVO_HIP(hipStreamWaitEvent(s_compute, ev_plan[i], 0));
    // (the plan of frame i HERE, in order ahead of its commit and behind apply(i-1) -- one hand-off between streams less in the
    // cycle -- was measured: 2000-2200 frames/s against 2030-2390, the sorts then run unthrottled beside the march: not kept)
    mark(i, 7, s_compute);
    SVO_TRY(svoslam_svo_fuse_commit_deferred(r->ws[i % kRing], staged ? r->in_rgb[i % kRing] : d_rgbs[i], npts, r->depth, r->pool, s_compute));
    SVO_HIP(hipEventRecord(ev_commit[1][i], s_compute));
    return SVOSLAM_OK;
  };
  auto enqueue_apply = [&](int i) -> int {
    SVO_HIP(hipStreamWaitEvent(r->s_map[0], ev_commit[1][i], 0));
    SVO_TRY(svoslam_svo_fuse_apply(r->ws[i % kRing], r->pool, r->s_map[0]));
    SVO_HIP(hipEventRecord(ev_commit[0][i], r->s_map[0]));
    mark(i, 8, r->s_map[0]);
    return SVOSLAM_OK;
  };
  auto enqueue_all_deferred = [&]() -> int {
    SVO_TRY(enqueue_maps(0));
    SVO_TRY(enqueue_track(0));
    if (n > 1) SVO_TRY(enqueue_maps(1));
    SVO_TRY(enqueue_prepare(0));
    SVO_TRY(enqueue_compute(0));
    for (int i = 0; i < n; i++) {
      if (i + 1 < n) SVO_TRY(enqueue_track(i + 1));
      if (i + 2 < n) SVO_TRY(enqueue_maps(i + 2));
      // (see the other schedule.)  ev_commit[k] here is the DEFERRED commit of frame k, enqueued one iteration earlier than the
      // in-place commit is there: a lead of 1 is the same distance as 2 there.  Measured, lead 1 / 2 / 3 / 4: cfg3 100 frames
      // 2380 / 2350 / 2310 / 2035 frames/s, the driver's 20 frames 2085 / 2000 / 1860 / 1770 -- the further the host runs ahead,
      // the more often (lead 4: always) the streams settle in the slower of their two steady states.
      const int lead = r->lead < 0 ? 1 : r->lead;
      if (lead > 0 && i >= lead) SVO_HIP(hipEventSynchronize(ev_commit[0][i - lead]));
      SVO_TRY(enqueue_apply(i));
      if (i + 1 < n) {
        SVO_TRY(enqueue_prepare(i + 1));  // its plan waits for apply i
        SVO_TRY(enqueue_compute(i + 1));  // host order: BEFORE the march of frame i, whose grid refresh must leave this commit's marks alone
      }
      if (!march || march[i]) {
        uint8_t *img = d_images ? d_images[i] : ((i == n - 1) ? d_image : r->scratch_image[0]);
        SVO_TRY(svoslam_cone_trace_svo_band(img, r->w, r->h, row_first, rows, r->fov, views + 16 * (size_t)i, r->pool->d_data,
                                            r->center, r->edge, r->mode, d_steps, r->s_map[0]));
      }
      mark(i, 9, r->s_map[0]);
    }
    return SVOSLAM_OK;
  };
  auto enqueue_all = [&]() -> int {
    if (R == 1 && deferred) return enqueue_all_deferred();
    SVO_TRY(enqueue_maps(0));
    SVO_TRY(enqueue_track(0));
    if (n > 1) SVO_TRY(enqueue_maps(1));
    SVO_TRY(enqueue_prepare(0));
    for (int i = 0; i < n; i++) {
      if (i + 1 < n) SVO_TRY(enqueue_track(i + 1));
      if (i + 2 < n) SVO_TRY(enqueue_maps(i + 2));  // (one stream: behind track i+1, ahead of track i+2 -- two map sets ahead at most)
      const int a = i & (R - 1);  // the replica frame i is marched on: it gets commit i first
      // The host stays at most `lead` commits ahead of the device (default 2; svoslam_config.runner_lead = 0: as far as the
      // pool's size ring allows, 8).  Whatever has been enqueued when the host STOPS enqueuing drains at 0.6 ms per
      // frame instead of 0.32 (measured with HIP events per stage: the kernels themselves keep their durations and the
      // clock stays at 2.4 GHz, the gaps between them grow; AMD_DIRECT_DISPATCH=0 does not show it but costs 10 % in
      // steady state).  With a lead of 8 frames that tail was 8 of the 20 frames of a short call: 2130 -> 2580 frames/s.
      // (a rank of 8, lead 1 / 2: 4480 / 5030 frames/s)
      const int lead = r->lead < 0 ? 2 : r->lead;
      if (lead > 0 && i >= lead) SVO_HIP(hipEventSynchronize(ev_commit[a][i - lead]));
      SVO_TRY(enqueue_commit(i, a, R == 1));
      if (i + 1 < n && !chain) SVO_TRY(enqueue_prepare(i + 1));  // host order: after ev_commit[a][i] has been recorded
      // one march at a time: two of them (1200 workgroups) leave no CU for the tracker's and the fusion's workgroups
      if (R == 2 && serial_marches && i > 0) SVO_HIP(hipStreamWaitEvent(r->s_map[a], ev_ray[i - 1], 0));
      if (!march || march[i]) {
        uint8_t *img = d_images ? d_images[i] : ((i == n - 1) ? d_image : r->scratch_image[a]);
        SVO_TRY(svoslam_cone_trace_svo_band(img, r->w, r->h, row_first, rows, r->fov, views + 16 * (size_t)i, replica(r, a)->d_data,
                                            r->center, r->edge, r->mode, d_steps, r->s_map[a]));
        if (chain) { SVO_HIP(hipEventRecord(ev_ray[i], r->s_map[a])); last_march = i; }
      }
      if (R == 2) SVO_HIP(hipEventRecord(ev_ray[i], r->s_map[a]));
      mark(i, 9, r->s_map[a]);
      if (i + 1 < n && chain) SVO_TRY(enqueue_prepare(i + 1));  // host order: after the march's event, which its plan may have to wait for
      if (R == 2) SVO_TRY(enqueue_commit(i, a ^ 1, true));  // behind the march of frame i-1 on that replica
    }
    return SVOSLAM_OK;
  };
  const int rc = enqueue_all();
  // join, also after an error: whatever was enqueued is ordered before the caller's next work
  for (int k = 0; k < 5; k++) {
    if (hipEventRecord(r->ev_end[k], all[k]) == hipSuccess) (void)hipStreamWaitEvent(cur, r->ev_end[k], 0);
  }
  if (rc != SVOSLAM_OK) (void)hipDeviceSynchronize();  // leave nothing in flight behind a failed call
  return rc;
}

int svoslam_runner_run(svoslam_runner *r, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs, const long long *timestamps,
                       const float *views, int32_t n, uint8_t *d_image, int32_t row_first, int32_t rows,
                       unsigned long long *d_steps, void *caller_stream) {
  if (n > 0 && !d_image) return SVOSLAM_ERR_INVALID_ARG;
  return runner_run_impl(r, d_depths, d_rgbs, timestamps, views, n, d_image, row_first, rows, d_steps, caller_stream, nullptr, nullptr,
                         nullptr, nullptr);
}

int svoslam_runner_run_sharded(svoslam_runner *r, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs,
                               const long long *timestamps, const float *views, int32_t n, const float *const *d_deltas,
                               void *const *delta_events, const uint8_t *march, uint8_t *const *d_images, int32_t row_first,
                               int32_t rows, unsigned long long *d_steps, void *caller_stream) {
  if (n > 0 && (!d_deltas || !d_images)) return SVOSLAM_ERR_INVALID_ARG;
  return runner_run_impl(r, d_depths, d_rgbs, timestamps, views, n, nullptr, row_first, rows, d_steps, caller_stream, d_deltas,
                         delta_events, march, d_images);
}

int svoslam_runner_run_sharded_presorted(svoslam_runner *r, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs,
                                         const long long *timestamps, const float *views, int32_t n, const float *const *d_deltas,
                                         void *const *delta_events, const uint8_t *march, uint8_t *const *d_images,
                                         const unsigned long long *const *d_sorted_keys, const uint32_t *const *d_sorted_idx,
                                         void *const *sorted_events, int32_t row_first, int32_t rows, unsigned long long *d_steps,
                                         void *caller_stream) {
  if (n > 0 && (!d_deltas || !d_images || !d_sorted_keys || !d_sorted_idx)) return SVOSLAM_ERR_INVALID_ARG;
  return runner_run_impl(r, d_depths, d_rgbs, timestamps, views, n, nullptr, row_first, rows, d_steps, caller_stream, d_deltas,
                         delta_events, march, d_images, d_sorted_keys, d_sorted_idx, sorted_events);
}

// ---- frame-to-model tracking inside the native loop (VERDICT r04 missing 3) --------------------------------------
// The reference leaves it as a TODO (src/sensor/rgbd_camera.cpp:185: "ICP should not swap, as last_frame should be updated by a
// different function"); the entry points that ARE that function (svoslam_raycast_model_depth, svoslam_camera_set_model_depth,
// svoslam_camera_set_frame_to_model: include/svoslam.h) were driven from the Python pipeline only.  Here the whole frame runs
// inside the library: track (against the model set once one has been accepted) -> back-project + fuse -> ray-cast the map into a
// depth image from the pose just tracked -> accept it as the next frame's model if it covers at least min_coverage of the
// pixels (else the next frame falls back to the previous frame's maps) -> cone-traced view.  The model of frame k+1 is a function
// of the map AFTER fusion k, so nothing of frame k+1 but its maps could overlap frame k: the loop is sequential on the caller's
// stream, with ONE 4-byte readback per frame (the coverage count decides on the host which map set the next launch reads).
__global__ __launch_bounds__(256) void count_nonzero_u16_kernel(const uint16_t *__restrict__ d, int n, unsigned *__restrict__ out) {
  __shared__ unsigned s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  unsigned c = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) c += d[i] != 0 ? 1u : 0u;
  const unsigned long long m = __ballot(c != 0);
  (void)m;
  for (int o = 32; o > 0; o >>= 1) c += (unsigned)__shfl_down((int)c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(out, s_cnt);
}

int svoslam_runner_run_model(svoslam_runner *r, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs, const long long *timestamps,
                             const float *views, int32_t n, uint8_t *d_image, int32_t row_first, int32_t rows,
                             unsigned long long *d_steps, float min_coverage, int32_t *models_used, void *caller_stream) {
  if (models_used) *models_used = 0;
  if (!r || n < 0 || (n > 0 && (!d_depths || !d_rgbs || !timestamps || !views || !d_image))) return SVOSLAM_ERR_INVALID_ARG;
  if (row_first < 0 || rows < 0 || row_first + rows > r->h || !(min_coverage >= 0.0f)) return SVOSLAM_ERR_INVALID_ARG;
  if (r->replicas != 1) return SVOSLAM_ERR_INVALID_ARG;
  for (int i = 0; i < n; i++) if (!d_depths[i] || !d_rgbs[i] || (i > 0 && timestamps[i] <= timestamps[i - 1])) return SVOSLAM_ERR_INVALID_ARG;
  if (n == 0) return SVOSLAM_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(caller_stream);
  // the same call bookkeeping as svoslam_runner_run: one caller stream per runner, timestamps after the camera's latest
  if (r->ran && s != r->last_caller) return SVOSLAM_ERR_INVALID_ARG;
  {
    int32_t have = 0; long long latest = 0;
    SVO_TRY(svoslam_camera_latest_timestamp(r->cam, &have, &latest));
    if (have && timestamps[0] <= latest) return SVOSLAM_ERR_INVALID_ARG;
  }
  r->ran = true; r->last_caller = s;
  const size_t px = (size_t)r->w * r->h;
  const int npts = r->w * r->h;
  if (!r->model_depth) {
    SVO_HIP(hipMalloc((void **)&r->model_depth, px * 2));
    SVO_HIP(hipMalloc((void **)&r->model_count, 4));
  }
  SVO_TRY(svoslam_camera_set_frame_to_model(r->cam, 1));
  svoslam_workspace *ws = r->ws[0];
  int used_models = 0;
  for (int i = 0; i < n; i++) {
    int32_t used = 0;
    SVO_TRY(svoslam_camera_prepare(r->cam, d_depths[i], d_rgbs[i], timestamps[i], &used, s));
    if (!used) return SVOSLAM_ERR_INVALID_ARG;  // (a timestamp the camera has already seen)
    SVO_TRY(svoslam_camera_track(r->cam, s));
    const float *pose = svoslam_camera_fusion_transform_device(r->cam);
    if (r->fused_front) {  // main.cpp:39-44 + computeKeys in one launch, then sort / plan / commit (svoFromPointCloud)
      SVO_TRY(svoslam_svo_fuse_sort_frame(ws, d_depths[i], pose, r->w, r->h, r->fx, r->fy, r->depth, r->center, r->edge, r->bbox, s));
    } else {
      float *pts = r->points[0];
      SVO_TRY(svoslam_generate_vertex_map(d_depths[i], pts, r->w, r->h, r->fx, r->fy, r->w, r->h, s));
      SVO_TRY(svoslam_transform_vertex_map_dmat(pts, pose, npts, s));
      SVO_TRY(svoslam_point_cloud_bbox_device(ws, pts, npts, r->bbox, s));
      SVO_TRY(svoslam_svo_fuse_sort(ws, pts, npts, r->depth, r->center, r->edge, s));
    }
    SVO_TRY(svoslam_svo_fuse_plan(ws, npts, r->depth, r->pool, s));
    SVO_TRY(svoslam_svo_fuse_commit(ws, d_rgbs[i], npts, r->depth, r->pool, s));
    // the map as the sensor would see it from the pose just tracked -> the maps the NEXT frame is tracked against
    SVO_TRY(svoslam_raycast_model_depth(r->model_depth, r->w, r->h, r->fx, r->fy, nullptr, pose, r->pool->d_data, r->center, r->edge, nullptr, s));
    SVO_HIP(hipMemsetAsync(r->model_count, 0, 4, s));
    count_nonzero_u16_kernel<<<256, 256, 0, s>>>(r->model_depth, npts, r->model_count);
    SVO_LAUNCH_CHECK();
    unsigned covered = 0;
    SVO_HIP(hipMemcpyAsync(&covered, r->model_count, 4, hipMemcpyDeviceToHost, s));
    SVO_HIP(hipStreamSynchronize(s));
    if ((double)covered >= (double)min_coverage * (double)px) {
      SVO_TRY(svoslam_camera_set_model_depth(r->cam, r->model_depth, s));
      used_models++;
    } else {
      SVO_TRY(svoslam_camera_set_model_depth(r->cam, nullptr, s));
    }
    SVO_TRY(svoslam_cone_trace_svo_band(i == n - 1 ? d_image : r->scratch_image[0], r->w, r->h, row_first, rows, r->fov, views + 16 * (size_t)i,
                                        r->pool->d_data, r->center, r->edge, r->mode, d_steps, s));
  }
  if (models_used) *models_used = used_models;
  // (the camera leaves with frame-to-model tracking ON and the last accepted model set: a second call continues the sequence exactly
  // as one longer call would.  svoslam_runner_run -- the loop without a model refresh -- clears both when it starts: ADVICE r05)
  r->model_pending = true;
  return SVOSLAM_OK;
}

// computePointCloudBoundingBox of the last frame enqueued (main.cpp:43): {min xyz, max xyz, any}.  Blocking.
int svoslam_runner_bbox(svoslam_runner *r, float h_bbox7[7]) {
  if (!r || !h_bbox7) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());
  SVO_HIP(hipMemcpy(h_bbox7, r->bbox, 7 * 4, hipMemcpyDeviceToHost));
  return SVOSLAM_OK;
}

// diagnostic: milliseconds of the stage marks of the last call relative to its first mark, h_ms[frames][10] =
// {maps begin, maps end, track begin, pose, prepare begin, plan begin, plan end, commit begin, commit end, march end}
// (-1 where unavailable); needs svoslam_config.runner_timeline = 1 at creation.  Blocking.
int svoslam_runner_timeline(svoslam_runner *r, float *h_ms, int32_t max_frames, int32_t *frames) {
  if (!r || !h_ms || !frames) return SVOSLAM_ERR_INVALID_ARG;
  *frames = 0;
  if (!r->timeline || r->tl_frames == 0) return SVOSLAM_OK;
  SVO_HIP(hipDeviceSynchronize());
  const int n = r->tl_frames < max_frames ? r->tl_frames : max_frames;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < kTlStages; k++) {
      float ms = -1.0f;
      if (hipEventElapsedTime(&ms, r->tl_events[0], r->tl_events[(size_t)i * kTlStages + k]) != hipSuccess) { (void)hipGetLastError(); ms = -1.0f; }
      h_ms[(size_t)i * kTlStages + k] = ms;
    }
  *frames = n;
  return SVOSLAM_OK;
}

}  // extern "C"
