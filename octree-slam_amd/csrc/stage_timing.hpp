// stage_timing.hpp -- optional HIP-event timing of the hot path's stages (bench.py's per-stage rooflines).
//
// A stage's kernels are bracketed by a pair of events recorded on the stream they are launched on (torch.cuda.Event
// would see torch's current stream only).  Off unless svoslam_stage_timing(mask) turns a stage on: an event record costs
// ~2.6 us of its stream's time (hardware lesson 6), so the headline run enables only the two candidates for
// "dominant kernel" (march on the map stream, tracker on its own stream) and measures the fusion's launches in a short
// sequential pass after the timed region.
#pragma once
#include "common.hpp"

namespace svoslam {

enum Stage {
  kStageMarch = SVOSLAM_STAGE_MARCH,          // cone_trace_kernel alone (not the grid refresh in front of it)
  kStageTracker = SVOSLAM_STAGE_TRACKER,      // track_persistent_kernel, or the launch chain of one frame
  kStageFuseSort = SVOSLAM_STAGE_FUSE_SORT,   // keys (fused front end) + radix sort
  kStageFusePlan = SVOSLAM_STAGE_FUSE_PLAN,   // plan_count + plan_scan_finish + plan_emit (+ early split_all)
  kStageFuseCommit = SVOSLAM_STAGE_FUSE_COMMIT,  // (split_all +) leaf blend / mip kernel + straddlers
  kStageMaps = SVOSLAM_STAGE_MAPS,            // bilateral + vertex / normal pyramids
  kStageMeshRaster = SVOSLAM_STAGE_MESH_RASTER,  // mesh.hip: tri_scanline_count + scans + scanline_kernel<false / true>
  kStageMeshSort = SVOSLAM_STAGE_MESH_SORT,      // mesh.hip: radix sort of the fragments
  kStageMeshEmit = SVOSLAM_STAGE_MESH_EMIT,      // mesh.hip: voxel_flag + scan + voxel_emit
  kStageCount = SVOSLAM_STAGE_COUNT
};

unsigned stage_timing_mask();                                  // stages that are on
int stage_timing(unsigned mask);                               // bit s = stage s on; clears the log of every stage
int stage_timing_read(int stage, float *ms_sum, int *pairs);   // blocking; resets that stage's log
// A bracket is ONE entry of its stage's log: stage_begin reserves the entry's two events under the lock and records the
// first, stage_end records the second into the SAME entry -- brackets of several streams / host threads that are open at the
// same time cannot mis-pair, a change of the mask or a read between the two halves drops the entry, and so does
// a failed record.  token < 0: the stage is off (stage_end is then a no-op).
int stage_begin(int stage, hipStream_t stream, long long *token);
int stage_end(int stage, long long token, hipStream_t stream);

struct StageScope {  // brackets a scope; tolerant of early returns
  int stage; hipStream_t s; long long token = -1;
  StageScope(int st, hipStream_t stream) : stage(st), s(stream) { (void)stage_begin(stage, s, &token); }
  ~StageScope() { (void)stage_end(stage, token, s); }
};

}  // namespace svoslam
