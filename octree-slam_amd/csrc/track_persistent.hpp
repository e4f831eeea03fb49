// track_persistent.hpp -- see track_persistent.hip
#pragma once
#include "common.hpp"

namespace svoslam {

struct CamState;

constexpr int kTrkThreads = 512;     // 8 wavefronts, 2 per SIMD: up to 256 VGPRs for the register-resident pixels
constexpr int kTrkSlots = 4;  // pixels of a level a lane may keep in registers (x 12 floats)
constexpr int kTrkMaxWorkers = 247;  // + the solver workgroup <= one per CU on an idle device

// words shared between the workgroups of one launch; zeroed (with the fan-in accumulators, two banks of which the solver
// clears the idle one) once when the camera is created / reset
struct TrackSync {
  unsigned long long granule[64];  // {tag = generation * 32 + epoch, value}: this_trans, update_trans, flags
  unsigned gen;                    // launches so far (device-resident: the launch is replay-safe)
  unsigned fail;                   // give-up code of a bounded spin (0 = none)
  unsigned pad[2];
  unsigned long long prof[32][8];  // SVO_TRK_PROF builds: s_memtime stamps per epoch (solver 0..3, worker 0 4..7)
};

struct TrackLevel { const float *lv, *ln, *cv, *cn; int first, end; };
struct TrackArgs {
  TrackLevel level[3];
  int iters[3];
  int workers;           // workgroups 1..workers; workgroup 0 solves
  int participants[3];   // workers taking part at each level (a prefix)
  int slots[3];          // pixels per lane at each level
  float *work_v = nullptr, *work_n = nullptr;  // streaming levels: the current frame's maps as transformed so far (finest-level size)
  int variant = 0;       // 0: register-resident form <kTrkSlots, 2>; 1: streaming form for large images
  int corrected = 0;     // the corrected tracker (icp_device.hpp icp_rot_rows)
};
constexpr int kTrkStreamSlots = 2;
constexpr int kTrkStreamMinWaves = 3;  // (round 6, beside the 80-VGPR march: 2 = 192 VGPRs without spills -- the tracker alone 0.53 ms instead of 0.60, cfg4 865-878 frames/s
// instead of 964-971: the sort's and the commit's workgroups find no room beside it; 4 = 128 VGPRs, 56 spilled: 921-926.  profiles/r06_tracker_diet_ab.txt)

int track_persistent_capacity(hipStream_t s, int *max_workgroups, int variant = 0);
int track_persistent_plan_stream(TrackArgs &A, int capacity);  // large images: coarsest level in registers, finer levels streamed through work maps
int track_persistent_plan(TrackArgs &A, int capacity);
int track_persistent_plan_coarse(TrackArgs &A, int capacity, int coarse_levels);  // only the coarsest 1 or 2 levels; the rest: launch chain
int track_persistent_profile(const TrackSync *d_sync, unsigned long long *out, hipStream_t s);
size_t track_persistent_ticket_bytes();  // fan-in accumulators: [2 banks][32 epochs][8][27 of 64] 64-bit words, zeroed with TrackSync
int track_persistent_launch(CamState *st, TrackSync *sy, unsigned *tickets, const TrackArgs &A, hipStream_t s);

}  // namespace svoslam
