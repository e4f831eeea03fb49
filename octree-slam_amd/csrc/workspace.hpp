// workspace.hpp -- grow-only device scratch buffers reused across calls.
// The reference cudaMallocs and frees 6+D temporaries per fused frame
// (svo.cu:184-188,221,594,652,713); here every temporary lives in one of these
// slots and is reallocated only when a call needs more than it has.
#pragma once

#include "common.hpp"
#include "graph_cache.hpp"

namespace svoslam {

struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return SVOSLAM_OK;
    if (ptr) { SVO_HIP(hipFree(ptr)); ptr = nullptr; bytes = 0; }
    size_t want = need + need / 4 + 256;  // 25 % headroom: frames vary slightly in size
    SVO_HIP(hipMalloc(&ptr, want));
    bytes = want;
    return SVOSLAM_OK;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr; bytes = 0;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(ptr); }
};

// Host-visible counters written by the device and read after ONE stream sync.
struct PlanCounts {
  int32_t total_records;                       // nodes to split this call
  int32_t pass_start[SVOSLAM_MAX_DEPTH + 2];   // record range of pass p = [pass_start[p], pass_start[p+1])
  int32_t any_valid;                           // at least one finite point
};

}  // namespace svoslam

struct svoslam_workspace {
  svoslam::DeviceBuffer keys_a, keys_b, vals_a, vals_b;  // radix sort ping-pong
  svoslam::DeviceBuffer tile_hist;                        // [256][tiles] digit / bucket histograms
  svoslam::DeviceBuffer small;                            // totals, bucket bases, counters
  svoslam::DeviceBuffer leaf_t, leaf_f;                   // per sorted key: first unsplit depth, frontier node
  svoslam::DeviceBuffer rec_key, rec_front;               // split records in reference order
  svoslam::DeviceBuffer rec_pass;                         // async path: pass of each record
  svoslam::DeviceBuffer leaf_rec0;                        // async path: per sorted key, rank of the pass-0 record it owns (early split)
  svoslam::DeviceBuffer leaf_start;                       // async path: per sorted key, child tile where the commit's walk resumes the plan's
  svoslam::DeviceBuffer path_nodes;                       // [(D-1)][n] node index per owned depth (mip lists)
  svoslam::DeviceBuffer strad;                            // [D][tiles][2] nodes whose leaf run crosses a workgroup (async commit)
  svoslam::DeviceBuffer strad_b;                          // the same for the commit of the plan to a second replica of the pool
  svoslam::DeviceBuffer kr_keys, kr_idx, kr_small;        // key-range sharded commit: the rank's slice of the sorted arrays; window, numbering table, scalars
  const void *keyrange_pool = nullptr;                    // ... svo_fuse_keyrange_commit has run for this pool, svo_fuse_keyrange_apply is due
  svoslam::DeviceBuffer apply_nodes;                      // deferred commit: per fill tile, the nodes its workgroup wrote (dense from the tile's start; counts behind the lists)
  // deferred commit waiting for svo_fuse_apply (what the apply launch needs)
  const void *deferred_pool = nullptr;
  int deferred_n = 0, deferred_depth = 0, deferred_tiles = 0;
  svoslam::DeviceBuffer bfs_a, bfs_b, bfs_mask, bfs_ptr;  // extraction
  svoslam::DeviceBuffer misc;                             // bbox partials etc.
  svoslam::DeviceBuffer scan_tmp;                         // chunk sums of exclusive_scan_u32
  svoslam::DeviceBuffer frame_bbox;                       // fused fusion front end: arrival ticket (word 0) + workgroup bounding boxes
  svoslam::PlanCounts *h_counts = nullptr;                // pinned host
  long long mesh_fragments = 0;                           // (cell, triangle) fragments of the last mesh_to_voxel_grid (svoslam_mesh_last_fragments)
  // phased fusion (svo_fuse_sort -> plan -> commit): where the sort left its output, what has been planned
  const unsigned long long *sorted_keys = nullptr;
  const unsigned int *sorted_idx = nullptr;
  int planned_n = -1;
  const void *early_split_pool = nullptr;                  // svo_fuse_split_early has initialised the planned splits' tiles in this pool
  bool structure_planned = false;                          // ... by svo_fuse_plan_structure (its reservation is released by its commit)
  long long keyrange_bound = 0;                            // key-range commit: the plan's reservation, released by svo_fuse_keyrange_apply's size readback
  const void *planned_pool = nullptr;                      // the pool svo_fuse_plan read (its reservation is already booked)
  svoslam::GraphCache g_sort, g_plan, g_commit;            // recorded launch sequences of the three phases
  // `small` (4 KB of totals / bases / counters) is zeroed when it is created: the planner's any_valid word and arrival
  // ticket must start at zero (every plan leaves them at zero).  Blocking, once per workspace.
  int reserve_small() {
    if (small.bytes >= 4096) return SVOSLAM_OK;
    SVO_TRY(small.reserve(4096));
    SVO_HIP(svoslam::memset_sync(small.ptr, 0, small.bytes));
    return SVOSLAM_OK;
  }
  // every buffer address the recorded phases bake in (a reallocation makes a new key)
  unsigned long long layout_hash() const {
    const void *p[] = {keys_a.ptr, keys_b.ptr, vals_a.ptr, vals_b.ptr, tile_hist.ptr, small.ptr, leaf_t.ptr, leaf_f.ptr,
                       rec_key.ptr, rec_front.ptr, rec_pass.ptr, path_nodes.ptr, strad.ptr, strad_b.ptr, apply_nodes.ptr, leaf_rec0.ptr, leaf_start.ptr};
    unsigned long long h = 1469598103934665603ull;
    for (const void *q : p) h = (h ^ (unsigned long long)(uintptr_t)q) * 1099511628211ull;
    return h;
  }
  void release_all() {
    keys_a.release(); keys_b.release(); vals_a.release(); vals_b.release(); tile_hist.release(); small.release();
    leaf_t.release(); leaf_f.release(); rec_key.release(); rec_front.release(); path_nodes.release(); strad.release(); strad_b.release();
    rec_pass.release(); apply_nodes.release(); leaf_rec0.release(); leaf_start.release();
    bfs_a.release(); bfs_b.release(); bfs_mask.release(); bfs_ptr.release(); misc.release(); scan_tmp.release(); frame_bbox.release();
    if (h_counts) { (void)hipHostFree(h_counts); h_counts = nullptr; }
    g_sort.clear(); g_plan.clear(); g_commit.clear();
  }
};
