// pool_grid.hpp -- the acceleration data of the ray march as PROPERTIES OF THE POOL, maintained incrementally: the level-8
// grid (round 2) and the occupancy bricks below it (round 3; further down in this file).
//
// Round 1 rebuilt a dense level-7 grid (16.8 MB, 9-12 us) from the pool at the start of every render and used the
// level-8 one (134 MB, 30 us to rebuild, 14 % less march time) only for megapixel renders.  A fusion changes the
// outcome of the walk over levels 1..8 only below the level-5 prefixes of the keys it inserted (a few hundred 25 cm
// blocks of 8^3 cells per frame at 640x480), so the grid now belongs to the pool: every commit marks the level-5
// blocks it touches in a 4 KB bitmap, and the next render rebuilds just those blocks (8 cells per lane, one walk
// each) before it marches -- level 8 for every render, no per-render rebuild.
//
// What a commit can change, and why block marking covers it (G = 8, B = 5):
//  * colour words of nodes on the path of an inserted key (leaf blend, mip averages): entries hold the colour of the
//    node a walk STOPS on (first childless node, or the level-G node), all of them descendants of the key's level-B
//    prefix -- or nodes above level B that have children, which no entry refers to;
//  * a split gives a childless node at level l eight children: the sibling tiles lie inside the key's level-B block
//    when l >= B; a split at l < B re-labels the whole cube of the split node, so all 8^(B-l) blocks under it are
//    marked (first frames of a map only).
// Every other way of changing a pool (blocking fusion, voxel grids, load / set_nodes / copy / reset / expand)
// invalidates the grid as a whole; it is rebuilt in full by the next render.  Memory written behind the library's
// back (hipMemcpy into svoslam_pool.d_data) must be followed by svoslam_pool_touch().
#pragma once
#include <memory>

#include "common.hpp"
#include "workspace.hpp"

namespace svoslam {

constexpr int kPoolGridLevel = 8;                          // (2^8)^3 cells x 8 B = 134 MB per pool that is rendered
// The grid PYRAMID (round 6): behind the level-G grid, the same kind of entry for the cells of levels 1 .. G - 1 (8 + 64 + ... + 8^(G-1)
// = 2.4 M entries, 19 MB): a sample whose cone LOD is COARSER than the grid level -- a ray far from a small root cube: every view of a
// mesh from outside (configs 2 and 5) -- ends the reference's walk at the LOD level on a node WITH children, whose alpha only the tree
// knew: LOD dependent loads from the root per step (cfg2's side view: 1.9 us per step of its longest rays).  Now ONE load:
//   entry of a level-l cell = (first childless node on the path at level st <= l: st, its colour word) or
//                             (flag | children tile of the level-l node, the level-l node's colour word).
// Kept current with the grid: every dirty level-5 block rewrites its cells of levels 6 .. G - 1, its own and its four ancestors'
// entries (the colour word of a node with children changes with anything below it; several blocks write an ancestor the same value).
__host__ __device__ constexpr uint32_t pyr_offset(int l) { return ((1u << (3 * l)) - 8u) / 7u; }   // entries of levels 1 .. l - 1
__host__ __device__ constexpr uint32_t pyr_entries(int g) { return pyr_offset(g); }                  // levels 1 .. g - 1
// the same for a run-time level 1 <= l <= 10: bits 3, 6, ..., 3 (l - 1) of 0b...001001001000
__device__ __forceinline__ uint32_t pyr_offset_rt(int l) { return 0x09249248u & ((1u << (3 * l)) - 1u); }
constexpr int kPoolGridBlockLevel = 5;                     // dirty tracking granularity: (2^5)^3 = 32768 blocks of 8^3 cells
constexpr int kPoolGridBlocks = 1 << (3 * kPoolGridBlockLevel);
constexpr int kPoolGridDirtyWords = kPoolGridBlocks / 32;  // 4 KB bitmap

struct PoolAccel {
  DeviceBuffer grid;          // uint2[2^(3 G)], empty until the pool is rendered for the first time
  // device, two of them: bitmap over the level-B blocks [kPoolGridDirtyWords], then the compacted list of the marked
  // blocks [kPoolGridBlocks] and its length [1] (written at the end of every commit).  Direct commits mark state 0, a
  // deferred commit the state of its epoch's parity -- it runs next to the render of the previous frame, whose
  // refresh must neither see nor clear its marks: a refresh leaves the state of a pending deferred commit alone.
  uint32_t *d_dirty[2] = {nullptr, nullptr};
  bool valid = false;         // false: rebuild everything at the next render
  // deferred commits (svo_build.hip, svo_fuse_commit_deferred / svo_fuse_apply): colour words a commit computes while the
  // previous frame is still being ray-marched; entry = epoch << 32 | word, valid for the commit whose epoch it carries
  DeviceBuffer shadow;
  size_t shadow_nodes = 0;
  uint32_t epoch = 0;
  bool deferred_pending = false;
  // The grid is shared by every stream that renders the pool, and `valid` flips when a build is ENQUEUED: a render on
  // another stream must not march through (or update) a grid the previous stream may still be building.
  // last_stream = the stream of the latest refresh; a refresh on a different stream first waits for everything enqueued
  // on that one so far (an event recorded there at that moment: no per-frame cost while one stream renders the pool).
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_order = nullptr;
  // occupancy bricks (below): the dense field and the bitmap of the 64 KB groups that hold anything.  Allocated by the
  // first reference-mode render of the pool; bricks_valid = false: rebuild all.
  uint16_t *bricks = nullptr;
  uint32_t *d_brick_touched = nullptr;  // [kBrickGroupWords]
  bool bricks_valid = false;
  unsigned brick_served[2] = {0u, 0u};  // how often each dirty state's ring has been served (its parity picks the mark: kBrickMarkOffset)
  bool bricks_failed = false;           // the field could not be allocated: this pool is marched through the tree
  // Every colour word of the pool was computed by this library's fusion from an empty pool (any path: blocking, phased,
  // deferred): then a node with children carries the MAXIMUM of its children's alphas (averageChildren, svo.cu:384-447,
  // re-run for every ancestor of a touched leaf), so a level-11 node with A < 254 has no saturated child and the brick
  // rebuild need not read its level-12 tile (it then reports "a level-12 node may have children": an LOD beyond 12 walks the
  // tree).  On the 300-frame cfg3 map that is every one of the 1.5 M tiles a refresh used to read (no 2 mm leaf collects
  // its 127 observations).  Foreign words (set_nodes / load / copy / paging / touch) clear the flag until the next reset.
  bool mip_consistent = true;
  // deepest commit so far, and the brick shape that follows from it (brick_shift_for_depth): bricks describe levels 9 + s .. 12 + s,
  // a sample whose LOD reaches below a level-(12 + s) node with children walks the tree.  Pools fused to depth <= 12 (640x480
  // at a 4 m half edge: LODs 9..12) take s = 0, depth 13 / 14 (1920x1080 into a depth-14 SVO: LOD 13 between one and two
  // metres) s = 1; deeper pools get no bricks.  brick_shift = the shape the field currently holds (-1: none built yet).
  int max_depth = 0;
  int brick_shift = -1;
  ~PoolAccel();  // device buffers live as long as the last holder of the entry (std::shared_ptr)
};

// registry (guarded by a mutex inside), keyed by the address of the node memory: pools are known from pool_init
// (or their first growth) to pool_free
void pool_accel_register(svoslam_pool *pool);
void pool_accel_rebind(const uint32_t *old_data, const uint32_t *new_data);  // the nodes moved to a larger allocation
void pool_accel_unregister(svoslam_pool *pool);
// depth > 0: of the fusion that changed the pool; < 0: the pool is empty again.  foreign_words: the nodes now hold words this
// library's fusion did not compute (set_nodes / load / copy / paging / svoslam_pool_touch): see PoolAccel::mip_consistent
void pool_accel_invalidate(svoslam_pool *pool, int depth = 0, bool foreign_words = true);
uint32_t *pool_accel_dirty_bitmap(svoslam_pool *pool, int parity, int commit_depth = 0, int *brick_shift = nullptr);  // nullptr for memory that is not a registered pool;
// commit_depth: the depth of the commit that is about to mark (the pool remembers the deepest one: see PoolAccel::max_depth);
// *brick_shift: the brick shape the commit lists stale bricks in (-1: the pool gets no bricks)
std::shared_ptr<PoolAccel> pool_accel_find(const uint32_t *d_data);  // the registered pool whose nodes start at d_data, or null;
// the caller holds the entry for the duration of its enqueue (a pool_free / growth on another host thread cannot pull it away)
void pool_accel_forget_stream(hipStream_t stream);       // the stream is about to be destroyed

// shadow words of the pool (allocated and zeroed on first use / growth), the epoch of the commit that starts now
int pool_shadow_begin(svoslam_pool *pool, hipStream_t stream, unsigned long long **d_shadow, uint32_t *epoch);
int pool_shadow_current(svoslam_pool *pool, unsigned long long **d_shadow, uint32_t *epoch);  // of the pending deferred commit
void pool_shadow_end(svoslam_pool *pool);
bool pool_shadow_pending(svoslam_pool *pool);

// enqueue on `stream`: bring the grid of `pa` up to date with its pool (full build or dirty blocks only); returns the grid
// want_bricks: also bring the occupancy bricks up to date (allocating them on first use); *d_bricks = the field, or
// nullptr when the pool has none (not wanted so far, svoslam_config.march_bricks = 0, or no memory for them)
int pool_accel_refresh(PoolAccel *pa, const uint32_t *d_octree, hipStream_t stream, const uint2 **d_grid, bool want_bricks,
                       const uint16_t **d_bricks, int *brick_shift, uint32_t *tile_cost = nullptr, uint32_t *tile_order = nullptr,
                       int n_tiles = 0, bool *order_done = nullptr);
// tile_cost / tile_order / n_tiles (optional): the caller's march takes its tiles costliest-first (cone_trace.hip TraceParams); the
// incremental refresh launch brings the order up to date in one extra workgroup (*order_done = true), other forms leave it to the caller  // *brick_shift: the shape of *d_bricks (pool_grid.hpp "Shapes")

constexpr int kPoolGridListOffset = kPoolGridDirtyWords;                     // words
constexpr int kPoolGridCountOffset = kPoolGridDirtyWords + kPoolGridBlocks;  // words
// behind them, the ring of the level-9 nodes whose occupancy brick is stale (see "occupancy bricks"): the number of entries
// appended so far (commits), and two marks "served up to here" that the refreshes write in turn -- refresh n of a state reads
// mark[(n + 1) & 1] (what refresh n - 1 served) and writes mark[n & 1]: no launch has to wait for its last workgroup to reset
// anything, and the grid's update and the bricks' rebuild can share ONE launch -- then the entries
constexpr int kBrickCountOffset = kPoolGridCountOffset + 4, kBrickMarkOffset = kBrickCountOffset + 1;
constexpr int kBrickListOffset = kBrickCountOffset + 4;
constexpr int kBrickListCap = 1 << 20;  // more than this pending = "rebuild every brick" (a commit appends <= its distinct level-9 prefixes)
// ... and one bit per level-9 node: "in this state's ring" (set by whoever appends it, cleared by the rebuild that serves it), so
// that a brick is listed ONCE however many commits touch it before the next render -- a rank of a frame-sharded session fuses
// N frames per march, and the bricks of consecutive frames are mostly the same ones (16 MB per state)
constexpr int kBrickBitsOffset = kBrickListOffset + kBrickListCap;
constexpr int kBrickBitsWords = 1 << (3 * 9 - 5);
// ... and a second, small ring of the same kind (count, two marks, entries): the listed bricks whose commit created nodes at or
// above the brick node's level on the key's path -- the refresh writes the lines of the childless siblings those splits created
// (pool_grid.hip, brick_siblings).  Lapping it loses nothing but speed: a sibling without its lines is marched through the tree.
constexpr int kSibCountOffset = kBrickBitsOffset + kBrickBitsWords, kSibMarkOffset = kSibCountOffset + 1;
constexpr int kSibListOffset = kSibCountOffset + 4;
constexpr int kSibListCap = 1 << 16;
constexpr int kPoolGridStateWords = kSibListOffset + kSibListCap;

// ---- occupancy bricks (round 3; north_star's "4^3 bricks") -------------------------------------------------
// In SVOSLAM_RENDER_REFERENCE mode a sample of the march needs two facts about the node the reference's walk ends on
// (cone_tracing_kernels.cu:76-105): its level (the step is size / 2^level, :126) and whether its alpha saturates the ray
// (A - 127 >= 127, :108-119) -- the colour matters on the retiring step only (Q9).  Below the level-8 grid the walk is a
// chain of dependent 8-byte loads from a multi-GB pool; the long rays of a mature map spend nearly all their steps there.
// A brick holds those two facts for the 4^3 level-11 cells of ONE level-9 node as 64 x 16 bits = one 128-byte line:
//   bits 0-2  where the path through the cell stops: 0 = the level-8 node has no children (ask the level grid),
//             1 / 2 / 3 = first childless node at level 9 / 10 / 11, 4 = the level-11 node has children
//   bit  3    a level-12 node below this cell has children itself (an LOD deeper than 12 takes the tree walk)
//   bits 4-6  A >= 254 of the path's nodes at levels 9 / 10 / 11
//   bits 8-15 A >= 254 of the eight level-12 children (octant order)
// so a step whose LOD lies in 9..12 costs ONE 2-byte load, whatever the depth of the tree.  Bricks are addressed, not
// allocated: brick (x9, y9, z9) lives at a fixed place of a dense 2 x 2048^3-byte field (16 GiB of the 288 GB; bricks of
// one level-6 cube are 64 KB contiguous), zero = "ask the level grid", so nothing is built for space the map never
// reaches.  Upkeep rides on the level grid's: the leaf kernel of a commit appends the level-9 prefix of every run of keys
// to a list (each change of a commit lies on the path of one of its keys), the refresh of the level grid adds the eight
// children of level-8 nodes that have just been split, and one wavefront per listed node rewrites its line before the
// march.  Anything that invalidates the level grid clears the touched 64 KB groups and rebuilds every brick.
// Shapes (round 4): shift s = 0 is the shape above; s = 1 moves every level down by one (bricks of level-10 nodes, level-12
// cells, bits for the level-13 children) for pools fused to depth 13 / 14, and adds what the gap between the level-8 grid
// and the brick node needs: stop code 5 = "the level-9 node on the path is childless", and bit 7 = A >= 254 of the level-(11 + s)
// = cell-level node for s = 1 -- the saturation bits of the path are bits 4.. for levels 9.., so `bit = level - 5` in both
// shapes.  The field stays 2 x 2048^3 bytes: for s = 1 it is a WINDOW of 2048^3 level-12 cells in the middle of the root cube
// (half its edge: 8.2 m of the 16.4 m root of BASELINE config 4); a sample outside the window has no brick entry and takes the
// level grid / the tree walk as before.  Ring entries and the dedupe bitmap hold window-relative brick coordinates (9 bits per
// axis) in every shape.
constexpr int kBrickGroupLevel = 6;  // (of the window: 64 KB groups of 8^3 bricks)
constexpr int kBrickWindowBits = 11;
constexpr uint32_t kBrickWindowCells = 1u << kBrickWindowBits;
constexpr int kBrickMaxShift = 1;
__host__ __device__ constexpr int brick_node_level(int s) { return 9 + s; }
__host__ __device__ constexpr int brick_cell_level(int s) { return 11 + s; }
__host__ __device__ constexpr int brick_bits_level(int s) { return 12 + s; }
// first cell of the window on every axis, in cells of level 11 + s (a multiple of 32: whole groups)
__host__ __device__ constexpr uint32_t brick_window_origin(int s) { return ((1u << (11 + s)) - kBrickWindowCells) >> 1; }
// the shape for a pool whose deepest fusion was `depth` levels (-1: no bricks)
// (round 5, measured and not kept: shape 1 for depth-15 / 16 pools as well -- bit-exact, test_gpu_bricks.py passes at depth 15 / 16,
// but config 5's 4K renders of a depth-16 SVO got SLOWER from outside the model, 0.113 -> 0.167 and 0.214 -> 0.262 ms, and 6 %
// faster from inside it, 0.566 -> 0.535: half the colonnade lies outside the window and far samples stop above the bricks' levels)
inline int brick_shift_for_depth(int depth) { return depth <= 12 ? 0 : (depth <= 14 ? 1 : -1); }
constexpr size_t kBrickFieldEntries = (size_t)1 << (3 * kBrickWindowBits);
constexpr size_t kBrickFieldBytes = 2 * kBrickFieldEntries;
constexpr int kBrickGroups = 1 << (3 * kBrickGroupLevel);
constexpr int kBrickGroupWords = kBrickGroups / 32;  // "touched" bitmap: 32 KB
constexpr size_t kBrickGroupBytes = kBrickFieldBytes / kBrickGroups;  // 64 KB

#ifdef __HIPCC__
// called by ALL threads of ONE workgroup of `threads` (a multiple of 64, <= 1024) lanes after the commit's marks are
// complete: list of the marked blocks (bits stay set until a render has rebuilt the block) behind the bitmap
__device__ inline void pool_grid_compact(uint32_t *dirty, int threads) {
  __shared__ uint32_t wave_total[16];
  __shared__ uint32_t running;
  uint32_t *list = dirty + kPoolGridListOffset;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = threads >> 6;
  if (tid == 0) running = 0u;
  __syncthreads();
  for (int base = 0; base < kPoolGridDirtyWords; base += threads) {
    const int w = base + tid;
    const uint32_t bits = w < kPoolGridDirtyWords ? dirty[w] : 0u;
    const uint32_t cnt = (uint32_t)__popc(bits);
    uint32_t incl = cnt;  // inclusive scan over the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t off = running;
    for (int k = 0; k < wave; k++) off += wave_total[k];
    uint32_t pos = off + incl - cnt, b = bits;
    while (b) {
      const int bit = __ffs((int)b) - 1;
      b &= b - 1u;
      list[pos++] = (uint32_t)w * 32u + (uint32_t)bit;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t t = running;
      for (int k = 0; k < nwaves; k++) t += wave_total[k];
      running = t;
    }
    __syncthreads();
  }
  if (tid == 0) dirty[kPoolGridCountOffset] = running;
}

// mark the level-B blocks under the node whose key prefix has `levels` octant triplets (key = 1 d1 d2 ... in base 8,
// first level in the highest digit); levels >= B marks exactly one block
__device__ inline void pool_grid_mark(uint32_t *dirty, unsigned long long key, int levels) {
  constexpr int B = kPoolGridBlockLevel;
  const int use = levels < B ? levels : B;
  uint32_t x = 0, y = 0, z = 0;
  for (int k = 1; k <= use; k++) {
    const uint32_t oct = (uint32_t)(key >> (3 * (levels - k))) & 7u;
    x = (x << 1) | (oct & 1u); y = (y << 1) | ((oct >> 1) & 1u); z = (z << 1) | (oct >> 2);
  }
  const int sh = B - use;  // free low bits per axis
  const uint32_t n = 1u << sh;
  for (uint32_t dz = 0; dz < n; dz++)
    for (uint32_t dy = 0; dy < n; dy++)
      for (uint32_t dx = 0; dx < n; dx++) {
        const uint32_t b = ((((z << sh) | dz) << (2 * B)) | (((y << sh) | dy) << B)) | ((x << sh) | dx);
        atomicOr(&dirty[b >> 5], 1u << (b & 31u));
      }
}

// entry index of window cell (x, y, z) (11 bits each, relative to brick_window_origin) in the brick field: [z>>5 | y>>5 | x>>5] group (level 6, linear),
// [z y x bits 4..2] brick in the group, [z y x bits 1..0] cell in the brick
__host__ __device__ inline unsigned long long brick_entry_index(uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t lo = ((y >> 5) << 21) | ((x >> 5) << 15) | (((z >> 2) & 7u) << 12) | (((y >> 2) & 7u) << 9) | (((x >> 2) & 7u) << 6) |
                      ((z & 3u) << 4) | ((y & 3u) << 2) | (x & 3u);
  return ((unsigned long long)(z >> 5) << 27) | lo;
}

// One workgroup: order[0 .. n) = the tiles of a render, costliest first by cost[] (wavefront-steps of the previous render of that
// geometry), and cost[] cleared for the render that follows.  Longest-processing-time-first: the in-order dispatcher then ends a
// render with its cheapest tiles instead of whatever rows come last (1920x1080, 45-frame map: the march alone 0.336 -> 0.267 ms).
// A counting sort on 256 cost classes: any cost array gives a permutation, an all-zero one the row-major order.
#ifdef __HIPCC__
__device__ inline void tile_order_block(uint32_t *__restrict__ cost, uint32_t *__restrict__ order, int n) {
  __shared__ uint32_t hist[256], base[256], s_max;
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  for (int b = tid; b < 256; b += nt) hist[b] = 0;
  if (tid == 0) s_max = 0;
  __syncthreads();
  uint32_t mx = 0;
  for (int i = tid; i < n; i += nt) mx = cost[i] > mx ? cost[i] : mx;
  atomicMax(&s_max, mx);
  __syncthreads();
  const uint32_t top = s_max;
  const uint32_t shift = top > 255u ? (uint32_t)(32 - __clz((int)top)) - 8u : 0u;  // class = cost >> shift <= 255
  for (int i = tid; i < n; i += nt) atomicAdd(&hist[255u - (cost[i] >> shift)], 1u);
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the 256 class counts by one wavefront: four per lane, then across the lanes
    uint32_t v[4], run = 0;
    for (int k = 0; k < 4; k++) { v[k] = run; run += hist[4 * tid + k]; }
    uint32_t incl = run;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if (tid >= o) incl += t; }
    for (int k = 0; k < 4; k++) base[4 * tid + k] = incl - run + v[k];
  }
  __syncthreads();
  for (int i = tid; i < n; i += nt) {
    const uint32_t pos = atomicAdd(&base[255u - (cost[i] >> shift)], 1u);  // (the order inside a class is free)
    order[pos] = (uint32_t)i;
    cost[i] = 0;
  }
}
#endif

// list entry of the brick at window-relative brick coordinates (x9, y9, z9): 9 bits each
__host__ __device__ inline uint32_t brick_list_entry(uint32_t x9, uint32_t y9, uint32_t z9) { return (z9 << 18) | (y9 << 9) | x9; }

// Listing a stale brick, in the three steps the leaf kernel spreads over its barriers (one ring atomic per WORKGROUP: one per
// wavefront -- 4800 same-address atomics with a return value per 640x480 commit -- cost the kernel 30 us):
//  1. brick_mark_test: the lane's level-9 prefix (key of `depth` >= 9 levels) as a ring entry; true when this lane is the one
//     that lists it (the brick was not in the ring: test-and-set of its bit);
//  2. the workgroup counts its true lanes in LDS and reserves that many ring slots with ONE atomic (brick_ring_reserve);
//  3. brick_ring_store: the lane's entry into its slot.  Lapping the consumer is allowed: more than the capacity pending tells
//     the refresh to rebuild every brick.
__device__ inline bool brick_mark_test(uint32_t *dirty, bool pred, unsigned long long key, int depth, int shift, uint32_t &entry) {
  entry = 0;
  if (!pred) return false;
  const int nl = brick_node_level(shift);
  uint32_t x = 0, y = 0, z = 0;
  for (int k = 1; k <= nl; k++) {
    const uint32_t oct = (uint32_t)(key >> (3 * (depth - k))) & 7u;
    x = (x << 1) | (oct & 1u); y = (y << 1) | ((oct >> 1) & 1u); z = (z << 1) | (oct >> 2);
  }
  const uint32_t org = brick_window_origin(shift) >> 2;  // in bricks
  x -= org; y -= org; z -= org;
  if ((x | y | z) >= (kBrickWindowCells >> 2)) return false;  // outside the window: no brick to rebuild
  entry = brick_list_entry(x, y, z);
  const uint32_t bit = 1u << (entry & 31u);
  return !(atomicOr(&dirty[kBrickBitsOffset + (entry >> 5)], bit) & bit);
}
__device__ inline uint32_t brick_ring_reserve(uint32_t *dirty, uint32_t count) { return atomicAdd(&dirty[kBrickCountOffset], count); }
__device__ inline void brick_ring_store(uint32_t *dirty, uint32_t pos, uint32_t entry) {
  dirty[kBrickListOffset + (pos & (uint32_t)(kBrickListCap - 1))] = entry;
}
// the sibling ring (one returning atomic per entry: a few hundred per frame, at the end of the leaf kernel)
__device__ inline void brick_sibling_list(uint32_t *dirty, uint32_t entry) {
  const uint32_t pos = atomicAdd(&dirty[kSibCountOffset], 1u);
  dirty[kSibListOffset + (pos & (uint32_t)(kSibListCap - 1))] = entry;
}
#endif

}  // namespace svoslam
