// pool_grid.hip -- see pool_grid.hpp
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>

#include "config.hpp"
#include "pool_grid.hpp"

namespace svoslam {

namespace {
// Keyed by the address of the node memory (not of the caller's svoslam_pool struct, which the compatibility shim
// rebuilds on the stack for every call): that is also what a render is given.
std::mutex g_mu;
std::map<const uint32_t *, std::shared_ptr<PoolAccel>> g_accel;
}  // namespace

PoolAccel::~PoolAccel() {
  grid.release();
  shadow.release();
  for (uint32_t *d : d_dirty) if (d) (void)hipFree(d);
  if (bricks) (void)hipFree(bricks);
  if (d_brick_touched) (void)hipFree(d_brick_touched);
  if (ev_order) (void)hipEventDestroy(ev_order);
}

void pool_accel_forget_stream(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto &kv : g_accel)
    if (kv.second->last_stream == stream) kv.second->last_stream = nullptr;
}

void pool_accel_register(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end()) {
    g_accel.emplace(pool->d_data, std::make_shared<PoolAccel>());
  } else {
    it->second->valid = false;  // freshly initialised memory at a recycled address: whatever the grid holds is stale
    it->second->bricks_valid = false;
    it->second->max_depth = 0;
    it->second->mip_consistent = true;
  }
}

void pool_accel_rebind(const uint32_t *old_data, const uint32_t *new_data) {
  if (!new_data || old_data == new_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto stale = g_accel.find(new_data);
  if (stale != g_accel.end()) g_accel.erase(stale);  // an entry left behind by memory freed without svoslam_pool_free
  auto it = old_data ? g_accel.find(old_data) : g_accel.end();
  if (it == g_accel.end()) {
    g_accel.emplace(new_data, std::make_shared<PoolAccel>());
  } else {  // same nodes in a larger allocation: the grid stays what it is
    std::shared_ptr<PoolAccel> pa = std::move(it->second);
    g_accel.erase(it);
    g_accel.emplace(new_data, std::move(pa));
  }
}

void pool_accel_unregister(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end()) return;
  g_accel.erase(it);  // (~PoolAccel releases the device buffers once no enqueue holds the entry)
}

void pool_accel_invalidate(svoslam_pool *pool, int depth, bool foreign_words) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end()) return;
  it->second->valid = false; it->second->bricks_valid = false;
  if (depth < 0) { it->second->max_depth = 0; it->second->mip_consistent = true; }  // an empty pool
  else {
    if (depth > it->second->max_depth) it->second->max_depth = depth;
    if (foreign_words) it->second->mip_consistent = false;
  }
}

static bool ensure_dirty_states(PoolAccel *pa) {
  for (int k = 0; k < 2; k++) {
    if (pa->d_dirty[k]) continue;
    if (hipMalloc((void **)&pa->d_dirty[k], kPoolGridStateWords * 4) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (memset_sync(pa->d_dirty[k], 0, kPoolGridStateWords * 4) != hipSuccess) { (void)hipGetLastError(); return false; }
  }
  return true;
}

// Commits mark from the first one on, whether or not a grid exists yet: whether the NEXT render builds the grid in full
// is decided when that render is enqueued, which may be after the commit was (the scheduler enqueues the deferred
// commit of frame k+1 before the render of frame k).
uint32_t *pool_accel_dirty_bitmap(svoslam_pool *pool, int parity, int commit_depth, int *brick_shift) {
  if (brick_shift) *brick_shift = -1;
  if (!pool || !pool->d_data) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end() || !ensure_dirty_states(it->second.get())) return nullptr;
  if (commit_depth > it->second->max_depth) it->second->max_depth = commit_depth;
  const int shift = brick_shift_for_depth(it->second->max_depth);
  // a commit SHALLOWER than the brick node's level (a pool fused at mixed depths) changes nodes between the level grid and the
  // brick nodes that no key of it lists a brick for: every brick is rebuilt by the next refresh
  if (shift > 0 && commit_depth > kPoolGridLevel && commit_depth < brick_node_level(shift)) it->second->bricks_valid = false;
  if (brick_shift) *brick_shift = shift;  // the shape the next refresh will hold the bricks in
  return it->second->d_dirty[parity & 1];
}

int pool_shadow_begin(svoslam_pool *pool, hipStream_t stream, unsigned long long **d_shadow, uint32_t *epoch) {
  if (!pool || !pool->d_data || !d_shadow || !epoch) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end()) return SVOSLAM_ERR_INVALID_ARG;  // not a pool of this library
  PoolAccel *pa = it->second.get();
  if (pa->deferred_pending) return SVOSLAM_ERR_INVALID_ARG;  // one deferred commit at a time: apply it first
  if (pa->shadow_nodes < (size_t)pool->capacity) {
    // (never between a deferred commit and its apply: the pool only grows with nothing in flight)
    pa->shadow.release();
    SVO_TRY(pa->shadow.reserve((size_t)pool->capacity * 8));
    SVO_HIP(hipMemsetAsync(pa->shadow.ptr, 0, (size_t)pool->capacity * 8, stream));  // epoch 0 is never current
    pa->shadow_nodes = (size_t)pool->capacity;
    pa->epoch = 0;
  }
  pa->epoch += 1;
  pa->deferred_pending = true;
  *d_shadow = pa->shadow.as<unsigned long long>();
  *epoch = pa->epoch;
  return SVOSLAM_OK;
}

int pool_shadow_current(svoslam_pool *pool, unsigned long long **d_shadow, uint32_t *epoch) {
  if (!pool || !pool->d_data || !d_shadow || !epoch) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end() || !it->second->deferred_pending) return SVOSLAM_ERR_INVALID_ARG;
  *d_shadow = it->second->shadow.as<unsigned long long>();
  *epoch = it->second->epoch;
  return SVOSLAM_OK;
}

void pool_shadow_end(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it != g_accel.end()) it->second->deferred_pending = false;
}

bool pool_shadow_pending(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return false;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  return it != g_accel.end() && it->second->deferred_pending;
}

std::shared_ptr<PoolAccel> pool_accel_find(const uint32_t *d_data) {
  if (!d_data) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(d_data);
  return it == g_accel.end() ? nullptr : it->second;
}

// outcome of the reference's walk over levels 1..G on the path of cell (xi, yi, zi) (cone_tracing_kernels.cu:76-105):
//   all G nodes have children:  x = flag | tile index of the level-G node's children, y = its colour word
//   first childless node at level st (1..G): x = st, y = that node's colour word
// the same for a cell of level L <= G (xi, yi, zi: L bits each): an entry of the pyramid (pool_grid.hpp)
__device__ inline uint2 grid_entry_level(const uint2 *__restrict__ nodes, uint32_t xi, uint32_t yi, uint32_t zi, int L) {
  uint32_t base = 0;
  uint2 out = make_uint2(0u, 0u);
  for (int l = 1; l <= L; l++) {
    const int sh = L - l;
    const uint32_t oct = ((xi >> sh) & 1u) | (((yi >> sh) & 1u) << 1) | (((zi >> sh) & 1u) << 2);
    const uint2 nd = nodes[base + oct];
    if (!(nd.x & kFlag)) { out = make_uint2((uint32_t)l, nd.y); break; }
    base = nd.x & kMask;
    out = make_uint2(kFlag | base, nd.y);
  }
  return out;
}
// entry `e` of the pyramid (levels 1 .. G - 1, level by level)
__device__ inline void pyramid_write(const uint2 *__restrict__ nodes, uint2 *__restrict__ grid, uint32_t e) {
  constexpr int G = kPoolGridLevel;
  int l = 1;
  while (l < G - 1 && e >= pyr_offset(l + 1)) l++;
  const uint32_t c = e - pyr_offset(l), m = (1u << l) - 1u;
  grid[((size_t)1 << (3 * G)) + e] = grid_entry_level(nodes, c & m, (c >> l) & m, c >> (2 * l), l);
}

__device__ inline uint2 grid_entry(const uint2 *__restrict__ nodes, uint32_t xi, uint32_t yi, uint32_t zi) {
  constexpr int G = kPoolGridLevel;
  uint32_t base = 0;
  uint2 out = make_uint2(0u, 0u);
  for (int l = 1; l <= G; l++) {
    const int sh = G - l;
    const uint32_t oct = ((xi >> sh) & 1u) | (((yi >> sh) & 1u) << 1) | (((zi >> sh) & 1u) << 2);
    const uint2 nd = nodes[base + oct];
    if (!(nd.x & kFlag)) { out = make_uint2((uint32_t)l, nd.y); break; }
    base = nd.x & kMask;
    out = make_uint2(kFlag | base, nd.y);
  }
  return out;
}

__global__ __launch_bounds__(256) void pool_grid_build_kernel(const uint32_t *__restrict__ octree, uint2 *__restrict__ grid,
                                                              uint32_t *__restrict__ dirty_a, uint32_t *__restrict__ dirty_b) {
  constexpr int G = kPoolGridLevel;
  constexpr uint32_t kAxisMask = (1u << G) - 1u;
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  grid[e] = grid_entry(reinterpret_cast<const uint2 *>(octree), e & kAxisMask, (e >> G) & kAxisMask, e >> (2 * G));
  if (e < pyr_entries(G)) pyramid_write(reinterpret_cast<const uint2 *>(octree), grid, e);
  // every mark made so far is served by this build (not those of a deferred commit still running: dirty_b == nullptr)
  if (e < (uint32_t)kPoolGridDirtyWords) { dirty_a[e] = 0u; if (dirty_b) dirty_b[e] = 0u; }
  if (e == 0) { dirty_a[kPoolGridCountOffset] = 0u; if (dirty_b) dirty_b[kPoolGridCountOffset] = 0u; }
}

// ---- occupancy bricks (pool_grid.hpp) ----------------------------------------------------------------------------------
// the 64 entries of the bricks of N brick nodes -- window-relative brick coordinates (xr, yr, zr)[k], shape S -- one entry per
// lane of ONE wavefront and brick: lane = octant at level NL + 1 (bits 5..3) and at level NL + 2 = the cell level (bits 2..0)
// of the cell's path below the brick node (level NL = 9 + S).  A brick is a chain of 5 + S dependent loads (grid entry, the
// nodes of levels 9 .. NL on the path -- the same for every lane --, the two per-lane levels, the tile of the bits level); N
// chains are walked side by side.
// the level-8 node above brick node (xa, ya, za) (level-NL coordinates): its grid entry, or -- when the entry does not show
// children: the node has just been split and the grid (refreshed by the same launch) may not say so yet -- the walk from the
// root (rare: a few hundred nodes per frame at the map's frontier).  Returns flag | children tile, or 0.
template <int NL>
__device__ inline uint32_t brick_level8(const uint2 *__restrict__ nodes, uint2 g, uint32_t xa, uint32_t ya, uint32_t za) {
  constexpr int G = kPoolGridLevel;
  if (g.x & kFlag) return g.x;
  uint32_t base = 0;
  bool has = true;
  for (int l = 1; l <= G && has; l++) {
    const int sh = NL - l;  // level-NL coordinates: bit sh is level l's octant bit
    const uint2 nd = nodes[base + (((xa >> sh) & 1u) | (((ya >> sh) & 1u) << 1) | (((za >> sh) & 1u) << 2))];
    has = (nd.x & kFlag) != 0u;
    base = nd.x & kMask;
  }
  return has ? (kFlag | base) : 0u;
}

// SIBLINGS of a brick in the sibling ring (the commit that listed it created nodes at or above the brick node's level on
// its path).  A split that gives a node its first children creates seven childless siblings per level
// beside the key's path; nobody lists them (their content follows from the parent's tile alone), and without their lines a
// ray through them would find "no entry" under a level-8 node with children and walk the tree.  Here the children tile of every
// node on the wavefront-uniform part of the path (levels 9 .. NL) is read whole, and a sibling of the path's node that is
// CHILDLESS gets its lines -- 8^(NL - l) bricks whose 64 entries all say "the path stops at level l" -- unless its first entry
// already says so.  A sibling WITH children has been listed by the commit that gave it its children.  A routine of its own,
// called after the brick's rebuild for the few flagged entries (the tiles are cached by then): woven into the rebuild's load
// chains it cost the refresh 23 us per frame on the map stream (51 -> 74 us, cfg3 2390 -> 2290 frames/s); round 3 listed the
// seven level-9 siblings from the leaf kernel instead, a divergent loop of atomics at that kernel's end.
template <int S>
__device__ inline void brick_siblings(const uint2 *__restrict__ nodes, const uint2 *__restrict__ grid, uint16_t *__restrict__ bricks,
                                            uint32_t *__restrict__ touched, uint32_t xr, uint32_t yr, uint32_t zr, unsigned lane) {
  constexpr int G = kPoolGridLevel, NL = brick_node_level(S);
  constexpr uint32_t kOrg = brick_window_origin(S) >> 2;
  const uint32_t xa = xr + kOrg, ya = yr + kOrg, za = zr + kOrg;
  const uint32_t g8 = brick_level8<NL>(nodes, grid[((za >> (NL - G)) << (2 * G)) | ((ya >> (NL - G)) << G) | (xa >> (NL - G))], xa, ya, za);
  if (!(g8 & kFlag)) return;
  uint32_t tile = g8 & kMask, v = 0u;
  for (int l = G + 1; l <= NL; l++) {
    const int sh = NL - l;
    const uint32_t oct = ((xa >> sh) & 1u) | (((ya >> sh) & 1u) << 1) | (((za >> sh) & 1u) << 2);
    const uint32_t q = lane & 7u;
    const uint2 sib = nodes[tile + q];  // lane q < 8: sibling q (the path's own node at q == oct)
    const uint32_t code = l == NL ? 1u : 4u + (uint32_t)(l - G);
    const uint32_t want = v | code | (((sib.y >> 24) >= 254u) ? (1u << (l - 5)) : 0u);
    // first brick of sibling q (window-relative brick coordinates): the parent's prefix, q's octant bit at sh, zeros below
    const uint32_t m = ~((2u << sh) - 1u);
    const uint32_t sx = (xr & m) | ((q & 1u) << sh), sy = (yr & m) | (((q >> 1) & 1u) << sh), sz = (zr & m) | ((q >> 2) << sh);
    bool need = lane < 8u && q != oct && !(sib.x & kFlag);
    if (need) need = bricks[brick_entry_index(sx << 2, sy << 2, sz << 2)] != (uint16_t)want;
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const uint32_t bx = (uint32_t)__shfl((int)sx, src), by = (uint32_t)__shfl((int)sy, src), bz = (uint32_t)__shfl((int)sz, src);
      const uint32_t val = (uint32_t)__shfl((int)want, src);
      const uint32_t span = 1u << sh;  // bricks per axis under the sibling
      for (uint32_t dz = 0; dz < span; dz++)
        for (uint32_t dy = 0; dy < span; dy++)
          for (uint32_t dx = 0; dx < span; dx++)
            bricks[brick_entry_index(((bx + dx) << 2) | (lane & 3u), ((by + dy) << 2) | ((lane >> 2) & 3u), ((bz + dz) << 2) | (lane >> 4))] = (uint16_t)val;
      if (lane == 0) {
        const uint32_t grp = ((bz >> 3) << (2 * kBrickGroupLevel)) | ((by >> 3) << kBrickGroupLevel) | (bx >> 3);
        if (!((touched[grp >> 5] >> (grp & 31u)) & 1u)) atomicOr(&touched[grp >> 5], 1u << (grp & 31u));
      }
    }
    // on along the path
    const uint2 w = make_uint2((uint32_t)__shfl((int)sib.x, (int)oct), (uint32_t)__shfl((int)sib.y, (int)oct));
    v |= ((w.y >> 24) >= 254u) ? (1u << (l - 5)) : 0u;
    if (!(w.x & kFlag)) return;
    tile = w.x & kMask;
  }
}

template <int N, int S>
__device__ inline void brick_rebuild(const uint2 *__restrict__ nodes, const uint2 *__restrict__ grid, uint16_t *__restrict__ bricks,
                                     uint32_t *__restrict__ touched, const uint32_t (&xr)[N], const uint32_t (&yr)[N], const uint32_t (&zr)[N],
                                     const bool (&live)[N], unsigned lane, bool trust_mip) {
  constexpr int G = kPoolGridLevel, NL = brick_node_level(S);
  constexpr uint32_t kOrg = brick_window_origin(S) >> 2;  // in bricks
  uint32_t xa[N], ya[N], za[N];  // the brick nodes' coordinates at level NL
  uint2 g[N];
  uint32_t v[N], tile[N];
  bool on[N], write[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    xa[k] = xr[k] + kOrg; ya[k] = yr[k] + kOrg; za[k] = zr[k] + kOrg;
    g[k] = live[k] ? grid[((za[k] >> (NL - G)) << (2 * G)) | ((ya[k] >> (NL - G)) << G) | (xa[k] >> (NL - G))] : make_uint2(0u, 0u);
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (live[k]) g[k].x = brick_level8<NL>(nodes, g[k], xa[k], ya[k], za[k]);
    on[k] = (g[k].x & kFlag) != 0u;   // the path is alive: its node at the level above has children
    write[k] = on[k];                 // a line is written iff the level-8 node has children
    tile[k] = g[k].x & kMask;
    v[k] = 0u;
  }
  // the wavefront-uniform levels 9 .. NL
#pragma unroll
  for (int l = G + 1; l <= NL; l++) {
    uint2 w[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
      const int sh = NL - l;
      const uint32_t oct = ((xa[k] >> sh) & 1u) | (((ya[k] >> sh) & 1u) << 1) | (((za[k] >> sh) & 1u) << 2);
      w[k] = on[k] ? nodes[tile[k] + oct] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
      if (on[k]) {
        v[k] |= ((w[k].y >> 24) >= 254u) ? (1u << (l - 5)) : 0u;
        if (!(w[k].x & kFlag)) { v[k] |= l == NL ? 1u : 4u + (uint32_t)(l - G); on[k] = false; }  // the path stops here for every cell of the brick
        tile[k] = w[k].x & kMask;
      }
    }
  }
  // the two per-lane levels and the tile of the bits level
  uint2 w10[N], w11[N];
  bool on11[N], on12[N];
#pragma unroll
  for (int k = 0; k < N; k++) w10[k] = on[k] ? nodes[tile[k] + (lane >> 3)] : make_uint2(0u, 0u);
#pragma unroll
  for (int k = 0; k < N; k++) {
    v[k] |= ((w10[k].y >> 24) >= 254u) ? (1u << (NL + 1 - 5)) : 0u;
    on11[k] = on[k] && (w10[k].x & kFlag);
    if (on[k] && !on11[k]) v[k] |= 2u;
    w11[k] = on11[k] ? nodes[(w10[k].x & kMask) + (lane & 7u)] : make_uint2(0u, 0u);
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    v[k] |= ((w11[k].y >> 24) >= 254u) ? (1u << (NL + 2 - 5)) : 0u;
    on12[k] = on11[k] && (w11[k].x & kFlag);
    if (on11[k] && !on12[k]) v[k] |= 3u;
#ifdef SVO_BRICK_DIAG
    {  // cell-level nodes with children, and how many of them are saturated (would their tiles have to be read?)
      const unsigned long long m12 = __ballot(on12[k]), msat = __ballot(on12[k] && (w11[k].y >> 24) >= 254u);
      if (lane == 0) { atomicAdd(&touched[kBrickGroupWords + 3], (uint32_t)__popcll(m12)); atomicAdd(&touched[kBrickGroupWords + 4], (uint32_t)__popcll(msat)); }
    }
#endif
    if (on12[k]) v[k] |= 4u;
    // PoolAccel::mip_consistent: an unsaturated cell-level node has no saturated child; whether a child has children is left open
    if (on12[k] && trust_mip && (w11[k].y >> 24) < 254u) { v[k] |= 8u; on12[k] = false; }
    if (on12[k]) {
      const uint2 *t12 = nodes + (w11[k].x & kMask);  // the eight children at the bits level
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint2 c = t12[q];
        v[k] |= ((c.y >> 24) >= 254u) ? (0x100u << q) : 0u;
        v[k] |= (c.x & kFlag) ? 8u : 0u;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (!write[k]) continue;
    // cell of this lane (window-relative): x bit 1 = the octant bit at level NL + 1, x bit 0 = at level NL + 2 (likewise y, z)
    const uint32_t cx = (xr[k] << 2) | (((lane >> 3) & 1u) << 1) | (lane & 1u);
    const uint32_t cy = (yr[k] << 2) | (((lane >> 4) & 1u) << 1) | ((lane >> 1) & 1u);
    const uint32_t cz = (zr[k] << 2) | (((lane >> 5) & 1u) << 1) | ((lane >> 2) & 1u);
#ifdef SVO_BRICK_DIAG
    {  // how many rebuilt bricks actually change (diagnostic build)
      const bool diff = bricks[brick_entry_index(cx, cy, cz)] != (uint16_t)v[k];
      const unsigned long long dm = __ballot(diff);
      if (lane == 0) { atomicAdd(&touched[kBrickGroupWords], 1u); if (dm) atomicAdd(&touched[kBrickGroupWords + 1], 1u); atomicAdd(&touched[kBrickGroupWords + 2], (uint32_t)__popcll(dm)); }
    }
#endif
    bricks[brick_entry_index(cx, cy, cz)] = (uint16_t)v[k];
    if (lane == 0) {
      const uint32_t grp = ((zr[k] >> 3) << (2 * kBrickGroupLevel)) | ((yr[k] >> 3) << kBrickGroupLevel) | (xr[k] >> 3);
      if (!((touched[grp >> 5] >> (grp & 31u)) & 1u)) atomicOr(&touched[grp >> 5], 1u << (grp & 31u));
    }
  }
}

constexpr int kBrickThreads = 256, kBrickBlocks = 2048, kBrickChains = 4;
// the rings of the served states: entries [mark of the previous refresh, appended count), and this refresh's mark
// (count_off = kBrickCountOffset: the stale bricks; kSibCountOffset: the sibling ring; the marks follow the count)
struct BrickRings {
  uint32_t count[2], first[2];
};
__device__ inline BrickRings brick_rings(uint32_t *dirty_a, uint32_t *dirty_b, int par_a, int par_b, bool writer, int count_off = kBrickCountOffset) {
  BrickRings r;
  const int mark_off = count_off + 1;
  // (nothing appends to a served ring while this launch runs: commits of the served states are ordered around the render)
  r.count[0] = dirty_a ? dirty_a[count_off] : 0u; r.count[1] = dirty_b ? dirty_b[count_off] : 0u;
  r.first[0] = dirty_a ? dirty_a[mark_off + (par_a ^ 1)] : 0u; r.first[1] = dirty_b ? dirty_b[mark_off + (par_b ^ 1)] : 0u;
  if (writer) {  // what this refresh serves (nobody reads this word before the next refresh)
    if (dirty_a) dirty_a[mark_off + par_a] = r.count[0];
    if (dirty_b) dirty_b[mark_off + par_b] = r.count[1];
  }
  return r;
}
__device__ inline bool brick_rings_lapped(const BrickRings &r, uint32_t cap = (uint32_t)kBrickListCap) {
  return r.count[0] - r.first[0] > cap || r.count[1] - r.first[1] > cap;
}
// Either ring of the served states has lost entries since the last refresh.  The sibling ring counts too: with shape
// S = 1 a childless level-9 sibling's eight bricks carry stop code 5, and only the path's brick is rebuilt through the main ring
// when a key later enters it -- a lost sibling entry would leave seven lines saying "childless" that the march answers at level 9
// without the tree.  Both cases take the same way out: every group zeroed ("ask the level grid"), bricks come back as commits touch them.
__device__ inline bool brick_rings_lost(uint32_t *dirty_a, uint32_t *dirty_b, int par_a, int par_b) {
  return brick_rings_lapped(brick_rings(dirty_a, dirty_b, par_a, par_b, false)) ||
         brick_rings_lapped(brick_rings(dirty_a, dirty_b, par_a, par_b, false, kSibCountOffset), (uint32_t)kSibListCap);
}

// one wavefront per listed brick, C of them side by side; `part` of `parts` wavefronts
template <int C, int S>
__device__ inline void brick_rebuild_listed(const uint2 *__restrict__ nodes, const uint2 *__restrict__ grid, uint16_t *__restrict__ bricks,
                                            uint32_t *__restrict__ touched, uint32_t *dirty_a, uint32_t *dirty_b, const BrickRings &r,
                                            uint32_t part, uint32_t parts, unsigned lane, bool trust_mip) {
  for (int state = 0; state < 2; state++) {
    uint32_t *dirty = state ? dirty_b : dirty_a;
    const uint32_t pending = r.count[state] - r.first[state], first = r.first[state];
    for (uint32_t i0 = part * C; i0 < pending; i0 += parts * C) {
      uint32_t xs[C], ys[C], zs[C];
      bool live[C];
#pragma unroll
      for (int k = 0; k < C; k++) {
        live[k] = i0 + k < pending;
        const uint32_t e = live[k] ? dirty[kBrickListOffset + ((first + i0 + k) & (uint32_t)(kBrickListCap - 1))] : 0u;
        xs[k] = e & 511u; ys[k] = (e >> 9) & 511u; zs[k] = e >> 18;
        if (live[k] && lane == 0) atomicAnd(&dirty[kBrickBitsOffset + (e >> 5)], ~(1u << (e & 31u)));  // served: may be listed again
      }
      brick_rebuild<C, S>(nodes, grid, bricks, touched, xs, ys, zs, live, lane, trust_mip);
    }
  }
}

// the sibling ring: one wavefront per entry, `part` of `parts` wavefronts -- run by the workgroups of the refresh that update
// the level grid (they have little to do), so that the rebuild's wavefronts carry none of this (woven into their load chains it
// cost the refresh 10-23 us per frame; as a flag in the bricks' ring, found by scanning it, 18 us)
template <int S>
__device__ inline void brick_siblings_listed(const uint2 *__restrict__ nodes, const uint2 *__restrict__ grid, uint16_t *__restrict__ bricks,
                                             uint32_t *__restrict__ touched, const uint32_t *dirty_a, const uint32_t *dirty_b, const BrickRings &r,
                                             uint32_t part, uint32_t parts, unsigned lane) {
  for (int state = 0; state < 2; state++) {
    const uint32_t *dirty = state ? dirty_b : dirty_a;
    const uint32_t pending = r.count[state] - r.first[state];  // (<= kSibListCap: a lapped ring does not get here, brick_rings_lost)
    const uint32_t first = r.first[state];
    for (uint32_t i = part; i < pending; i += parts) {
      const uint32_t f = dirty[kSibListOffset + ((first + i) & (uint32_t)(kSibListCap - 1))] & 0x07FFFFFFu;
      brick_siblings<S>(nodes, grid, bricks, touched, f & 511u, (f >> 9) & 511u, f >> 18, lane);
    }
  }
}

// Every brick of the pool (a fresh field, a pool changed by anything but this library's commits, a change of shape): a
// wavefront reads 64 cells of the level grid and rebuilds the 8^(S + 1) bricks below each one that has children (those inside
// the window).  Everything listed so far in the served states is served by this pass.
template <int S>
__global__ __launch_bounds__(kBrickThreads) void brick_rebuild_kernel(const uint32_t *__restrict__ octree, const uint2 *__restrict__ grid,
                                                                      uint16_t *__restrict__ bricks, uint32_t *__restrict__ touched,
                                                                      uint32_t *__restrict__ dirty_a, uint32_t *__restrict__ dirty_b,
                                                                      int trust_mip, int par_a, int par_b) {
  const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  constexpr unsigned kWaves = kBrickThreads / 64;
  (void)brick_rings(dirty_a, dirty_b, par_a, par_b, blockIdx.x == 0 && threadIdx.x == 0);  // (writes this refresh's marks)
  (void)brick_rings(dirty_a, dirty_b, par_a, par_b, blockIdx.x == 0 && threadIdx.x == 0, kSibCountOffset);
  constexpr int G = kPoolGridLevel;
  constexpr uint32_t kCells = 1u << (3 * G);
  constexpr uint32_t kOrg = brick_window_origin(S) >> 2, kSpan = kBrickWindowCells >> 2;  // the window, in bricks
  for (uint32_t w = blockIdx.x * kBrickThreads + threadIdx.x; w < (uint32_t)kBrickBitsWords; w += kBrickBlocks * kBrickThreads) {  // no brick is "in the ring" any more
    if (dirty_a) dirty_a[kBrickBitsOffset + w] = 0u;
    if (dirty_b) dirty_b[kBrickBitsOffset + w] = 0u;
  }
  for (uint32_t c0 = (blockIdx.x * kWaves + wave) * 64u; c0 < kCells; c0 += kBrickBlocks * kWaves * 64u) {
    const uint32_t cell = c0 + lane;
    unsigned long long m = __ballot((grid[cell].x & kFlag) != 0u);
    while (m) {
      const uint32_t hit = c0 + (uint32_t)(__ffsll((long long)m) - 1);
      m &= m - 1ull;
      const uint32_t x8 = hit & ((1u << G) - 1u), y8 = (hit >> G) & ((1u << G) - 1u), z8 = hit >> (2 * G);
      // the bricks below the cell, eight at a time: `up` = the octants of the levels between the cell and the brick node's parent
      for (uint32_t up = 0; up < (1u << (3 * S)); up++) {
        uint32_t px = x8, py = y8, pz = z8;
        for (int q = S - 1; q >= 0; q--) { const uint32_t o = (up >> (3 * q)) & 7u; px = (px << 1) | (o & 1u); py = (py << 1) | ((o >> 1) & 1u); pz = (pz << 1) | (o >> 2); }
        uint32_t xs[8], ys[8], zs[8];
        bool live[8];
#pragma unroll
        for (uint32_t o = 0; o < 8u; o++) {
          xs[o] = ((px << 1) | (o & 1u)) - kOrg; ys[o] = ((py << 1) | ((o >> 1) & 1u)) - kOrg; zs[o] = ((pz << 1) | (o >> 2)) - kOrg;
          live[o] = (xs[o] | ys[o] | zs[o]) < kSpan;
        }
        if (live[0]) brick_rebuild<8, S>(nodes, grid, bricks, touched, xs, ys, zs, live, lane, trust_mip != 0);  // (aligned groups: all eight in or out; this pass visits the siblings itself)
      }
    }
  }
}

// zero the 64 KB groups that hold bricks (before every brick is rebuilt: what the field says about a pool that has been
// reset / reloaded / re-rooted since is stale) and clear their bits
__global__ __launch_bounds__(256) void brick_clear_kernel(uint16_t *__restrict__ bricks, uint32_t *__restrict__ touched) {
  for (uint32_t grp = blockIdx.x; grp < (uint32_t)kBrickGroups; grp += gridDim.x) {
    const bool set = (touched[grp >> 5] >> (grp & 31u)) & 1u;
    __syncthreads();
    if (!set) continue;
    uint4 *p = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(bricks) + (size_t)grp * kBrickGroupBytes);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(kBrickGroupBytes / 16); i += 256u) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0) atomicAnd(&touched[grp >> 5], ~(1u << (grp & 31u)));
  }
}

// one WORKGROUP per listed block (the list is compacted from the bitmap at the end of every commit), one cell per lane:
// rebuild its 8^3 cells if its bit is still set, then clear the bit -- a second render without a commit in between
// finds them clear.  (One wavefront per block with 8 cells per lane took 15-17 us per frame at 640x480: eight walks
// of eight dependent loads in sequence.)
constexpr int kUpdateThreads = 1 << (3 * (kPoolGridLevel - kPoolGridBlockLevel)), kUpdateBlocks = 2048;
// Both dirty states the render serves are consumed by ONE launch (the second is empty unless deferred commits are in
// use; as a launch of its own it showed as 6 us per frame in the kernel statistics -- frames/s are the same either way:
// 2804 against 2804 over 100 frames, A/B on one box).
__device__ inline void pool_grid_update_blocks(const uint2 *__restrict__ nodes, uint2 *__restrict__ grid, uint32_t *dirty_a, uint32_t *dirty_b,
                                               uint32_t part, uint32_t parts) {
  constexpr int G = kPoolGridLevel, B = kPoolGridBlockLevel, S = G - B;  // 2^S cells per block and axis
  for (int state = 0; state < 2; state++) {
    uint32_t *dirty = state ? dirty_b : dirty_a;
    if (!dirty) continue;
    const uint32_t count = dirty[kPoolGridCountOffset];
    const uint32_t *list = dirty + kPoolGridListOffset;
    for (uint32_t i = part; i < count; i += parts) {
      const uint32_t b = list[i];
      const bool set = (dirty[b >> 5] >> (b & 31u)) & 1u;
      __syncthreads();  // every lane has read the bit before lane 0 clears it
      if (!set) continue;
      const uint32_t bx = b & ((1u << B) - 1u), by = (b >> B) & ((1u << B) - 1u), bz = b >> (2 * B);
      for (uint32_t c = threadIdx.x; c < (uint32_t)kUpdateThreads; c += blockDim.x) {  // (one cell per lane of a 512-lane workgroup)
        const uint32_t xi = (bx << S) | (c & ((1u << S) - 1u)), yi = (by << S) | ((c >> S) & ((1u << S) - 1u)), zi = (bz << S) | (c >> (2 * S));
        grid[(zi << (2 * G)) | (yi << G) | xi] = grid_entry(nodes, xi, yi, zi);
      }
      // the pyramid's cells inside the block (levels B + 1 .. G - 1), the block's own and its ancestors' (levels B .. 1)
      for (uint32_t c = threadIdx.x; c < (uint32_t)B + pyr_offset(S); c += blockDim.x) {
        int l; uint32_t xi, yi, zi;
        if (c < (uint32_t)B) {  // level B - c: the block (c = 0) and its ancestors
          l = B - (int)c;
          xi = bx >> c; yi = by >> c; zi = bz >> c;
        } else {                // level B + d, 1 <= d < S: cell q of the 8^d inside the block
          const uint32_t r = c - (uint32_t)B;   // 0 .. 8 + 64 + ... - 1
          int d = 1;
          while (r >= pyr_offset(d + 1)) d++;
          const uint32_t q = r - pyr_offset(d), m = (1u << d) - 1u;
          l = B + d;
          xi = (bx << d) | (q & m); yi = (by << d) | ((q >> d) & m); zi = (bz << d) | (q >> (2 * d));
        }
        grid[((size_t)1 << (3 * G)) + pyr_offset(l) + ((zi << (2 * l)) | (yi << l) | xi)] = grid_entry_level(nodes, xi, yi, zi, l);
      }
      if (threadIdx.x == 0) atomicAnd(&dirty[b >> 5], ~(1u << (b & 31u)));
    }
  }
}

__global__ __launch_bounds__(kUpdateThreads) void pool_grid_update_kernel(const uint32_t *__restrict__ octree, uint2 *__restrict__ grid,
                                                                          uint32_t *__restrict__ dirty_a, uint32_t *__restrict__ dirty_b) {
  pool_grid_update_blocks(reinterpret_cast<const uint2 *>(octree), grid, dirty_a, dirty_b, blockIdx.x, kUpdateBlocks);
}

// The refresh before a march in ONE launch (round 3): workgroups [0, kRefreshBrickBlocks) rebuild the listed bricks, the rest
// the listed blocks of the level grid -- neither needs the other's result (a brick whose grid entry does not show the
// level-8 node's children yet walks down from the root; the bricks of a freshly split level-8 node's eight children are
// listed by the commit that split it, svo_build.hip), so the 14 us of the grid's update and a launch boundary leave the map stream.
// (workgroups of kBrickThreads; a frame marks a few hundred level-5 blocks and lists ~45 k bricks.)  The bricks' part in other shapes --
// kernel-trace means over 100 frames of cfg3, three runs each (tools/prof/refresh_ab.sh): 2048 workgroups x 4 bricks per wavefront 53-59 us,
// 2048 x 2 43-44, 4096 x 2 34-44, 4096 x 1 39-45, 2048 x 8 95.  The FRAME RATE does not follow: 4096 x 2 against 2048 x 4, eleven runs each on
// two boxes: 2028-2372 (median 2236) against 2019-2401 (median 2362) -- a march that starts 15 us earlier meets the other streams' launches
// at a worse moment (the schedule's steady states).  The slower shape stays.
constexpr int kRefreshBrickBlocks = kBrickBlocks, kRefreshGridBlocks = 1024, kRefreshChains = kBrickChains;
template <int C, int S>
__global__ __launch_bounds__(kBrickThreads) void pool_refresh_kernel(const uint32_t *__restrict__ octree, uint2 *grid, uint16_t *__restrict__ bricks,
                                                                      uint32_t *__restrict__ touched, uint32_t *dirty_a, uint32_t *dirty_b,
                                                                      int trust_mip, int par_a, int par_b, unsigned brick_blocks,
                                                                      uint32_t *__restrict__ tile_cost, uint32_t *__restrict__ tile_order, int n_tiles) {
  if (n_tiles > 0 && blockIdx.x == gridDim.x - 1) { tile_order_block(tile_cost, tile_order, n_tiles); return; }  // (the march's tile order: pool_grid.hpp)
  const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
  // (the bricks' workgroups first: they are the long ones; 2048 nearly empty grid workgroups ahead of them cost the launch 20 us)
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  constexpr unsigned kWaves = kBrickThreads / 64;
  const unsigned bb = blockIdx.x;
  if (bb >= brick_blocks) {
    pool_grid_update_blocks(nodes, grid, dirty_a, dirty_b, bb - brick_blocks, kRefreshGridBlocks);
    // ... and the childless siblings of the listed bricks whose commit created them (brick_siblings_listed)
    const BrickRings rs = brick_rings(dirty_a, dirty_b, par_a, par_b, bb == brick_blocks && threadIdx.x == 0, kSibCountOffset);
    if (!brick_rings_lost(dirty_a, dirty_b, par_a, par_b))  // (a lapped ring zeroes every group: nothing to complete)
      brick_siblings_listed<S>(nodes, grid, bricks, touched, dirty_a, dirty_b, rs, (bb - brick_blocks) * kWaves + wave, kRefreshGridBlocks * kWaves, lane);
    return;
  }
  const BrickRings r = brick_rings(dirty_a, dirty_b, par_a, par_b, bb == 0 && threadIdx.x == 0);
  if (brick_rings_lost(dirty_a, dirty_b, par_a, par_b)) {
    // more than a million distinct bricks (or 65536 sibling entries) listed since the last refresh: a ring has lost entries.  Every group
    // that holds bricks is zeroed ("ask the level grid": such samples walk the tree, correctly) and every brick may be listed
    // again; the bricks come back as commits touch them.
    for (uint32_t w = bb * kBrickThreads + threadIdx.x; w < (uint32_t)kBrickBitsWords; w += brick_blocks * kBrickThreads) {
      if (dirty_a) dirty_a[kBrickBitsOffset + w] = 0u;
      if (dirty_b) dirty_b[kBrickBitsOffset + w] = 0u;
    }
    for (uint32_t grp = bb; grp < (uint32_t)kBrickGroups; grp += brick_blocks) {
      if (!((touched[grp >> 5] >> (grp & 31u)) & 1u)) continue;
      uint4 *p = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(bricks) + (size_t)grp * kBrickGroupBytes);
      for (uint32_t i = threadIdx.x; i < (uint32_t)(kBrickGroupBytes / 16); i += kBrickThreads) p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  brick_rebuild_listed<C, S>(nodes, grid, bricks, touched, dirty_a, dirty_b, r, bb * kWaves + wave, brick_blocks * kWaves, lane, trust_mip != 0);
}

// svoslam_config.march_bricks = 0: no occupancy bricks (the march walks the tree below the level grid, as in round 2)
static bool bricks_enabled() { return config().march_bricks != 0; }

// The field is 16 GiB of the 288 GB, allocated by the first reference-mode render of a pool.  It is only taken when it leaves
// room: after it, at least twice its size must stay free for pool growth, shadow words and the caller's own
// tensors -- several live pools, or ranks / test processes sharing one device, otherwise end in an out-of-memory error far from
// here; a pool that does not get its field is marched through the tree (correct, slower).
static void ensure_bricks(PoolAccel *pa, hipStream_t stream) {
  if (pa->bricks || pa->bricks_failed) return;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); pa->bricks_failed = true; return; }
  if (free_b < 3 * kBrickFieldBytes) { pa->bricks_failed = true; return; }
  void *field = nullptr, *touched = nullptr;
  if (hipMalloc(&field, kBrickFieldBytes) != hipSuccess || hipMalloc(&touched, (kBrickGroupWords + 8) * 4) != hipSuccess ||
      hipMemsetAsync(field, 0, kBrickFieldBytes, stream) != hipSuccess ||
      hipMemsetAsync(touched, 0, (kBrickGroupWords + 8) * 4, stream) != hipSuccess) {
    (void)hipGetLastError();  // no room for the field on this device: the pool is marched through the tree
    if (field) (void)hipFree(field);
    if (touched) (void)hipFree(touched);
    pa->bricks_failed = true;
    return;
  }
  pa->bricks = reinterpret_cast<uint16_t *>(field);
  pa->d_brick_touched = reinterpret_cast<uint32_t *>(touched);
  pa->bricks_valid = false;
}

int pool_accel_refresh(PoolAccel *pa, const uint32_t *d_octree, hipStream_t stream, const uint2 **d_grid, bool want_bricks,
                       const uint16_t **d_bricks, int *brick_shift, uint32_t *tile_cost, uint32_t *tile_order, int n_tiles, bool *order_done) {
  if (order_done) *order_done = false;
  if (!pa || !d_octree || !d_grid) return SVOSLAM_ERR_INVALID_ARG;
  constexpr size_t kCells = (size_t)1 << (3 * kPoolGridLevel);
  // the whole enqueue under the lock: the grid's host-side state (valid, last_stream) and the launches that make it
  // true are one step for every other host thread
  std::lock_guard<std::mutex> lock(g_mu);
  if (!pa->grid.ptr) {
    SVO_TRY(pa->grid.reserve((kCells + pyr_entries(kPoolGridLevel)) * sizeof(uint2)));   // the level-8 grid, then the pyramid of levels 1 .. 7
    pa->valid = false;
  }
  if (!ensure_dirty_states(pa)) return SVOSLAM_ERR_HIP;
  if (pa->last_stream != nullptr && pa->last_stream != stream) {
    // another stream refreshed (and may still be building / marching through) this grid: order this refresh behind
    // everything enqueued there so far
    if (!pa->ev_order) SVO_HIP(hipEventCreateWithFlags(&pa->ev_order, hipEventDisableTiming));
    if (hipEventRecord(pa->ev_order, pa->last_stream) == hipSuccess) SVO_HIP(hipStreamWaitEvent(stream, pa->ev_order, 0));
    else (void)hipGetLastError();  // (a stream destroyed without svoslam_cone_trace_release: nothing left to wait for)
  }
  pa->last_stream = stream;
  // the brick shape follows the deepest fusion the pool has seen (commits enqueued so far list stale bricks in that shape: they
  // asked pool_accel_dirty_bitmap, which raised max_depth first); a change of shape rebuilds every brick
  const int shift = brick_shift_for_depth(pa->max_depth);
  if (want_bricks && bricks_enabled() && shift >= 0) ensure_bricks(pa, stream);  // (behind the ordering above: the first fill runs on `stream`)
  if (pa->bricks && shift != pa->brick_shift) { pa->brick_shift = shift; pa->bricks_valid = false; }
  const bool fresh = !pa->valid;
  pa->valid = true;
  uint32_t *serve[2] = {nullptr, nullptr};  // the dirty states this render consumes
  int serve_idx[2] = {-1, -1};
  if (pa->deferred_pending) {  // that commit's marks (parity of its epoch) belong to the render after its apply
    serve_idx[0] = (int)((pa->epoch + 1u) & 1u);
  } else {
    serve_idx[0] = 0; serve_idx[1] = 1;
  }
  for (int k = 0; k < 2; k++) serve[k] = serve_idx[k] >= 0 ? pa->d_dirty[serve_idx[k]] : nullptr;
  uint2 *grid = pa->grid.as<uint2>();
  const bool use_bricks = pa->bricks && shift >= 0;
  // once a pool has bricks every refresh keeps them current, whatever the mode of the render that asks
  const bool bricks_all = use_bricks && (fresh || !pa->bricks_valid);
  // the rings' marks alternate per state and refresh (kBrickMarkOffset)
  const int par_a = serve_idx[0] >= 0 ? (int)(pa->brick_served[serve_idx[0]] & 1u) : 0, par_b = serve_idx[1] >= 0 ? (int)(pa->brick_served[serve_idx[1]] & 1u) : 0;
  if (use_bricks) {
    const int trust = pa->mip_consistent ? 1 : 0;  // (PoolAccel::mip_consistent: rebuild 65 -> 39 us at cfg3)
    if (bricks_all) {
      if (fresh) pool_grid_build_kernel<<<(unsigned)(kCells / 256), 256, 0, stream>>>(d_octree, grid, serve[0], serve[1]);
      else pool_grid_update_kernel<<<kUpdateBlocks, kUpdateThreads, 0, stream>>>(d_octree, grid, serve[0], serve[1]);
      brick_clear_kernel<<<2048, 256, 0, stream>>>(pa->bricks, pa->d_brick_touched);
      if (shift == 0) brick_rebuild_kernel<0><<<kBrickBlocks, kBrickThreads, 0, stream>>>(d_octree, grid, pa->bricks, pa->d_brick_touched, serve[0], serve[1], trust, par_a, par_b);
      else brick_rebuild_kernel<1><<<kBrickBlocks, kBrickThreads, 0, stream>>>(d_octree, grid, pa->bricks, pa->d_brick_touched, serve[0], serve[1], trust, par_a, par_b);
    } else {
      // grid update and brick rebuild in ONE launch.  (Measured and not kept, round 3: the rebuild on a stream of its own beside
      // the march -- rays reach the surfaces before the rebuild does and pay tree walks: march 0.325 -> 0.395 ms; two launches;
      // 4096 workgroups x 2 bricks per wavefront -- kernel 55 -> 40 us, frame rate lower: DESIGN.md section 4.)
      const int extra = tile_cost && tile_order && n_tiles > 0 ? 1 : 0;  // one more workgroup: the tile order of the caller's march
      if (shift == 0) pool_refresh_kernel<kRefreshChains, 0><<<kRefreshBrickBlocks + kRefreshGridBlocks + extra, kBrickThreads, 0, stream>>>(d_octree, grid, pa->bricks, pa->d_brick_touched, serve[0], serve[1], trust, par_a, par_b, kRefreshBrickBlocks, tile_cost, tile_order, extra ? n_tiles : 0);
      else pool_refresh_kernel<kRefreshChains, 1><<<kRefreshBrickBlocks + kRefreshGridBlocks + extra, kBrickThreads, 0, stream>>>(d_octree, grid, pa->bricks, pa->d_brick_touched, serve[0], serve[1], trust, par_a, par_b, kRefreshBrickBlocks, tile_cost, tile_order, extra ? n_tiles : 0);
      if (extra && order_done) *order_done = true;
    }
    for (int k = 0; k < 2; k++)
      if (serve_idx[k] >= 0) pa->brick_served[serve_idx[k]]++;
    pa->bricks_valid = true;
#ifdef SVO_BRICK_DIAG
    {
      static int calls = 0;
      if (++calls % 50 == 0) {
        uint32_t c[5];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(c, pa->d_brick_touched + kBrickGroupWords, 20, hipMemcpyDeviceToHost);
        fprintf(stderr, "brick diag after %d refreshes: rebuilt %u bricks, %u of them changed, %u entries changed; %u cell-level nodes with children, %u of them saturated\n",
                calls, c[0], c[1], c[2], c[3], c[4]);
        (void)hipMemset(pa->d_brick_touched + kBrickGroupWords, 0, 20);
      }
    }
#endif
  } else {
    pa->bricks_valid = false;  // (a field that exists but is not kept current -- a pool deeper than any shape -- is stale from here on)
    if (fresh) pool_grid_build_kernel<<<(unsigned)(kCells / 256), 256, 0, stream>>>(d_octree, grid, serve[0], serve[1]);
    else pool_grid_update_kernel<<<kUpdateBlocks, kUpdateThreads, 0, stream>>>(d_octree, grid, serve[0], serve[1]);
  }
  SVO_LAUNCH_CHECK();
  *d_grid = grid;
  if (d_bricks) *d_bricks = use_bricks ? pa->bricks : nullptr;
  if (brick_shift) *brick_shift = use_bricks ? shift : -1;
  return SVOSLAM_OK;
}

}  // namespace svoslam
