// pool_grid.hip -- see pool_grid.hpp
#include <map>
#include <memory>
#include <mutex>

#include "pool_grid.hpp"

namespace svoslam {

namespace {
// Keyed by the address of the node memory (not of the caller's svoslam_pool struct, which the compatibility shim
// rebuilds on the stack for every call): that is also what a render is given.
std::mutex g_mu;
std::map<const uint32_t *, std::unique_ptr<PoolAccel>> g_accel;
}  // namespace

void pool_accel_register(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end()) {
    std::unique_ptr<PoolAccel> pa(new PoolAccel());
    g_accel.emplace(pool->d_data, std::move(pa));
  } else {
    it->second->valid = false;  // freshly initialised memory at a recycled address: whatever the grid holds is stale
  }
}

void pool_accel_rebind(const uint32_t *old_data, const uint32_t *new_data) {
  if (!new_data || old_data == new_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto stale = g_accel.find(new_data);
  if (stale != g_accel.end()) {  // an entry left behind by memory freed without svoslam_pool_free
    stale->second->grid.release();
    if (stale->second->d_dirty) (void)hipFree(stale->second->d_dirty);
    g_accel.erase(stale);
  }
  auto it = old_data ? g_accel.find(old_data) : g_accel.end();
  if (it == g_accel.end()) {
    g_accel.emplace(new_data, std::unique_ptr<PoolAccel>(new PoolAccel()));
  } else {  // same nodes in a larger allocation: the grid stays what it is
    std::unique_ptr<PoolAccel> pa = std::move(it->second);
    g_accel.erase(it);
    g_accel.emplace(new_data, std::move(pa));
  }
}

void pool_accel_unregister(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end()) return;
  it->second->grid.release();
  if (it->second->d_dirty) (void)hipFree(it->second->d_dirty);
  g_accel.erase(it);
}

void pool_accel_invalidate(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it != g_accel.end()) it->second->valid = false;
}

uint32_t *pool_accel_dirty_bitmap(svoslam_pool *pool) {
  if (!pool || !pool->d_data) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(pool->d_data);
  if (it == g_accel.end() || !it->second->valid) return nullptr;  // nothing to keep up to date (a full build is pending anyway)
  return it->second->d_dirty;
}

PoolAccel *pool_accel_find(const uint32_t *d_data) {
  if (!d_data) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_accel.find(d_data);
  return it == g_accel.end() ? nullptr : it->second.get();
}

// outcome of the reference's walk over levels 1..G on the path of cell (xi, yi, zi) (cone_tracing_kernels.cu:76-105):
//   all G nodes have children:  x = flag | tile index of the level-G node's children, y = its colour word
//   first childless node at level st (1..G): x = st, y = that node's colour word
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
                                                              uint32_t *__restrict__ dirty) {
  constexpr int G = kPoolGridLevel;
  constexpr uint32_t kAxisMask = (1u << G) - 1u;
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  grid[e] = grid_entry(reinterpret_cast<const uint2 *>(octree), e & kAxisMask, (e >> G) & kAxisMask, e >> (2 * G));
  if (e < (uint32_t)kPoolGridDirtyWords) dirty[e] = 0u;
  if (e == 0) dirty[kPoolGridCountOffset] = 0u;
}

// one WORKGROUP per listed block (the list is compacted from the bitmap at the end of every commit), one cell per lane:
// rebuild its 8^3 cells if its bit is still set, then clear the bit -- a second render without a commit in between
// finds them clear.  (One wavefront per block with 8 cells per lane took 15-17 us per frame at 640x480: eight walks
// of eight dependent loads in sequence.)
constexpr int kUpdateThreads = 1 << (3 * (kPoolGridLevel - kPoolGridBlockLevel)), kUpdateBlocks = 2048;
__global__ __launch_bounds__(kUpdateThreads) void pool_grid_update_kernel(const uint32_t *__restrict__ octree, uint2 *__restrict__ grid,
                                                                          uint32_t *__restrict__ dirty) {
  constexpr int G = kPoolGridLevel, B = kPoolGridBlockLevel, S = G - B;  // 2^S cells per block and axis
  const uint32_t count = dirty[kPoolGridCountOffset];
  const uint32_t *list = dirty + kPoolGridListOffset;
  const uint2 *nodes = reinterpret_cast<const uint2 *>(octree);
  const uint32_t c = threadIdx.x;
  for (uint32_t i = blockIdx.x; i < count; i += kUpdateBlocks) {
    const uint32_t b = list[i];
    const bool set = (dirty[b >> 5] >> (b & 31u)) & 1u;
    __syncthreads();  // every lane has read the bit before lane 0 clears it
    if (!set) continue;
    const uint32_t bx = b & ((1u << B) - 1u), by = (b >> B) & ((1u << B) - 1u), bz = b >> (2 * B);
    const uint32_t xi = (bx << S) | (c & ((1u << S) - 1u)), yi = (by << S) | ((c >> S) & ((1u << S) - 1u)), zi = (bz << S) | (c >> (2 * S));
    grid[(zi << (2 * G)) | (yi << G) | xi] = grid_entry(nodes, xi, yi, zi);
    if (c == 0) atomicAnd(&dirty[b >> 5], ~(1u << (b & 31u)));
  }
}

int pool_accel_refresh(PoolAccel *pa, const uint32_t *d_octree, hipStream_t stream, const uint2 **d_grid) {
  if (!pa || !d_octree || !d_grid) return SVOSLAM_ERR_INVALID_ARG;
  constexpr size_t kCells = (size_t)1 << (3 * kPoolGridLevel);
  bool fresh = false;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!pa->grid.ptr) {
      SVO_TRY(pa->grid.reserve(kCells * sizeof(uint2)));
      SVO_HIP(hipMalloc((void **)&pa->d_dirty, kPoolGridStateWords * 4));
      pa->valid = false;
    }
    fresh = !pa->valid;
    pa->valid = true;  // commits enqueued from now on mark their blocks
  }
  uint2 *grid = pa->grid.as<uint2>();
  if (fresh) pool_grid_build_kernel<<<(unsigned)(kCells / 256), 256, 0, stream>>>(d_octree, grid, pa->d_dirty);
  else pool_grid_update_kernel<<<kUpdateBlocks, kUpdateThreads, 0, stream>>>(d_octree, grid, pa->d_dirty);
  SVO_LAUNCH_CHECK();
  *d_grid = grid;
  return SVOSLAM_OK;
}

}  // namespace svoslam
