// pool_paging.hip -- out-of-core paging of sub-trees through the linear-tree format.
//
// Reference: OctreeNode::pushToGPU / pullToCPU / addToLinearTree / pullFromLinearTree (src/world/octree.cpp:41-169)
// move a sub-tree between a host pointer tree and a device "stackless" array: 2-word nodes, bit 30 = has children,
// low 30 bits = index of the first of the 8 children INSIDE that array, the sub-tree's own 8 top nodes first.  (The
// reference's writer never sets the flag, :138-160, so it has no byte stream to match; the format is what its reader
// expects.)  Here the device pool IS such an array for the whole map, so paging a sub-tree out means:
//
//   evict    breadth-first walk below the node named by an octant path (count + scan + emit per level: a
//            deterministic order), gather its tiles into a stand-alone linear tree with RELATIVE child indices -- a
//            valid pool on its own: it can be loaded with svoslam_pool_set_nodes and rendered / extracted --, write
//            it to a file together with the tiles' original indices; the node becomes childless (it keeps its colour
//            word, so a render sees the mip value), the tiles are zeroed;
//   restore  the tiles go back to the indices they came from, with absolute child indices, the node gets its flag
//            back: the pool is bit-identical to one that was never paged, provided nothing was fused INTO the evicted
//            cube meanwhile (then the node has children again and restore refuses).  Fusions elsewhere append their
//            tiles behind the evicted ones exactly as they would have without the eviction.
//
// Indices are never re-used (node numbering has to stay that of the uninterrupted run), so eviction does not shrink the
// allocation: what it buys is the host copy plus a device range that is no longer touched; a compacting re-index is a
// different operation (it changes every node index).  Both calls are blocking and wait for the whole device.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pool_grid.hpp"
#include "radix_sort.hpp"
#include "svo_build.hpp"

namespace svoslam {

typedef uint32_t u32;

__global__ __launch_bounds__(256) void page_count_kernel(const u32 *__restrict__ pool, const u32 *__restrict__ tiles, u32 n,
                                                         u32 *__restrict__ count) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint2 *t = reinterpret_cast<const uint2 *>(pool) + tiles[i];
  u32 c = 0;
  for (int j = 0; j < 8; j++) c += (t[j].x & kFlag) ? 1u : 0u;
  count[i] = c;
}

// next level's tiles in (parent order, octant order); the parents' nodes go to the blob with relative child indices
__global__ __launch_bounds__(256) void page_emit_kernel(const u32 *__restrict__ pool, const u32 *__restrict__ tiles, u32 n,
                                                        const u32 *__restrict__ offset, u32 next_base, u32 *__restrict__ next_tiles,
                                                        uint2 *__restrict__ blob_level) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint2 *t = reinterpret_cast<const uint2 *>(pool) + tiles[i];
  u32 k = offset[i];
  for (int j = 0; j < 8; j++) {
    uint2 nd = t[j];
    if (nd.x & kFlag) {
      next_tiles[k] = nd.x & kMask;
      nd.x = kFlag | (((next_base + k) * 8u) & kMask);
      k++;
    }
    blob_level[(size_t)i * 8 + j] = nd;
  }
}

__global__ __launch_bounds__(256) void page_zero_kernel(u32 *__restrict__ pool, const u32 *__restrict__ tiles, u32 n) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n * 8u) return;
  uint2 *nd = reinterpret_cast<uint2 *>(pool) + tiles[i >> 3] + (i & 7u);
  *nd = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(256) void page_scatter_kernel(u32 *__restrict__ pool, const u32 *__restrict__ tiles, u32 n,
                                                           const uint2 *__restrict__ blob) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n * 8u) return;
  uint2 nd = blob[i];
  if (nd.x & kFlag) nd.x = kFlag | (tiles[(nd.x & kMask) >> 3] & kMask);
  reinterpret_cast<uint2 *>(pool)[tiles[i >> 3] + (i & 7u)] = nd;
}

struct SubtreeHeader {
  char magic[8];  // "SVOSUBT1"
  uint32_t version;
  int32_t levels;
  uint8_t path[16];
  uint32_t node_index;   // the evicted node
  uint32_t num_tiles;
  int32_t pool_size;     // nodes in the pool when the sub-tree left it
  uint32_t reserved;
  uint64_t checksum;     // FNV-1a over tile indices + nodes
  uint8_t pad[8];
};
static_assert(sizeof(SubtreeHeader) == 64, "header is 64 bytes");

// a header's num_tiles is trusted only after it has been checked against the file's length (a corrupt
// count would otherwise size a std::vector / malloc and throw through the C ABI)
static bool tiles_fit_file(FILE *f, uint32_t num_tiles) {
  if (num_tiles == 0 || num_tiles > (kMask + 1u) / 8u) return false;
  const long at = ftell(f);
  if (at < 0 || fseek(f, 0, SEEK_END) != 0) return false;
  const long end = ftell(f);
  if (fseek(f, at, SEEK_SET) != 0) return false;
  return end >= 0 && (uint64_t)(end - at) >= (uint64_t)num_tiles * 68ull;
}

static uint64_t fnv1a(const void *p, size_t bytes, uint64_t h = 1469598103934665603ull) {
  const uint8_t *b = reinterpret_cast<const uint8_t *>(p);
  for (size_t i = 0; i < bytes; i++) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

// index of the node reached by the octant path; false if the path leaves the tree
static int walk_path(const svoslam_pool *pool, const uint8_t *path, int levels, u32 *node_out, u32 *word0_out) {
  u32 base = 0, node = 0, w0 = 0;
  for (int k = 0; k < levels; k++) {
    if (path[k] > 7) return SVOSLAM_ERR_INVALID_ARG;
    node = base + path[k];
    if ((int64_t)node >= (int64_t)pool->size) return SVOSLAM_ERR_FORMAT;
    SVO_HIP(hipMemcpy(&w0, pool->d_data + 2 * (size_t)node, 4, hipMemcpyDeviceToHost));
    if (k + 1 < levels) {
      if (!(w0 & kFlag)) return SVOSLAM_ERR_INVALID_ARG;  // the path ends above the requested level
      base = w0 & kMask;
    }
  }
  *node_out = node; *word0_out = w0;
  return SVOSLAM_OK;
}

int pool_evict_subtree(svoslam_pool *pool, const uint8_t *path, int levels, const char *file, hipStream_t stream) {
  if (!pool || !pool->d_data || !path || !file || levels < 1 || levels > SVOSLAM_MAX_DEPTH) return SVOSLAM_ERR_INVALID_ARG;
  SVO_HIP(hipDeviceSynchronize());
  SVO_TRY(pool_sync(pool, stream));
  u32 node = 0, w0 = 0;
  SVO_TRY(walk_path(pool, path, levels, &node, &w0));
  if (!(w0 & kFlag)) return SVOSLAM_ERR_INVALID_ARG;  // nothing below this node
  svoslam_workspace ws;
  std::vector<DeviceBuffer> level_tiles, level_blob;
  std::vector<u32> level_n;
  DeviceBuffer count, d_total;
  SVO_TRY(d_total.reserve(4));
  auto cleanup = [&]() {
    for (auto &b : level_tiles) b.release();
    for (auto &b : level_blob) b.release();
    count.release(); d_total.release(); ws.release_all();
  };
  {
    DeviceBuffer first;
    if (first.reserve(4) != SVOSLAM_OK) { cleanup(); return SVOSLAM_ERR_OOM; }
    const u32 root_tile = w0 & kMask;
    if (hipMemcpy(first.ptr, &root_tile, 4, hipMemcpyHostToDevice) != hipSuccess) { first.release(); cleanup(); return SVOSLAM_ERR_HIP; }
    level_tiles.push_back(first);
    level_n.push_back(1u);
  }
  u32 total_tiles = 1;
  int rc = SVOSLAM_OK;
  for (size_t k = 0; rc == SVOSLAM_OK; k++) {
    const u32 n = level_n[k];
    DeviceBuffer blob, next;
    if ((rc = count.reserve((size_t)n * 4)) != SVOSLAM_OK) break;
    if ((rc = blob.reserve((size_t)n * 64)) != SVOSLAM_OK) break;
    level_blob.push_back(blob);
    page_count_kernel<<<cdiv(n, 256), 256, 0, stream>>>(pool->d_data, level_tiles[k].as<u32>(), n, count.as<u32>());
    if ((rc = exclusive_scan_u32(&ws, count.as<u32>(), n, d_total.as<u32>(), stream)) != SVOSLAM_OK) break;
    u32 next_n = 0;
    if (hipMemcpyAsync(&next_n, d_total.ptr, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { rc = SVOSLAM_ERR_HIP; break; }
    if ((rc = next.reserve((size_t)(next_n ? next_n : 1) * 4)) != SVOSLAM_OK) break;
    page_emit_kernel<<<cdiv(n, 256), 256, 0, stream>>>(pool->d_data, level_tiles[k].as<u32>(), n, count.as<u32>(), total_tiles, next.as<u32>(),
                                                       level_blob[k].as<uint2>());
    if (hipStreamSynchronize(stream) != hipSuccess) { next.release(); rc = SVOSLAM_ERR_HIP; break; }
    if (next_n == 0) { next.release(); break; }
    if ((uint64_t)total_tiles + next_n > (uint64_t)(kMask + 1u) / 8u) { next.release(); rc = SVOSLAM_ERR_POOL_LIMIT; break; }
    level_tiles.push_back(next);
    level_n.push_back(next_n);
    total_tiles += next_n;
  }
  if (rc != SVOSLAM_OK) { cleanup(); return rc; }
  // host copy: tile indices and nodes in breadth-first order
  std::vector<u32> tiles(total_tiles);
  std::vector<u32> nodes((size_t)total_tiles * 16);
  size_t at = 0;
  for (size_t k = 0; k < level_n.size(); k++) {
    if (hipMemcpy(tiles.data() + at, level_tiles[k].ptr, (size_t)level_n[k] * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(nodes.data() + at * 16, level_blob[k].ptr, (size_t)level_n[k] * 64, hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); return SVOSLAM_ERR_HIP; }
    at += level_n[k];
  }
  SubtreeHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "SVOSUBT1", 8);
  h.version = 1; h.levels = levels;
  memcpy(h.path, path, (size_t)levels);
  h.node_index = node; h.num_tiles = total_tiles; h.pool_size = pool->size;
  h.checksum = fnv1a(nodes.data(), nodes.size() * 4, fnv1a(tiles.data(), tiles.size() * 4));
  FILE *f = fopen(file, "wb");
  if (!f) { cleanup(); return SVOSLAM_ERR_IO; }
  const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(tiles.data(), 4, tiles.size(), f) == tiles.size() &&
                  fwrite(nodes.data(), 4, nodes.size(), f) == nodes.size();
  if (fclose(f) != 0 || !ok) { cleanup(); return SVOSLAM_ERR_IO; }
  // only now touch the pool: the node becomes childless, the tiles are cleared
  DeviceBuffer all;
  if ((rc = all.reserve((size_t)total_tiles * 4)) != SVOSLAM_OK) { cleanup(); return rc; }
  const u32 zero = 0;
  if (hipMemcpy(all.ptr, tiles.data(), (size_t)total_tiles * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(pool->d_data + 2 * (size_t)node, &zero, 4, hipMemcpyHostToDevice) != hipSuccess) { all.release(); cleanup(); return SVOSLAM_ERR_HIP; }
  page_zero_kernel<<<cdiv((long long)total_tiles * 8, 256), 256, 0, stream>>>(pool->d_data, all.as<u32>(), total_tiles);
  const hipError_t e = hipStreamSynchronize(stream);
  all.release();
  cleanup();
  pool_accel_invalidate(pool);
  if (e != hipSuccess) { set_last_error("pool_evict_subtree", e); return SVOSLAM_ERR_HIP; }
  return SVOSLAM_OK;
}

int pool_restore_subtree(svoslam_pool *pool, const char *file, hipStream_t stream) {
  if (!pool || !pool->d_data || !file) return SVOSLAM_ERR_INVALID_ARG;
  FILE *f = fopen(file, "rb");
  if (!f) return SVOSLAM_ERR_IO;
  SubtreeHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SVOSUBT1", 8) != 0 || h.version != 1 || h.levels < 1 ||
      h.levels > SVOSLAM_MAX_DEPTH || !tiles_fit_file(f, h.num_tiles)) { fclose(f); return SVOSLAM_ERR_FORMAT; }
  std::vector<u32> tiles(h.num_tiles), nodes((size_t)h.num_tiles * 16);
  const bool ok = fread(tiles.data(), 4, tiles.size(), f) == tiles.size() && fread(nodes.data(), 4, nodes.size(), f) == nodes.size();
  fclose(f);
  if (!ok || fnv1a(nodes.data(), nodes.size() * 4, fnv1a(tiles.data(), tiles.size() * 4)) != h.checksum) return SVOSLAM_ERR_FORMAT;
  SVO_HIP(hipDeviceSynchronize());
  SVO_TRY(pool_sync(pool, stream));
  if (pool->size < h.pool_size) return SVOSLAM_ERR_FORMAT;  // not the pool (or not the state) the sub-tree came from
  for (u32 t : tiles)
    if ((t & 7u) || (int64_t)t + 8 > (int64_t)h.pool_size) return SVOSLAM_ERR_FORMAT;
  {  // a tile index may appear once
    std::vector<u32> sorted(tiles);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return SVOSLAM_ERR_FORMAT;
  }
  for (size_t i = 0; i < (size_t)h.num_tiles * 8; i++) {
    const u32 w0 = nodes[2 * i];
    if ((w0 & kFlag) && ((w0 & kMask) & 7u || ((w0 & kMask) >> 3) >= h.num_tiles)) return SVOSLAM_ERR_FORMAT;
  }
  u32 node = 0, w0 = 0;
  SVO_TRY(walk_path(pool, h.path, h.levels, &node, &w0));
  if (node != h.node_index) return SVOSLAM_ERR_FORMAT;
  if (w0 & kFlag) return SVOSLAM_ERR_INVALID_ARG;  // the cube was fused into while it was paged out
  DeviceBuffer d_tiles, d_nodes;
  SVO_TRY(d_tiles.reserve(tiles.size() * 4));
  int rc = d_nodes.reserve(nodes.size() * 4);
  if (rc != SVOSLAM_OK) { d_tiles.release(); return rc; }
  hipError_t e = hipMemcpy(d_tiles.ptr, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_nodes.ptr, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    page_scatter_kernel<<<cdiv((long long)h.num_tiles * 8, 256), 256, 0, stream>>>(pool->d_data, d_tiles.as<u32>(), h.num_tiles, d_nodes.as<uint2>());
    e = hipStreamSynchronize(stream);
  }
  const u32 flagged = kFlag | (tiles[0] & kMask);
  if (e == hipSuccess) e = hipMemcpy(pool->d_data + 2 * (size_t)node, &flagged, 4, hipMemcpyHostToDevice);
  d_tiles.release(); d_nodes.release();
  pool_accel_invalidate(pool);
  if (e != hipSuccess) { set_last_error("pool_restore_subtree", e); return SVOSLAM_ERR_HIP; }
  return SVOSLAM_OK;
}

// the stand-alone linear tree of a paged-out sub-tree (its 8 top nodes first): host words for svoslam_pool_set_nodes
int subtree_file_nodes(const char *file, uint32_t **h_words, int32_t *num_nodes) {
  if (!file || !h_words || !num_nodes) return SVOSLAM_ERR_INVALID_ARG;
  FILE *f = fopen(file, "rb");
  if (!f) return SVOSLAM_ERR_IO;
  SubtreeHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "SVOSUBT1", 8) != 0 || h.version != 1 || !tiles_fit_file(f, h.num_tiles)) { fclose(f); return SVOSLAM_ERR_FORMAT; }
  if (fseek(f, (long)h.num_tiles * 4, SEEK_CUR) != 0) { fclose(f); return SVOSLAM_ERR_FORMAT; }
  const size_t words = (size_t)h.num_tiles * 16;
  uint32_t *w = (uint32_t *)malloc(words * 4);
  if (!w) { fclose(f); return SVOSLAM_ERR_OOM; }
  const bool ok = fread(w, 4, words, f) == words;
  fclose(f);
  if (!ok) { free(w); return SVOSLAM_ERR_FORMAT; }
  *h_words = w; *num_nodes = (int32_t)(h.num_tiles * 8);
  return SVOSLAM_OK;
}

}  // namespace svoslam
