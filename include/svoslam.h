/*
 * svoslam.h -- C ABI of libsvoslam_hip.so, the MI355X (gfx950) drop-in for the
 * GPU kernel API of dkotfis/Octree-SLAM (the "L3" free functions of
 * SURVEY.md section 8b).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *  - every pointer named d_* is DEVICE memory (HBM); h_* is host memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    calls enqueue work on it and return without synchronising unless the
 *    comment says "blocking" (the reference API is blocking everywhere);
 *  - vectors are packed floats: vec3 = 3 floats (12 B, glm::vec3 layout),
 *    vec4 = 4 floats; mat4 = 16 floats column-major (glm::mat4 layout);
 *  - every function returns SVOSLAM_OK (0) or a negative svoslam_status; the
 *    reference returns void and checks nothing (SURVEY.md 8b "Errors");
 *  - the library fails loudly (SVOSLAM_ERR_NO_DEVICE) when no gfx950 device is
 *    present: there is no CPU fallback.
 *
 * Each entry point cites the reference declaration it replaces
 * (paths relative to the reference checkout).
 */
#ifndef SVOSLAM_H_
#define SVOSLAM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVOSLAM_ABI_VERSION 1

typedef enum {
  SVOSLAM_OK = 0,
  SVOSLAM_ERR_INVALID_ARG = -1,
  SVOSLAM_ERR_NO_DEVICE = -2,
  SVOSLAM_ERR_HIP = -3,
  SVOSLAM_ERR_OOM = -4,
  SVOSLAM_ERR_DEPTH = -5,      /* max_depth outside [1, SVOSLAM_MAX_DEPTH] */
  SVOSLAM_ERR_POOL_LIMIT = -6, /* pool would exceed 2^30 nodes (30-bit child index) */
  SVOSLAM_ERR_TRACKING_LOST = -7,
  SVOSLAM_ERR_IO = -8,         /* file could not be opened / read / written */
  SVOSLAM_ERR_FORMAT = -9      /* file is not what it should be (magic, size, checksum, structure) */
} svoslam_status;

#define SVOSLAM_MAX_DEPTH 16
#define SVOSLAM_FLAG_CHILDREN 0x40000000u /* word0 bit 30, svo.cu:130 */
#define SVOSLAM_CHILD_MASK 0x3FFFFFFFu    /* word0 bits 0-29, svo.cu:136 */

int svoslam_abi_version(void);
const char *svoslam_status_string(int status);

/* Library-wide settings (SURVEY section 5 "config / flags": the reference has compile-time constants and one #define,
 * include/octree_slam/world/svo/svo.h:8).  One struct through the ABI; a process may also preset it with the environment
 * variable SVOSLAM_CONFIG = "name=value,name=value" (field names below; tools and child-process tests), read once when the
 * library first needs a setting.  svoslam_config_set takes effect for objects created afterwards (cameras, runners) and for
 * the next render / fusion call.  -1 = automatic where noted. */
typedef struct svoslam_config {
  int32_t march_bricks;     /* 1: reference-mode renders of this library's pools march over occupancy bricks (0: the tree march) */
  int32_t track_mode;       /* 0: automatic (one launch per frame; its streaming form for large images); 1: the launch chain (two
                               launches per ICP iteration: several PROCESSES sharing one device); 2: the register-resident one-launch
                               form forced whatever the image size (tests) */
  int32_t track_workers;    /* 0: automatic; > 0: cap on the worker workgroups of the one-launch tracker */
  int32_t track_stream;     /* 1: large images use the streaming one-launch form; 0: coarsest level in one launch, then the chain */
  int32_t runner_deferred;  /* frame scheduler: deferred commits -1 automatic (on up to 640x480-class images), 0, 1 */
  int32_t runner_lead;      /* commits the host may run ahead of the device: -1 automatic */
  int32_t runner_prio;      /* stream priorities (map stream highest): -1 automatic, 0, 1 */
  int32_t runner_replicas;  /* 1, or 2 map replicas marched alternately (measured slower on one GPU: DESIGN.md) */
  int32_t runner_timeline;  /* 1: HIP-event marks at the stage boundaries (svoslam_runner_timeline); costs ~6 % */
  int32_t sort_pairs;       /* 1: force the (key, index) pair sort instead of the packed one-word sort */
  int32_t graphs;           /* 1: launch sequences recorded and replayed as HIP graphs (0: direct launches, the default) */
  int32_t march_ahead;      /* brick march: n >= 0 (default 60): past its first n steps a ray is marched in BURSTS of three samples -- the
                               current one and two reached by advancing with the previous step's level, their entries requested
                               together -- and follows the burst as long as each step ends on that level (results identical; overlaps
                               the round trips of a render's long rays); -1: one sample per iteration throughout */
  int32_t reserved[4];
} svoslam_config;
int svoslam_config_get(svoslam_config *out);
int svoslam_config_set(const svoslam_config *in);
/* last HIP error text seen by the calling thread (empty string if none) */
const char *svoslam_last_error(void);
/* name of the device the library runs on, e.g. "gfx950:..."; NULL if none */
const char *svoslam_device_arch(void);

/* ------------------------------------------------------------------------
 * Node pool.  Same layout as the reference (common_types.h:75-79, svo.cu):
 * node i = words 2i, 2i+1; word0 = children flag | 30-bit index of the first
 * of 8 contiguous children; word1 = R | G<<8 | B<<16 | A<<24; nodes 0-7 are
 * the root's children.  Unlike the reference (realloc + whole-pool copy per
 * frame, svo.cu:663-668) the pool keeps spare capacity and grows geometrically.
 * ---------------------------------------------------------------------- */
typedef struct {
  uint32_t *d_data;    /* 2*capacity words, device */
  int32_t size;        /* nodes in use (the reference's octree_size); exact unless `pending` > 0 */
  int32_t capacity;    /* nodes allocated */
  int32_t *d_size;     /* device-resident copy of the size (kept by the library; NULL for foreign pools) */
  int32_t pending;     /* asynchronous fusion calls enqueued since `size` was last exact */
  int64_t pending_bound; /* upper bound on the nodes those calls can add */
  void *tracker;       /* library-internal: non-blocking size readbacks of the asynchronous calls (NULL until first use) */
} svoslam_pool;

/* replaces svo::initOctree (svo.cu:24-31): 8 zeroed root children */
int svoslam_pool_init(svoslam_pool *pool, int32_t capacity_nodes, void *stream);
int svoslam_pool_reserve(svoslam_pool *pool, int32_t capacity_nodes, void *stream);
int svoslam_pool_free(svoslam_pool *pool);
/* back to the 8 zeroed root children of initOctree; the allocation (and with it every launch graph the
 * library has recorded against this pool) is kept.  Blocking: waits for the whole device. */
int svoslam_pool_reset(svoslam_pool *pool, void *stream);
/* makes pool->size exact again after asynchronous fusion calls (one stream sync + 4-byte readback) */
int svoslam_pool_sync(svoslam_pool *pool, void *stream);
/* Growing the mapped volume (a correct Octree::expandBySize, octree.cpp:362-378 -- the reference
 * rescales size_ without moving a node, Q16).  One doubling of the root cube towards `toward`: the old
 * root's children move to a new tile, nodes 0..7 become the new root's children, the one holding the old
 * root is flagged and coloured with its mean; all other node indices are unchanged.  center[3] and
 * *edge_length (half edge) are updated; fuse with max_depth + 1 afterwards to keep the resolution.  Blocking. */
int svoslam_pool_expand(svoslam_pool *pool, float center[3], float *edge_length, const float toward[3], void *stream);
/* Checkpoint / resume of a map.  The file is the linear tree as it sits in HBM -- the
 * layout OctreeNode::pushToGPU assembles (src/world/octree.cpp:41-79) -- behind a 64-byte header
 * (magic "SVOPOOL1", node count, root centre / half edge / depth, FNV-1a checksum).  Blocking; both wait
 * for the whole device first.  load() verifies the checksum and that every child tile lies inside the
 * pool, (re)allocates the pool and returns the root parameters (each may be NULL). */
int svoslam_pool_save(svoslam_pool *pool, const char *path, const float center[3], float edge_length, int32_t max_depth,
                      void *stream);
/* Tells the library that the pool's node memory was written behind its back (e.g. a hipMemcpy into pool->d_data):
 * derived data it keeps per pool -- the level grid of the ray march -- is rebuilt at the next render.  Pools must
 * otherwise be modified through the library only.  Host state only, no device access. */
int svoslam_pool_touch(svoslam_pool *pool);
/* What the ray march of this pool currently runs on (host state only; round 5: the fallback to the tree march when the 16 GiB
 * occupancy-brick field does not fit beside other pools / ranks of the device used to be silent): *has_grid = the level-8 grid
 * exists, *brick_state = 1 bricks in use, 0 not built (no reference-mode render yet, svoslam_config.march_bricks = 0, or a pool
 * deeper than any shape), -1 the field could not be allocated -- the pool is marched through the tree (correct, slower);
 * *brick_shift = the shape (0: depth <= 12, 1: depth 13 / 14, -1: none).  Any of the pointers may be NULL. */
int svoslam_pool_march_accel(const svoslam_pool *pool, int32_t *has_grid, int32_t *brick_state, int32_t *brick_shift);
/* Replaces the pool's contents by num_nodes host nodes (2 words each, reference format; child pointers validated) and
 * resets all size bookkeeping incl. the device-resident size the asynchronous fusion allocates from.  Blocking. */
int svoslam_pool_set_nodes(svoslam_pool *pool, const uint32_t *h_words, int32_t num_nodes, void *stream);
/* Out-of-core paging of sub-trees through the linear-tree format (the reference's unfinished
 * OctreeNode::pushToGPU / pullToCPU / addToLinearTree / pullFromLinearTree, src/world/octree.cpp:41-169).
 * evict: the sub-tree below the node reached by `path` (octants 0..7 from the root, `levels` of them; the node must have
 *   children) is written to `file` as a stand-alone linear tree (2-word nodes, bit 30 = children, low 30 bits = index
 *   of the first child inside the file's own array, the sub-tree's 8 top nodes first) plus the original tile indices;
 *   in the pool the node becomes childless (it keeps its colour word) and the tiles are zeroed.
 * restore: the tiles return to the indices they came from; the pool is then bit-identical to one that was never
 *   paged, also when other parts of the map were fused meanwhile (resume == uninterrupted).  Refused with
 *   SVOSLAM_ERR_INVALID_ARG when the cube has been fused into while it was out.
 * svoslam_subtree_file_nodes: the file's linear tree as host words (free() them) -- a pool of its own for
 *   svoslam_pool_set_nodes.  Node indices are never re-used, so eviction does not shrink the allocation.  Blocking. */
int svoslam_pool_evict_subtree(svoslam_pool *pool, const uint8_t *path, int32_t levels, const char *file, void *stream);
int svoslam_pool_restore_subtree(svoslam_pool *pool, const char *file, void *stream);
int svoslam_subtree_file_nodes(const char *file, uint32_t **h_words, int32_t *num_nodes);
/* dst becomes a byte-identical replica of src (nodes, size, at least src's capacity); dst may be zero-initialised.
 * Blocking (waits for the device). */
int svoslam_pool_copy(svoslam_pool *dst, svoslam_pool *src, void *stream);
int svoslam_pool_load(svoslam_pool *pool, const char *path, float center[3], float *edge_length, int32_t *max_depth,
                      void *stream);

/* Opaque scratch arena reused across calls (sort buffers, plan records...).
 * The reference cudaMallocs ~6+D temporaries per call instead. */
typedef struct svoslam_workspace svoslam_workspace;
int svoslam_workspace_create(svoslam_workspace **ws);
int svoslam_workspace_destroy(svoslam_workspace *ws);

/* per-call statistics of the fusion path (host, filled after the call) */
typedef struct {
  int32_t num_points;       /* n */
  int32_t num_split;        /* nodes split = new tiles allocated */
  int32_t pass_sizes[SVOSLAM_MAX_DEPTH + 1]; /* code_sizes[] of svo.cu:179-237 */
  int32_t pool_size_before;
  int32_t pool_size_after;
} svoslam_fuse_stats;

/* replaces svo::svoFromPointCloud (include/octree_slam/world/svo/svo.h:16,
 * src/world/svo/svo.cu:642-696).  d_points: n x vec3, d_colors: n x 3 bytes
 * (Color256).  center/edge_length = root centre and HALF edge.
 * Blocking once (one 4-byte count readback, as the reference's `int&
 * octree_size` requires the new size on the host). stats may be NULL. */
int svoslam_svo_from_point_cloud(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors,
                                 int32_t n, int32_t max_depth, svoslam_pool *pool, const float center[3],
                                 float edge_length, svoslam_fuse_stats *stats, void *stream);

/* Same fusion, fully asynchronous: nothing is read back, the new size stays on the device
 * (pool->d_size) and pool->size becomes exact again at the next blocking call or svoslam_pool_sync.
 * Capacity is reserved ahead for the worst case (sum over levels of min(8^d, n) splits per call);
 * only when that reservation runs out does the call synchronise once to learn the true size. */
int svoslam_svo_from_point_cloud_async(svoslam_workspace *ws, const float *d_points, const uint8_t *d_colors,
                                       int32_t n, int32_t max_depth, svoslam_pool *pool, const float center[3],
                                       float edge_length, void *stream);

/* The asynchronous fusion in its three phases, for callers that overlap it with a render of the
 * previous state (svoFromPointCloud then coneTraceSVO per frame, src/main.cpp:44,56):
 *   sort   keys + sort of the points; touches only the workspace
 *   plan   reads the pool's tree (must follow the previous commit; may run while the pool is ray-marched)
 *   commit writes the pool (splits, leaf blend, mip levels)
 * sort -> plan -> commit on one workspace == svoslam_svo_from_point_cloud_async.  Fusions in flight at
 * the same time need a workspace each.  If plan has to grow the pool it first waits for the whole device. */
int svoslam_svo_fuse_sort(svoslam_workspace *ws, const float *d_points, int32_t n, int32_t max_depth,
                          const float center[3], float edge_length, void *stream);
/* The sort phase fed by a raw depth frame: generateVertexMap (image_kernels.cu:24-58), transformVertexMap by the
 * DEVICE-resident mat4 d_pose (main.cpp:40-41), computePointCloudBoundingBox (image_kernels.cu:60-102; main.cpp:43) and
 * computeKeys (svo.cu:33-66) in ONE launch, without a point cloud in memory, then the sort: same sorted keys as
 * generate_vertex_map -> transform_vertex_map_dmat -> point_cloud_bbox_device -> svo_fuse_sort.  d_bbox7 (optional) =
 * {min xyz, max xyz, any}.  Needs 3 max_depth + 1 + ceil(log2(width height)) <= 64 (SVOSLAM_ERR_INVALID_ARG otherwise). */
int svoslam_svo_fuse_sort_frame(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int32_t width, int32_t height,
                                float fx, float fy, int32_t max_depth, const float center[3], float edge_length, float *d_bbox7,
                                void *stream);
int svoslam_svo_fuse_plan(svoslam_workspace *ws, int32_t n, int32_t max_depth, svoslam_pool *pool, void *stream);
int svoslam_svo_fuse_commit(svoslam_workspace *ws, const uint8_t *d_colors, int32_t n, int32_t max_depth,
                            svoslam_pool *pool, void *stream);
/* Sorted keys from elsewhere (frame-sharded sessions, DESIGN.md section 5: the rank that owns a frame sorts it and the sorted
 * arrays are all-gathered): _export_sorted copies the outcome of this workspace's sort phase to d_keys_out[n] (ascending
 * Morton keys, invalid points = 1) and d_idx_out[n] (point index per key; equal keys in ascending index order: R1);
 * _adopt_sorted makes such arrays the outcome of a workspace's sort phase -- svoslam_svo_fuse_plan / _split_early /
 * _commit follow as after svoslam_svo_fuse_sort.  The adopted arrays stay the caller's and must stay valid until the
 * commit has run.  Replaces nothing in the reference (svo.cu:602 sorts every cloud where it is fused). */
/* Row bands ("each GPU computes keys for its band ... all-gather of the sorted lists, merge -> identical global
 * list on every rank -> identical newNode = num_nodes + 8 x rank"): _sort_frame_band is svoslam_svo_fuse_sort_frame for rows
 * [first_row, first_row + rows) of the image, with whole-image point indices; _merge_sorted merges `lists` (<= 16) sorted
 * (key, index) lists with ascending, disjoint index ranges (bands in order) into d_keys_out / d_idx_out, which then hold
 * exactly what one sort of the whole frame produces: adopt them (below), plan, commit. */
int svoslam_svo_fuse_sort_frame_band(svoslam_workspace *ws, const uint16_t *d_depth, const float *d_pose, int32_t width, int32_t height,
                                     float fx, float fy, int32_t max_depth, const float center[3], float edge_length, int32_t first_row,
                                     int32_t rows, void *stream);
int svoslam_svo_fuse_merge_sorted(const unsigned long long *const *d_keys, const uint32_t *const *d_idx, const int32_t *counts, int32_t lists,
                                  unsigned long long *d_keys_out, uint32_t *d_idx_out, void *stream);
int svoslam_svo_fuse_export_sorted(svoslam_workspace *ws, int32_t n, unsigned long long *d_keys_out, uint32_t *d_idx_out, void *stream);
int svoslam_svo_fuse_adopt_sorted(svoslam_workspace *ws, const unsigned long long *d_keys, const uint32_t *d_idx, int32_t n, int32_t max_depth);
/* Optional, between plan and commit (same workspace, same pool, direct commit only): initialises the child tiles of the
 * planned splits (splitNodes' tile writes, svo.cu:239-276) beyond the pool's present size -- nothing a ray march of the pool
 * can reach -- so that it may run WHILE the previous frame is still being rendered; the commit then writes the links from
 * its leaf kernel and runs two launches instead of three.  Same pool contents as without the call. */
int svoslam_svo_fuse_split_early(svoslam_workspace *ws, int32_t n, int32_t max_depth, svoslam_pool *pool, void *stream);
/* The structure chain, for callers that do NOT render every frame (one rank of a frame-sharded session).  A plan reads only
 * the tree's structure words, and those are final once the previous frame's splits are in (splitNodes, svo.cu:239-276); its
 * leaf blend and mip levels (fillNodes / mipmapNodes, :291-465) write colour words only.  plan_structure = plan + all splits
 * with their links, tiles numbered from a size that follows the plans; it waits for no commit, only for the previous
 * plan_structure (same stream, or ordered by the caller).  The commits (svoslam_svo_fuse_commit on the same workspace: leaf
 * kernel + straddlers) follow in frame order on another stream.  The caller must keep a render of frame k away from the
 * structure of frame k+1: no plan_structure(k+1) before the render of frame k has finished.  pool_structure_begin: once per
 * sequence, ordered after everything earlier on the pool.  Same pool contents as plan + commit per frame. */
int svoslam_pool_structure_begin(svoslam_pool *pool, void *stream);
int svoslam_svo_fuse_plan_structure(svoslam_workspace *ws, int32_t n, int32_t max_depth, svoslam_pool *pool, void *stream);
/* The planned commit applied to one of several BYTE-IDENTICAL replicas of a map (a plan made against any replica in
 * the state before this commit fits all of them: same tree, same tile numbering).  Each application uses its own
 * slot (0 or 1; applications with different slots may run concurrently), all but the last pass keep_plan != 0.
 * Replicas must have been given the same capacity.  svoslam_svo_fuse_commit == (slot 0, keep_plan 0). */
int svoslam_svo_fuse_commit_to(svoslam_workspace *ws, const uint8_t *d_colors, int32_t n, int32_t max_depth,
                               svoslam_pool *pool, int32_t slot, int32_t keep_plan, void *stream);
/* The commit in two halves, for callers that ray-march the map while the next frame is being fused (the frame
 * scheduler): svoslam_svo_fuse_commit_deferred does all the work of the commit -- splitNodes, fillNodes, mipmapNodes,
 * svo.cu:239-465 -- without a store a concurrent cone trace of the pool in its present state can observe (new tiles
 * lie beyond the pool's size, colour words go to a shadow array of 8 bytes per node of capacity, the links of the
 * first-pass split nodes wait); svoslam_svo_fuse_apply (same workspace, before it is used again; any stream ordered
 * after the deferred call AND after the last reader of the old state) publishes it in one short launch.  Between the
 * two the pool may be read (old state) but not fused into, saved or resized; one deferred commit per pool at a time.
 * deferred + apply == svoslam_svo_fuse_commit, byte for byte. */
int svoslam_svo_fuse_commit_deferred(svoslam_workspace *ws, const uint8_t *d_colors, int32_t n, int32_t max_depth,
                                     svoslam_pool *pool, void *stream);
int svoslam_svo_fuse_apply(svoslam_workspace *ws, svoslam_pool *pool, void *stream);
/* Key-range sharded fusion: the plan + commit of ONE frame (splitKeys .. mipmapNodes, svo.cu:179-465) cut across `world` ranks by key range
 * instead of being replicated on every rank (the reference is single-GPU, cuda_renderer.cpp:68; protocol:
 * tests/test_keyrange_gloo.py).  Every rank holds a byte-identical replica of the pool and the frame's sorted keys (svoslam_svo_fuse_sort* +
 * export / merge).  Per frame and rank: keyrange_commit plans and commits the rank's slice -- the keys under a contiguous run of level-3
 * octree cells holding about n / world keys; every rank computes the same cuts -- where no replica can see it yet and writes the rank's
 * DELTA into d_delta (delta_bytes of room; 64 bytes per key of the slice is ample): bucket sizes, new tiles, frontier links, {node,
 * colour word} of the existing nodes it changed, records the receivers' ray-march marks need.  The caller all-gathers the deltas
 * (word 10 of a delta = the number of 32-bit words that carry data).  keyrange_apply (same workspace) renumbers every rank's tiles into the
 * reference's order (pass-major, depth, key), writes all deltas -- the own one included -- to their global place, makes the marks of the
 * level grid / occupancy bricks from all keys, recomputes the colour words above the splitter level and sets the pool's size: the replica
 * is then byte-identical to a pool that fused the frame in one piece.  d_deltas: HOST array of `world` device pointers in rank order.
 * Frames whose splits reach above level 3 (a node of level 1 or 2 without children: the first frames of a map) make several ranks plan the
 * same records: the deltas list those by key and the apply ranks them in the ranks' union -- no special case for the caller.
 * keyrange_status (blocking; optional, e.g. once per call of a frame loop): 0, or why an apply did NOT happen -- until it has been
 * read every later apply on the workspace is refused as well (a replica that silently missed a frame must not go on).
 * world <= 16, max_depth >= 6. */
#define SVOSLAM_KEYRANGE_OVERFLOW 2  /* a delta did not fit its buffer */
#define SVOSLAM_KEYRANGE_MISMATCH 4  /* deltas of different frames / pool states */
#define SVOSLAM_KEYRANGE_USED_WORD 10
int svoslam_svo_fuse_keyrange_commit(svoslam_workspace *ws, const unsigned long long *d_sorted_keys, const uint32_t *d_sorted_idx,
                                     const uint8_t *d_colors, int32_t n, int32_t max_depth, svoslam_pool *pool, int32_t rank, int32_t world,
                                     uint32_t *d_delta, int64_t delta_bytes, void *stream);
int svoslam_svo_fuse_keyrange_apply(svoslam_workspace *ws, const unsigned long long *d_sorted_keys, int32_t n, int32_t max_depth, svoslam_pool *pool,
                                    const uint32_t *const *d_deltas, int32_t world, void *stream);
int svoslam_svo_fuse_keyrange_status(svoslam_workspace *ws, int32_t *flags, void *stream);
/* instead of keyrange_apply: the delta was wanted, the commit is dropped (the pool stays as it was, the plan's reservation is released) */
int svoslam_svo_fuse_keyrange_discard(svoslam_workspace *ws, svoslam_pool *pool);

/* replaces svo::svoFromVoxelGrid (svo.h:14, svo.cu:584-640).  d_centers,
 * d_colors: n x vec4 (VoxelGrid, common_types.h:55-63). */
int svoslam_svo_from_voxel_grid(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int32_t n,
                                int32_t max_depth, svoslam_pool *pool, const float center[3], float edge_length,
                                svoslam_fuse_stats *stats, void *stream);

/* replaces svo::extractVoxelGridFromSVO (svo.h:18, svo.cu:699-745).  On return
 * *d_centers / *d_colors are hipMalloc'ed n x vec4 arrays owned by the caller
 * (free with svoslam_free), *n_out the voxel count.  Blocking. */
int svoslam_extract_voxel_grid(svoslam_workspace *ws, const svoslam_pool *pool, int32_t max_depth,
                               const float center[3], float edge_length, float **d_centers, float **d_colors,
                               int32_t *n_out, void *stream);
int svoslam_free(void *d_ptr);
/* device allocation / copies for callers that do not link the HIP runtime themselves (blocking copies) */
int svoslam_malloc(void **d_ptr, size_t bytes);
int svoslam_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int svoslam_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);

/* ------------------------------------------------------------------------
 * Mesh path (configs 1, 2, 5): OBJ/BMP loading and mesh -> voxel grid
 * ---------------------------------------------------------------------- */
/* Mesh of include/octree_slam/common_types.h:20-32 as Scene::loadObjFile fills it
 * (src/world/scene.cpp:26-33,115-133): HOST arrays, non-indexed (9 floats per triangle),
 * recentred (x/z centred, y min = 0; obj.cpp:227-238). */
typedef struct {
  float *vbo;       /* 9 * n_tris floats */
  float *tbo;       /* 6 * n_tris floats (uv per vertex) or NULL */
  int32_t n_tris;
  int32_t tbosize;  /* floats in tbo */
  float bbox0[3], bbox1[3];
} svoslam_mesh;
/* bmp_texture of common_types.h:34-38: HOST width*height*3 floats, r,g,b in 0..1 */
typedef struct {
  float *data;
  int32_t width, height;
} svoslam_texture;
/* Scene::loadObjFile (scene.cpp:26-33) = objLoader + obj::buildVBOs + objToMesh */
int svoslam_mesh_load_obj(const char *path, svoslam_mesh *out);
int svoslam_mesh_free(svoslam_mesh *mesh);
/* Scene::loadBMP (scene.cpp:35-62) */
int svoslam_texture_load_bmp(const char *path, svoslam_texture *out);
int svoslam_texture_free(svoslam_texture *tex);
/* replaces voxelization::meshToVoxelGrid (include/octree_slam/world/voxelization/voxelization.h:21,
 * src/world/voxelization/voxelization.cu:381-405) with N = 2^log_N cells per axis over the mesh's own
 * AABB (the reference fixes log_N = 8, log_T = 3: voxelization.cu:24-25).  tex may be NULL (voxels are
 * green, voxelization.cu:101-103).  Outputs are hipMalloc'ed n x vec4 arrays in ascending framebuffer
 * (tiled) index order, free with svoslam_free; d_indices (optional) receives those indices;
 * *scale_out = computeScale (half a voxel edge along x).  Blocking. */
int svoslam_mesh_to_voxel_grid(svoslam_workspace *ws, const svoslam_mesh *mesh, const svoslam_texture *tex, int32_t log_N,
                               int32_t log_T, float **d_centers, float **d_colors, unsigned long long **d_indices,
                               int32_t *n_out, float *scale_out, void *stream);
/* Statistics of the workspace's LAST svoslam_mesh_to_voxel_grid (measurement aid, no reference counterpart): the number of
 * (cell, triangle) fragments the scan-line rasteriser emitted and sorted -- the unit of the voxelizer's per-stage algorithmic
 * bytes; 0 before any call. */
int svoslam_mesh_last_fragments(const svoslam_workspace *ws, int64_t *fragments_out);

/* replaces voxelization::voxelGridToMesh (include/octree_slam/world/voxelization/voxelization.h:19,
 * src/world/voxelization/voxelization.cu:184-217, :325-379; SURVEY 8f.4, a display aid): one copy of the cube mesh
 * (host arrays: cube_vbosize floats of positions and of normals, cube_ibosize indices) per voxel, positions
 * cube * scale_factor + centre, colours replicated per vertex component, indices + idx * cube_ibosize (the
 * reference's offset).  scale_factor = computeScale(bbox) / CUBE_MESH_SCALE (0.1) in the reference.  Outputs are
 * device arrays of the caller: n * cube_vbosize floats (vbo, nbo, cbo), n * cube_ibosize ints (ibo).  Blocking. */
int svoslam_voxel_grid_to_mesh(svoslam_workspace *ws, const float *d_centers, const float *d_colors, int32_t n, float scale_factor,
                               const float *cube_vbo, int32_t cube_vbosize, const int32_t *cube_ibo, int32_t cube_ibosize,
                               const float *cube_nbo, float *d_vbo, int32_t *d_ibo, float *d_nbo, float *d_cbo, void *stream);

/* ------------------------------------------------------------------------
 * Recorded-sensor input: replaces sensor::OpenNIDevice
 * (src/sensor/openni_device.cpp:13-150) as the producer of RawFrame (common_types.h:65-73).
 * The association file is a TUM-RGB-D style list, one frame per line:
 *   <timestamp> <file> <timestamp> <file>     (depth and colour image in either order, '#' comments,
 *   paths relative to the list).  Images: PNG (8-bit RGB, 16-bit grey), binary PGM (16 bit) / PPM.
 * depth_units_per_metre converts the stored depth to the millimetres of RawFrame (1000 = already mm,
 * 5000 = TUM).  Timestamps are returned in microseconds.
 * ---------------------------------------------------------------------- */
typedef struct svoslam_frame_reader svoslam_frame_reader;
int svoslam_frame_reader_open(svoslam_frame_reader **reader, const char *association_file, float depth_units_per_metre);
int svoslam_frame_reader_close(svoslam_frame_reader *reader);
int svoslam_frame_reader_info(const svoslam_frame_reader *reader, int32_t *width, int32_t *height, int32_t *num_frames);
int svoslam_frame_reader_rewind(svoslam_frame_reader *reader);
/* next frame into HOST buffers (width*height uint16 / width*height*3 bytes); *got = 0 at the end of the list */
int svoslam_frame_reader_next_host(svoslam_frame_reader *reader, uint16_t *h_depth, uint8_t *h_color, long long *timestamp,
                                   int32_t *got);
/* OpenNIDevice::readFrame (:93-150): next frame into DEVICE buffers (pinned staging + async upload on `stream`) */
int svoslam_frame_reader_next(svoslam_frame_reader *reader, uint16_t *d_depth, uint8_t *d_color, long long *timestamp,
                              int32_t *got, void *stream);
/* openni_device.cpp:64-65: focal = size / (2 tan(fov / 2)), fov in radians */
int svoslam_focal_from_fov(int32_t width, int32_t height, float hfov_rad, float vfov_rad, float *fx, float *fy);
/* one image file into a malloc'ed host buffer (free with free()); 16-bit samples in host byte order */
int svoslam_image_load(const char *path, void **h_data, int32_t *width, int32_t *height, int32_t *channels, int32_t *bits);

/* ------------------------------------------------------------------------
 * Host objects: world::Scene + world::Octree (include/octree_slam/world/scene.h:20-81,
 * octree.h:80-125; src/world/scene.cpp, octree.cpp:251-385) as an opaque handle.
 * ---------------------------------------------------------------------- */
typedef struct svoslam_scene svoslam_scene;
int svoslam_scene_create(svoslam_scene **scene);                       /* Scene::Scene, scene.cpp:11-17 */
int svoslam_scene_destroy(svoslam_scene *scene);
int svoslam_scene_load_obj(svoslam_scene *scene, const char *path);    /* Scene::loadObjFile, scene.cpp:26-33 */
int svoslam_scene_load_bmp(svoslam_scene *scene, const char *path);    /* Scene::loadBMP, scene.cpp:35-62 */
/* optional: create the Octree explicitly (resolution, root centre, root half edge) before the first
 * insertion and/or pin the tree depth (depth_override > 0) instead of deriving it from
 * ceil(log2(edge/resolution)) (octree.cpp:284).  The reference derives everything from the first
 * cloud / mesh bounding box (scene.cpp:77-79,101). */
int svoslam_scene_set_octree(svoslam_scene *scene, float resolution, const float center[3], float size,
                             int32_t depth_override);
/* Scene::voxelizeMeshes(octree), scene.cpp:64-85.  log_N <= 0 selects the reference's 8. */
int svoslam_scene_voxelize_meshes(svoslam_scene *scene, int32_t octree, int32_t log_N, void *stream);
/* Scene::extractVoxelGridFromOctree, scene.cpp:87-96 (scale 0.01) */
int svoslam_scene_extract_voxel_grid(svoslam_scene *scene, void *stream);
/* Scene::addPointCloudToOctree, scene.cpp:98-113 (bbox = computePointCloudBoundingBox of the cloud) */
int svoslam_scene_add_point_cloud(svoslam_scene *scene, const float origin[3], const float *d_points,
                                  const uint8_t *d_colors, int32_t n, const float bbox0[3], const float bbox1[3],
                                  void *stream);
/* Scene::voxel_grid() accessor: device vec4 arrays owned by the scene */
int svoslam_scene_voxel_grid(svoslam_scene *scene, const float **d_centers, const float **d_colors, int32_t *n,
                             float *scale);
/* Scene::svo(bbox) -> Octree::extractSVO (octree.cpp:339-360): non-owning view of the pool */
int svoslam_scene_svo(svoslam_scene *scene, const uint32_t **d_data, float center[3], float *size, int32_t *num_nodes,
                      int32_t *max_depth);

/* ------------------------------------------------------------------------
 * Rendering
 * ---------------------------------------------------------------------- */
#define SVOSLAM_RENDER_REFERENCE 0 /* pixel stored only on retirement, as the reference does */
#define SVOSLAM_RENDER_CARRY 1     /* local pixel carried across march steps */

/* replaces rendering::coneTraceSVO (include/octree_slam/rendering/
 * cone_tracing_kernels.h:16, src/rendering/cone_tracing_kernels.cu:157-198).
 * d_pos: w*h uchar4 offscreen framebuffer (stands in for the mapped GL PBO of
 * cuda_renderer.cpp:158-171).  view = Camera.view.  d_steps (optional, may be
 * NULL): 2 x uint64 device counters {march steps, levels descended} the kernel
 * adds to (for the bytes model). */
int svoslam_cone_trace_svo(uint8_t *d_pos, int32_t width, int32_t height, float fov, const float view[16],
                           const uint32_t *d_octree, const float center[3], float size, int32_t mode,
                           unsigned long long *d_steps, void *stream);

/* Row band of the same render: traces rows [row_first, row_first+rows) of the
 * width x height frame into the full-frame buffer d_pos (multi-GPU image tiles). */
int svoslam_cone_trace_svo_band(uint8_t *d_pos, int32_t width, int32_t height, int32_t row_first, int32_t rows,
                                float fov, const float view[16], const uint32_t *d_octree, const float center[3],
                                float size, int32_t mode, unsigned long long *d_steps, void *stream);
/* Re-entrancy: the per-render acceleration data (level grid, split-plane tables) lives in a library-owned buffer PER
 * STREAM (17 MB, 135 MB for renders of a megapixel and more): renders enqueued on one stream share it in stream order,
 * renders on different streams never touch each other's; calls may come from several host threads.  The timing log
 * below and svoslam_timer_* are process-wide.  _release frees the buffer of one stream (all_streams != 0: of every
 * stream) once the caller has synchronised it. */
int svoslam_cone_trace_release(void *stream, int32_t all_streams);
/* Measurement aid: while enabled, every cone-trace call brackets its trace kernel (not the small
 * acceleration-structure build before it) with a pair of HIP events on the launch stream.
 * _read waits for the logged launches, returns the summed kernel time and their number, and clears the log. */
int svoslam_cone_trace_timing(int32_t enable);
int svoslam_cone_trace_timing_read(float *h_ms_sum, int32_t *h_launches);
/* The same for every stage of the frame (bench.py's per-stage rooflines; startTiming / stopTiming of
 * src/utils/timing_utils.cu:11-32 applied per stage): bit s of `mask` turns stage s on -- its launches are bracketed by
 * a pair of HIP events on the stream they run on (~2.6 us of that stream per record).  svoslam_stage_timing clears
 * every log; _read waits for the logged pairs of one stage, returns their summed duration and number, clears that log.
 * svoslam_cone_trace_timing(e) == turning SVOSLAM_STAGE_MARCH on / off. */
#define SVOSLAM_STAGE_MARCH 0        /* cone_trace_kernel alone */
#define SVOSLAM_STAGE_TRACKER 1      /* the 19 ICP iterations of a frame (one launch, or the launch chain) */
#define SVOSLAM_STAGE_FUSE_SORT 2    /* keys (+ back-projection in the frame loop) + radix sort */
#define SVOSLAM_STAGE_FUSE_PLAN 3    /* split planning (+ the early tile initialisation) */
#define SVOSLAM_STAGE_FUSE_COMMIT 4  /* splits, leaf blend, mip levels */
#define SVOSLAM_STAGE_MAPS 5         /* bilateral filter + vertex / normal pyramids */
#define SVOSLAM_STAGE_MESH_RASTER 6  /* meshToVoxelGrid: scan-line counts + scans + fragment emission (incl. the two count readbacks) */
#define SVOSLAM_STAGE_MESH_SORT 7    /* meshToVoxelGrid: sort of the (cell, triangle) fragments */
#define SVOSLAM_STAGE_MESH_EMIT 8    /* meshToVoxelGrid: last fragment per cell -> voxel centres + colours (incl. the count readback) */
#define SVOSLAM_STAGE_COUNT 9
int svoslam_stage_timing(uint32_t mask);
int svoslam_stage_timing_read(int32_t stage, float *h_ms_sum, int32_t *h_pairs);

/* ------------------------------------------------------------------------
 * Sensor image kernels (include/octree_slam/sensor/image_kernels.h:21-55,
 * src/sensor/image_kernels.cu)
 * ---------------------------------------------------------------------- */
/* generateVertexMap, image_kernels.h:24 / .cu:24-58 */
int svoslam_generate_vertex_map(const uint16_t *d_depth, float *d_vertex, int32_t width, int32_t height, float fx,
                                float fy, int32_t img_w, int32_t img_h, void *stream);
/* same, restricted to image rows [first_row, first_row+rows) of the full-frame buffers
 * (row band of a multi-GPU image split); pixel coordinates stay absolute */
int svoslam_generate_vertex_map_rows(const uint16_t *d_depth, float *d_vertex, int32_t width, int32_t height,
                                     int32_t first_row, int32_t rows, float fx, float fy, int32_t img_w, int32_t img_h,
                                     void *stream);
/* generateNormalMap, image_kernels.h:30 / .cu:104-139 */
int svoslam_generate_normal_map(const float *d_vertex, float *d_normal, int32_t width, int32_t height, void *stream);
/* bilateralFilter, image_kernels.h:34 / .cu:142-186 */
int svoslam_bilateral_filter(const uint16_t *d_in, uint16_t *d_out, int32_t width, int32_t height, void *stream);
/* subsampleDepth<uint16_t|float>, image_kernels.h:41-42 / .cu:236-289.  In place:
 * the (width/2 x height/2) result overwrites the head of d_data. d_tmp: scratch of
 * width*height/4 elements (the reference cudaMallocs it per call). */
int svoslam_subsample_depth_u16(uint16_t *d_data, uint16_t *d_tmp, int32_t width, int32_t height, void *stream);
int svoslam_subsample_depth_f32(float *d_data, float *d_tmp, int32_t width, int32_t height, void *stream);
/* subsample<float|Color256>, image_kernels.h:37-38 / .cu:291-326 */
int svoslam_subsample_f32(float *d_data, float *d_tmp, int32_t width, int32_t height, void *stream);
int svoslam_subsample_rgb8(uint8_t *d_data, uint8_t *d_tmp, int32_t width, int32_t height, void *stream);
/* colorToIntensity, image_kernels.h:45 / .cu:188-203 */
int svoslam_color_to_intensity(const uint8_t *d_rgb, float *d_out, int32_t n, void *stream);
/* transformVertexMap / transformNormalMap, image_kernels.h:52,55 / .cu:206-234 */
int svoslam_transform_vertex_map(float *d_vertex, const float trans[16], int32_t n, void *stream);
int svoslam_transform_normal_map(float *d_normal, const float trans[16], int32_t n, void *stream);
/* computePointCloudBoundingBox, image_kernels.h:27 / .cu:60-102.  bbox0/bbox1 are
 * host in/out (zero = unset sentinel).  Blocking. */
int svoslam_point_cloud_bbox(const float *d_points, int32_t n, float h_bbox0[3], float h_bbox1[3], void *stream);
/* non-blocking form for a device-resident frame loop: d_out7 = {min x,y,z, max x,y,z,
 * any_valid} of the valid points (no "zero = unset" merge with a previous box) */
int svoslam_point_cloud_bbox_device(svoslam_workspace *ws, const float *d_points, int32_t n, float *d_out7,
                                    void *stream);

/* ------------------------------------------------------------------------
 * ICP (include/octree_slam/sensor/localization_kernels.h:17-42,
 * src/sensor/localization_kernels.cu)
 * ---------------------------------------------------------------------- */
/* Photometric RGB-D term.  The reference DECLARES gradient / difference (image_kernels.h:45-49) and
 * computeRGBDCost (localization_kernels.h:42) but ships no definition of the first two, an empty body for the third
 * (localization_kernels.cu:328-331) and its call site commented out (rgbd_camera.cpp:126-141; W_RGBD = 0.1, :20).
 * There is no reference behaviour to match; the specification is this build's own (restated in
 * oracle/svoslam_oracle.c):
 *   gradient    Sobel 3x3 / 8 on interior pixels, (0,0) on the border; d_gradient = (gx, gy) per pixel
 *   difference  out = in1 - in2
 *   rgbd_cost   same-index association as computeICPCost2 (no reprojection), gates = finite + depth range + distance;
 *               r = I_last - I_cur; J = G_T * (gx du/dv + gy dv/dv) with the LAST frame's gradient, the pinhole
 *               derivative at the CURRENT vertex and the geometric term's own G_T rows (:208-213), so that both systems
 *               share one parametrisation; exact fixed-point sums.  fx, fy and the full image size are extra
 *               parameters (the reference's RGBDFrame carries no intrinsics).  Blocking (h_A, h_b on the host).
 *   svoslam_camera_set_rgbd(cam, 1), before the first frame: every ICP iteration solves A1 + W_RGBD A2, b1 + W_RGBD b2
 *   (the commented-out block of rgbd_camera.cpp:130-141); default off = the reference as shipped. */
int svoslam_gradient(const float *d_intensity, float *d_gradient, int32_t width, int32_t height, void *stream);
int svoslam_difference(const float *d_in1, const float *d_in2, float *d_out, int32_t n, void *stream);
int svoslam_rgbd_cost(const float *d_last_intensity, const float *d_last_gradient, const float *d_last_vertex,
                      const float *d_cur_intensity, const float *d_cur_vertex, int32_t width, int32_t height, float fx, float fy,
                      int32_t img_width, int32_t img_height, float h_A[36], float h_b[6], void *stream);
/* computeICPCost2, localization_kernels.h:39 / .cu:154-229,303-326.  h_A (36) and
 * h_b (6) are host outputs as in the reference.  Blocking. */
int svoslam_icp_cost2(const float *d_last_vertex, const float *d_last_normal, const float *d_cur_vertex,
                      const float *d_cur_normal, int32_t width, int32_t height, float h_A[36], float h_b[6],
                      void *stream);
/* computeICPCost, localization_kernels.h:38 / .cu:59-152,231-301: the variant with an explicit
 * correspondence stencil (finite, distance and normal gates; no depth-range gate), load size 10 and a
 * reduce over floor(M/10) partials: the first floor(M/10)*10 correspondences in pixel order contribute.
 * With M = 0 h_A and h_b are left untouched, as in the reference (:251-253).  *num_correspondences
 * (may be NULL) receives M.  Blocking. */
int svoslam_icp_cost(const float *d_last_vertex, const float *d_last_normal, const float *d_cur_vertex,
                     const float *d_cur_normal, int32_t width, int32_t height, float h_A[36], float h_b[6],
                     int32_t *num_correspondences, void *stream);
/* Non-blocking building block used by the tracker and by multi-GPU row bands:
 * adds the 27 exact fixed-point accumulators (21 upper-triangle A terms then 6
 * b terms, carried in float64) of pixels [first_pixel, first_pixel+num_pixels)
 * into d_acc[27].  Integer-valued doubles: sums are exact and associative, so
 * band partials can be all-reduced (RCCL sum, float64) in any order. */
int svoslam_icp_accumulate(const float *d_last_vertex, const float *d_last_normal, const float *d_cur_vertex,
                           const float *d_cur_normal, int32_t width, int32_t height, int32_t first_pixel,
                           int32_t num_pixels, double *d_acc, void *stream);

/* ------------------------------------------------------------------------
 * Tracker: host mirror of sensor::RGBDCamera (include/octree_slam/sensor/
 * rgbd_camera.h, src/sensor/rgbd_camera.cpp) with every per-iteration step
 * (accumulate, 6x6 Cholesky, pose compose) resident on the device.
 * ---------------------------------------------------------------------- */
typedef struct svoslam_camera svoslam_camera;
/* RGBDCamera::RGBDCamera, rgbd_camera.cpp:22-24.  band_first_row/band_rows select
 * the image rows this process owns for ICP accumulation (0, height = all). */
int svoslam_camera_create(svoslam_camera **cam, int32_t width, int32_t height, float fx, float fy);
int svoslam_camera_destroy(svoslam_camera *cam);
/* back to a new RGBDCamera (identity pose, no frame seen) keeping buffers and recorded launch graphs.  Blocking. */
int svoslam_camera_reset(svoslam_camera *cam);
int svoslam_camera_set_band(svoslam_camera *cam, int32_t first_row, int32_t rows);
/* RGBDCamera::update, rgbd_camera.cpp:53-191.  Non-blocking.  Returns 1 in
 * *processed if the frame was used, 0 if its timestamp was stale (:55-59). */
int svoslam_camera_update(svoslam_camera *cam, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp,
                          int32_t *processed, void *stream);
/* update() in its two halves, for callers that overlap them: prepare() filters the depth image and builds
 * the vertex/normal pyramids of the next frame (independent of earlier poses; up to two frames may be
 * prepared ahead), track() estimates the pose of the oldest prepared frame.  update() == prepare() then
 * track() on one stream.  With several streams the caller orders prepare(f) after track(f-2) and
 * track(f) after prepare(f) (the maps live in three rotating sets). */
int svoslam_camera_prepare(svoslam_camera *cam, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp,
                           int32_t *processed, void *stream);
int svoslam_camera_track(svoslam_camera *cam, void *stream);
/* Frame-parallel tracking (several GPUs, or several streams of one).  RGBDCamera::update starts update_trans at the
 * identity for every frame (rgbd_camera.cpp:100) and iterates on the maps of frames k-1 and k only (:103-168): a frame's
 * ICP is a function of two depth images; only :172-173 (position, orientation *= update_trans) chain the frames.
 *   pair_delta()  runs :62-168 for the pair (prev, cur) and writes SVOSLAM_DELTA_FLOATS floats to d_delta: update_trans
 *                 (mat4, column-major), the number of pyramid levels abandoned (:148-151) as int32 bits, 3 x padding.
 *                 `cam` serves as scratch (map sets, tracker); its pose afterwards means nothing and it must not be fed by
 *                 update() / apply_delta().  Non-blocking.
 *   apply_delta() :172-173 + main.cpp:40 for a delta from anywhere: a stream of apply_delta(pair_delta(k-1, k)) leaves the
 *                 camera exactly where a stream of update(k) leaves it.  d_delta = NULL, or the camera's first frame (no
 *                 ICP: `pass >= 1`, :99): pose unchanged.  A camera is fed either by update() or by apply_delta(), not both
 *                 (reset in between).  *processed as for update().  Non-blocking. */
#define SVOSLAM_DELTA_FLOATS 20
int svoslam_camera_pair_delta(svoslam_camera *cam, const uint16_t *d_depth_prev, const uint8_t *d_rgb_prev, const uint16_t *d_depth_cur,
                              const uint8_t *d_rgb_cur, float *d_delta, void *stream);
int svoslam_camera_apply_delta(svoslam_camera *cam, const float *d_delta, long long timestamp, int32_t *processed, void *stream);
/* Multi-GPU stepping: update() split at the all-reduce points.  begin() builds
 * the pyramids; for level = 2,1,0 and it = 0..iters(level)-1 call
 * icp_accumulate() [adds this band's 27 doubles into svoslam_camera_acc()], then
 * all-reduce that buffer across ranks, then icp_solve(); finally end(). */
int svoslam_camera_begin(svoslam_camera *cam, const uint16_t *d_depth, const uint8_t *d_rgb, long long timestamp,
                         int32_t *processed, void *stream);
int svoslam_camera_icp_iters(int32_t level); /* PYRAMID_ITERS, rgbd_camera.cpp:19 */
int svoslam_camera_icp_accumulate(svoslam_camera *cam, int32_t level, int32_t iter, void *stream);
double *svoslam_camera_acc(svoslam_camera *cam); /* device double[27] */
/* redirect the accumulators to caller-owned device memory (e.g. a torch tensor that
 * torch.distributed all-reduces); NULL restores the internal buffer.  Must be zeroed
 * by the caller once; icp_solve() re-zeroes it after every iteration. */
int svoslam_camera_set_acc(svoslam_camera *cam, double *d_acc);
/* number of pyramid levels abandoned because the solve returned NaN
 * ("Camera tracking is lost.", rgbd_camera.cpp:148-151).  Blocking. */
int svoslam_camera_tracking_lost_count(svoslam_camera *cam, int32_t *count, void *stream);
/* switches the photometric RGB-D term on (see svoslam_rgbd_cost above); only before the first frame */
int svoslam_camera_set_rgbd(svoslam_camera *cam, int32_t enable);
/* strict = 1 (default): RGBDCamera::update as the reference has it, every quirk included.  strict = 0: this build's CORRECTED
 * tracker (own specification, no reference behaviour; SURVEY section 7 step 1 "every Appendix-B quirk behind a strict_reference
 * switch").  Three places where the reference's update is not a rigid-motion estimate are replaced, everything else -- gates,
 * pyramid, iteration counts, exact fixed-point sums, Cholesky, glm products -- stays:
 *   - the rotational rows of the Jacobian are those of [v2]x (A_T[0..2] = v2 x n1) instead of the rows of
 *     src/sensor/localization_kernels.cu:207-213 (Q14);
 *   - this_trans = translate(x3, x4, x5) * Rz(x2) * Ry(x1) * Rx(x0) (a current-frame point goes to R v + t) instead of
 *     Rz(-x2) * Ry(-x1) * Rx(-x0) * translate(..) (rgbd_camera.cpp:154-158);
 *   - position = (position + t) * update_trans in the row-vector product of rgbd_camera.cpp:172 (Q17 drops t), so that
 *     main.cpp:40's orientation * (x + position) composes the frame-to-frame transforms.
 * Restated in oracle/svoslam_oracle.c (ora_camera_set_strict_reference) and a second time in tests/test_oracle_second_opinion.py
 * (track_second_opinion(..., corrected=True)).
 * Before the first frame only; not combined with the photometric term. */
int svoslam_camera_set_strict_reference(svoslam_camera *cam, int32_t strict);
/* Frame-to-model tracking (second half).  OWN SPECIFICATION: the reference tracks every frame against the
 * previous FRAME's maps and leaves the rest as a TODO (src/sensor/rgbd_camera.cpp:185: "ICP should not swap, as
 * last_frame should be updated by a different function"); these three entry points are that different function.
 *   svoslam_raycast_model_depth   the map ray-cast into a depth image in the SENSOR's pixel grid and unit: pixel (x, y)
 *       looks along ((x - w/2)/fx, (h/2 - y)/fy, 1) (the direction generateVertexMap gives it, image_kernels.cu:24-58)
 *       carried into the map by cam_to_world (the fusion transform of main.cpp:40; exactly one of the host matrix /
 *       the device pointer -- e.g. svoslam_camera_fusion_transform_device -- is given); marched as coneTrace marches
 *       (cone_tracing_kernels.cu:53-146, pixel scale 1/fy) to the first sample whose node has A >= 254; the pixel is
 *       rint(1000 z) as uint16, 0 = nothing met.  d_steps (optional): 1 x uint64 device counter of march steps.
 *   svoslam_camera_set_model_depth   such an image goes through the front end of a sensor frame (bilateral filter,
 *       pyramid, vertex + normal maps: rgbd_camera.cpp:62-93) into a map set of its own.  Non-blocking, stream-ordered.
 *       d_depth == NULL: no model -- the following frames are tracked against the previous frame, until the next one.
 *   svoslam_camera_set_frame_to_model(cam, 1)   every ICP iteration associates the incoming frame with that set (once one
 *       has been given) instead of the previous frame's maps; everything else -- gates, sums, solve, pose composition
 *       -- is RGBDCamera::update unchanged.  Not combined with the photometric term (SVOSLAM_ERR_INVALID_ARG).
 * Restated on the CPU as ora_raycast_model_depth / ora_camera_set_model_depth / ora_camera_set_frame_to_model. */
int svoslam_raycast_model_depth(uint16_t *d_depth, int32_t width, int32_t height, float fx, float fy, const float *cam_to_world,
                                const float *d_cam_to_world, const uint32_t *d_octree, const float center[3], float size,
                                unsigned long long *d_steps, void *stream);
int svoslam_camera_set_model_depth(svoslam_camera *cam, const uint16_t *d_depth, void *stream);
int svoslam_camera_set_frame_to_model(svoslam_camera *cam, int32_t enable);
/* the newest timestamp the camera has accepted (rgbd_camera.cpp:55-59 skips frames that are not newer); *have = 0
 * before the first frame.  Host state, no device access. */
int svoslam_camera_latest_timestamp(svoslam_camera *cam, int32_t *have, long long *timestamp);
/* diagnostic (libraries built with -DSVO_TRK_PROF; zeros otherwise): device clock stamps of the last tracked frame's
 * one-launch tracker, h_stamps[32][8] = per ICP iteration {solver: start, fan-in done, rows summed, published;
 * worker 0: start, terms done, row stored, broadcast received}.  Blocking. */
int svoslam_camera_track_profile(svoslam_camera *cam, unsigned long long *h_stamps, void *stream);
int svoslam_camera_icp_solve(svoslam_camera *cam, int32_t level, int32_t iter, void *stream);
int svoslam_camera_end(svoslam_camera *cam, void *stream);
/* position() / orientation() accessors, rgbd_camera.h:33-36.  Blocking (D2H). */
int svoslam_camera_pose(svoslam_camera *cam, float h_position[3], float h_orientation[9], void *stream);
/* device mat4 = mat4(orientation) * translate(I, position) (main.cpp:40); stays on the device */
const float *svoslam_camera_fusion_transform_device(svoslam_camera *cam);
/* last A, b, x (host copies; blocking) for tests */
int svoslam_camera_last_system(svoslam_camera *cam, float h_A[36], float h_b[6], float h_x[6], void *stream);
/* current-frame pyramid maps (device, level 0..2) for tests: after update() the
 * frames have been swapped, so these are the maps of the frame just processed */
const float *svoslam_camera_last_vertex(svoslam_camera *cam, int32_t level);
const float *svoslam_camera_last_normal(svoslam_camera *cam, int32_t level);

/* transformVertexMap with a DEVICE matrix (keeps main.cpp:40 off the host) */
int svoslam_transform_vertex_map_dmat(float *d_vertex, const float *d_trans, int32_t n, void *stream);

/* ------------------------------------------------------------------------
 * Native frame scheduler: the per-frame loop of main.cpp:31-84 (RGBDCamera::update -> generateVertexMap ->
 * transformVertexMap -> computePointCloudBoundingBox -> Scene::addPointCloudToOctree -> coneTraceSVO), software-
 * pipelined over four HIP streams inside the library (csrc/runner.hip) so that a frame costs the host ~0.1 ms
 * instead of the 0.45 ms of a scripted loop.  Results are those of calling the stages one after the other.
 * svoslam_runner_run enqueues n device-resident frames (depth u16 mm, RGB888; strictly increasing timestamps;
 * views = n column-major 4x4 view matrices) after the work already queued on caller_stream, makes caller_stream
 * wait for all of it, and returns without waiting; d_image receives rows [row_first, row_first + rows) of each
 * frame's raycast (the last frame's remain), d_steps (optional) 2 x u64 step / level counters.
 * ---------------------------------------------------------------------- */
typedef struct svoslam_runner svoslam_runner;
int svoslam_runner_create(svoslam_runner **runner, svoslam_camera *cam, svoslam_pool *pool, int32_t width, int32_t height,
                          int32_t max_depth, const float center[3], float edge_length, float fx, float fy, int32_t render_mode);
int svoslam_runner_destroy(svoslam_runner *runner);
/* diagnostic (runner created with SVOSLAM_RUNNER_TIMELINE=1 in the environment): HIP-event times in ms of the stage
 * boundaries of the last call, relative to its first mark: h_ms[frame][10] = {maps begin, maps end, track begin, pose,
 * prepare begin, plan begin, plan end, commit begin, commit end, march end}.  Blocking. */
int svoslam_runner_timeline(svoslam_runner *runner, float *h_ms, int32_t max_frames, int32_t *frames);
int svoslam_runner_run(svoslam_runner *runner, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs,
                       const long long *timestamps, const float *views, int32_t n, uint8_t *d_image, int32_t row_first,
                       int32_t rows, unsigned long long *d_steps, void *caller_stream);
/* The loop with FRAME-TO-MODEL tracking (own specification, see svoslam_raycast_model_depth below: the reference
 * leaves it as a TODO, src/sensor/rgbd_camera.cpp:185).  Per frame: track (against the model set once one has been accepted) ->
 * back-project + fuse -> the map ray-cast into a depth image from the pose just tracked -> accepted as the next frame's model if
 * at least min_coverage (0..1) of its pixels met the map, else the next frame is tracked against the previous frame's maps ->
 * cone-traced view (rows of the LAST frame in d_image).  Sequential on caller_stream and BLOCKING (one 4-byte coverage readback
 * per frame); *models_used (optional) = frames whose model was accepted.  Same call rules as svoslam_runner_run (one caller stream
 * per runner, timestamps newer than the camera's latest).  Leaves frame-to-model tracking switched on in the camera and the last
 * accepted model set, so that a second call continues the sequence as one longer call would; a later svoslam_runner_run on the
 * same runner -- the loop without a model refresh -- clears both when it starts and tracks frame to frame.  One replica only. */
int svoslam_runner_run_model(svoslam_runner *runner, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs,
                             const long long *timestamps, const float *views, int32_t n, uint8_t *d_image, int32_t row_first,
                             int32_t rows, unsigned long long *d_steps, float min_coverage, int32_t *models_used, void *caller_stream);
/* the bounding box the loop computed for the last frame enqueued (computePointCloudBoundingBox, main.cpp:43):
 * {min xyz, max xyz, any point}.  Blocking. */
int svoslam_runner_bbox(svoslam_runner *runner, float h_bbox7[7]);
/* The same loop for ONE RANK of a frame-sharded session (frames are tracked in parallel, one process
 * per GPU, each with a full replica of the map).  The poses come from svoslam_camera_apply_delta(d_deltas[i]) -- n device
 * pointers to SVOSLAM_DELTA_FLOATS floats, produced by svoslam_camera_pair_delta on whichever rank tracked frame i and
 * all-gathered; delta_events (optional; n hipEvent_t, NULL entries allowed): what the pose stream waits for before it reads
 * d_deltas[i] -- so the runner's camera never sees a depth image; EVERY frame is back-projected, planned and committed
 * (the replicas stay byte-identical), and only the frames with march[i] != 0 (march = NULL: all) are ray-marched, frame i
 * into d_images[i].  Entry 0 of d_deltas is ignored for a camera's first frame.  Needs the default one-replica,
 * direct-commit schedule. */
int svoslam_runner_run_sharded(svoslam_runner *runner, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs,
                               const long long *timestamps, const float *views, int32_t n, const float *const *d_deltas,
                               void *const *delta_events, const uint8_t *march, uint8_t *const *d_images, int32_t row_first,
                               int32_t rows, unsigned long long *d_steps, void *caller_stream);
/* The same with the SORT sharded as well (round 3): the rank that owns frame i has back-projected and sorted it
 * (svoslam_svo_fuse_sort_frame with the pose its own chain of svoslam_camera_apply_delta gives, svoslam_svo_fuse_export_sorted)
 * and the sorted arrays have been all-gathered: d_sorted_keys[i] / d_sorted_idx[i] (all n entries required),
 * sorted_events[i] (optional) = what the plan waits for before it reads them.  This rank then only plans and commits every
 * frame and ray-marches its own; the bounding box of svoslam_runner_bbox is not computed in this form. */
int svoslam_runner_run_sharded_presorted(svoslam_runner *runner, const uint16_t *const *d_depths, const uint8_t *const *d_rgbs,
                                         const long long *timestamps, const float *views, int32_t n, const float *const *d_deltas,
                                         void *const *delta_events, const uint8_t *march, uint8_t *const *d_images,
                                         const unsigned long long *const *d_sorted_keys, const uint32_t *const *d_sorted_idx,
                                         void *const *sorted_events, int32_t row_first, int32_t rows, unsigned long long *d_steps,
                                         void *caller_stream);

/* ------------------------------------------------------------------------
 * Timing hook: replaces startTiming/stopTiming (include/octree_slam/
 * timing_utils.h:5-10, src/timing_utils.cu:11-32) with hipEvents on `stream`.
 * ---------------------------------------------------------------------- */
int svoslam_timer_start(void *stream);
int svoslam_timer_stop(void *stream, float *h_ms); /* blocking */

/* ------------------------------------------------------------------------
 * One-shot peer-to-peer exchange of small records between the ranks of one node (csrc/mailbox.hip; SURVEY 5 "comm
 * backend" / 8e).  Replaces the per-iteration 168-byte device-to-host copy of the reference's ICP reduction
 * (src/sensor/localization_kernels.cu:318-325) across GPUs: every rank stores its record into each peer's inbox (device
 * memory mapped by the peers: hipIpc handles between processes, plain pointers inside one) and polls its own; sums are
 * formed in rank order, so every rank gets the same bits.  Two short launches on the caller's stream per collective, no
 * host round trip, no communication library.  All ranks must issue the collectives of a mailbox in the same order.
 * ---------------------------------------------------------------------- */
typedef struct svoslam_mailbox svoslam_mailbox;
int svoslam_mailbox_create(svoslam_mailbox **mailbox, int32_t rank, int32_t world);
int svoslam_mailbox_destroy(svoslam_mailbox *mailbox);
int svoslam_mailbox_handle(svoslam_mailbox *mailbox, void *handle64);                 /* 64 bytes for the other processes */
int svoslam_mailbox_connect(svoslam_mailbox *mailbox, const void *handles);           /* world x 64 bytes, any order of arrival */
int svoslam_mailbox_connect_local(svoslam_mailbox *mailbox, svoslam_mailbox *const *all);  /* peers inside this process */
int svoslam_mailbox_all_gather(svoslam_mailbox *mailbox, const void *d_src, int32_t bytes, void *d_dst, void *stream);
int svoslam_mailbox_all_reduce_f64(svoslam_mailbox *mailbox, double *d_values, int32_t count, void *stream);
/* the two halves of a collective, for callers that drive several mailboxes from one stream (post them all, then collect):
 * _post stages d_src and stores it into every inbox (next epoch), _collect waits for every rank's record of that epoch */
int svoslam_mailbox_post(svoslam_mailbox *mailbox, const void *d_src, int32_t bytes, void *stream);
int svoslam_mailbox_collect(svoslam_mailbox *mailbox, void *d_dst, int32_t bytes, int32_t reduce_f64, void *stream);
int svoslam_mailbox_failed(svoslam_mailbox *mailbox, int32_t *failed);
/* polls per granule before a wait gives up (default 2^24, about a second).  A wait that gives up writes all-ones granules (NaN
 * as binary32 / binary64) in place of the missing record and sets the sticky flag svoslam_mailbox_failed() reports. */
int svoslam_mailbox_set_wait_limit(svoslam_mailbox *mailbox, uint32_t polls);

#ifdef __cplusplus
}
#endif
#endif /* SVOSLAM_H_ */
