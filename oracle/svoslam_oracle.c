/*
 * svoslam_oracle.c -- CPU restatement of the dkotfis/Octree-SLAM hot path.
 * TEST INFRASTRUCTURE ONLY (see svoslam_oracle.h).  Parity unpinned by the
 * reference (it ships no tests); pinned by SURVEY.md Appendix C KATs.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 */
#include "svoslam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* small helpers                                                             */
/* ======================================================================== */

static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* isfinite on the bit pattern (no libm dependence) */
static inline int finitef_(float f) { return (f2bits(f) & 0x7F800000u) != 0x7F800000u; }

static int cmp_key(const void *a, const void *b) {
  octkey x = *(const octkey *)a, y = *(const octkey *)b;
  return (x > y) - (x < y);
}

/* ======================================================================== */
/* keys  (src/world/svo/svo.cu:33-106)                                       */
/* ======================================================================== */

/* svo.cu:33-66.  Q1: the finite test reads x, z, z (never y). */
octkey ora_compute_key(const float p[3], const float center_in[3], int tree_depth, float edge_length) {
  if (!finitef_(p[0]) || !finitef_(p[2]) || !finitef_(p[2])) return 1;
  float c[3] = {center_in[0], center_in[1], center_in[2]};
  octkey morton = 1;
  for (int i = 0; i < tree_depth; i++) {
    morton = morton << 3;
    int x = p[0] > c[0];
    int y = p[1] > c[1];
    int z = p[2] > c[2];
    morton += (x + 2 * y + 4 * z);
    edge_length /= 2.0f;
    c[0] += edge_length * (x ? 1 : -1);
    c[1] += edge_length * (y ? 1 : -1);
    c[2] += edge_length * (z ? 1 : -1);
  }
  return morton;
}

/* svo.cu:68-78 computes (bit_length(key)-1)/3 through a 31-bit mask (valid for
 * depth <= 10).  The oracle uses the same formula on all 64 bits so
 * that depth 11..21 is defined; identical for every key the reference handles. */
int ora_depth_from_key(octkey key) {
  if (key <= 0) return 0;
  int bl = 64 - __builtin_clzll((unsigned long long)key);
  return (bl - 1) / 3;
}

/* svo.cu:80-82 */
static inline int get_value_from_key(octkey key, int depth) { return (int)((key >> (3 * depth)) & 0x7); }

/* svo.cu:84-90 : pop the top 3-bit group, keep the leading 1 */
int ora_get_first_value_and_shift_down(octkey *key) {
  int depth = ora_depth_from_key(*key);
  int value = get_value_from_key(*key, depth - 1);
  *key -= ((octkey)(8 + value) << (3 * (depth - 1)));
  *key += ((octkey)1 << (3 * (depth - 1)));
  return value;
}

/* svo.cu:92-106 */
void ora_compute_keys(const float *pts, int stride, int n, int max_depth, const float center[3],
                      float edge_length, octkey *keys) {
  for (int i = 0; i < n; i++) keys[i] = ora_compute_key(pts + (size_t)i * stride, center, max_depth, edge_length);
}

/* ======================================================================== */
/* split planning (svo.cu:108-237)                                           */
/* ======================================================================== */

/* svo.cu:108-142.  Q3: the loop condition is r_key >= 15, so the last level
 * is examined only when its octant is 7. */
static void split_key(octkey key, const uint32_t *octree, octkey *left, octkey *right) {
  octkey r_key = key, l_key = -1, temp_key = 1;
  int node_idx = 0;
  while (r_key >= 15) {
    int value = ora_get_first_value_and_shift_down(&r_key);
    temp_key = (temp_key << 3) + value;
    node_idx += value;
    if (!(octree[2 * (size_t)node_idx] & ORA_FLAG_CHILDREN)) {
      l_key = temp_key;
      break;
    }
    node_idx = (int)(octree[2 * (size_t)node_idx] & ORA_CHILD_MASK);
  }
  *left = l_key;
  *right = r_key;
}

/* svo.cu:144-171 */
static void right_to_left_shift(octkey *left, octkey *right) {
  if (*left == -1 || *right == 1) { *left = -1; return; }
  octkey r_key = *right;
  int moved = ora_get_first_value_and_shift_down(&r_key);
  *right = r_key;
  if (*right == 1) { *left = -1; return; }
  *left = (*left << 3) + moved;
}

/* svo.cu:179-237.  thrust::remove_if(negative) / sort / unique per pass.
 * `negative` truncates to int in the reference (Q2); the oracle tests the
 * 64-bit sign, identical for depth <= 10. */
static int prepass(const octkey *keys, int n, int max_depth, const uint32_t *octree, octkey **codes,
                   int *code_sizes) {
  int num_split = 0;
  octkey *left = (octkey *)malloc(sizeof(octkey) * (size_t)(n > 0 ? n : 1));
  octkey *right = (octkey *)malloc(sizeof(octkey) * (size_t)(n > 0 ? n : 1));
  octkey *tmp = (octkey *)malloc(sizeof(octkey) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) split_key(keys[i], octree, &left[i], &right[i]);
  for (int i = 0; i < max_depth; i++) { codes[i] = NULL; code_sizes[i] = 0; }
  for (int i = 0; i < max_depth; i++) {
    int size = 0;
    for (int j = 0; j < n; j++)
      if (!(left[j] < 0)) tmp[size++] = left[j];
    if (size == 0) break;
    qsort(tmp, (size_t)size, sizeof(octkey), cmp_key);
    int u = 1;
    for (int j = 1; j < size; j++)
      if (tmp[j] != tmp[u - 1]) tmp[u++] = tmp[j];
    size = u;
    code_sizes[i] = size;
    codes[i] = (octkey *)malloc(sizeof(octkey) * (size_t)size);
    memcpy(codes[i], tmp, sizeof(octkey) * (size_t)size);
    num_split += size;
    for (int j = 0; j < n; j++) right_to_left_shift(&left[j], &right[j]);
  }
  free(left); free(right); free(tmp);
  return num_split;
}

int ora_prepass(const octkey *keys, int n, int max_depth, const uint32_t *octree, int *pass_sizes,
                octkey **codes_out) {
  octkey **codes = (octkey **)malloc(sizeof(octkey *) * (size_t)max_depth);
  int total = prepass(keys, n, max_depth, octree, codes, pass_sizes);
  if (codes_out) {
    *codes_out = (octkey *)malloc(sizeof(octkey) * (size_t)(total > 0 ? total : 1));
    int o = 0;
    for (int i = 0; i < max_depth; i++)
      for (int j = 0; j < pass_sizes[i]; j++) (*codes_out)[o++] = codes[i][j];
  }
  for (int i = 0; i < max_depth; i++) free(codes[i]);
  free(codes);
  return total;
}

/* walk a key from the root: returns node index, *child_idx = its child pointer.
 * The loop shared by svo.cu:255-263, 308-316, 352-364, 404-412, 551-571. */
static int walk_key(const uint32_t *octree, octkey key, int *child_idx_out) {
  int node_idx = 0, child_idx = 0;
  while (key != 1) {
    node_idx = child_idx + ora_get_first_value_and_shift_down(&key);
    child_idx = (int)(octree[2 * (size_t)node_idx] & ORA_CHILD_MASK);
  }
  if (child_idx_out) *child_idx_out = child_idx;
  return node_idx;
}

/* svo.cu:239-289 */
static void expand_tree_at_keys(octkey **codes, const int *sizes, int depth, uint32_t *octree, int *num_nodes) {
  for (int i = 0; i < depth; i++) {
    if (sizes[i] == 0) break;
    for (int index = 0; index < sizes[i]; index++) {
      octkey key = codes[i][index];
      if (key == 1) continue;
      int node_idx = walk_key(octree, key, NULL);
      int new_node = *num_nodes + 8 * index;
      octree[2 * (size_t)node_idx] = (1u << 30) + ((uint32_t)new_node & ORA_CHILD_MASK);
      for (int off = 0; off < 8; off++) {
        octree[2 * (size_t)(new_node + off)] = 0;
        octree[2 * (size_t)(new_node + off) + 1] = 127u << 24;
      }
    }
    *num_nodes += 8 * sizes[i];
  }
}

/* svo.cu:24-31 */
static void init_octree(ora_pool *pool) {
  pool->data = (uint32_t *)calloc(16, sizeof(uint32_t));
  pool->size = 8;
}

void ora_pool_free(ora_pool *p) {
  free(p->data);
  p->data = NULL;
  p->size = 0;
}

/* the pool becomes a copy of `num_nodes` nodes given as words (a map fused elsewhere: bench.py's cpu_baseline continues
 * the GPU's map, whose words equal this oracle's own -- tests/test_gpu_fullsize.py); not a reference function */
int ora_pool_load_words(ora_pool *p, const uint32_t *words, int num_nodes) {
  if (num_nodes < 8 || !words) return -1;
  uint32_t *d = (uint32_t *)realloc(p->data, sizeof(uint32_t) * 2 * (size_t)num_nodes);
  if (!d) return -1;
  memcpy(d, words, sizeof(uint32_t) * 2 * (size_t)num_nodes);
  p->data = d;
  p->size = num_nodes;
  return 0;
}

/* the blend of svo.cu:366-381 (Color256) : values are exact in binary32, so the
 * result is floor((new*(256-a) + cur*a)/256) regardless of FMA contraction. */
static uint32_t blend_color256(uint32_t current_value, const uint8_t rgb[3]) {
  short current_alpha = (short)(current_value >> 24);
  uint8_t cr = current_value & 0xFF, cg = (current_value >> 8) & 0xFF, cb = (current_value >> 16) & 0xFF;
  float f1 = (1 - ((float)current_alpha / 256.0f));
  float f2 = (float)current_alpha / 256.0f;
  uint8_t nr = (uint8_t)(rgb[0] * f1 + cr * f2);
  uint8_t ng = (uint8_t)(rgb[1] * f1 + cg * f2);
  uint8_t nb = (uint8_t)(rgb[2] * f1 + cb * f2);
  int a = current_alpha + 2 < 255 ? current_alpha + 2 : 255;
  return (uint32_t)((int)nr) + ((uint32_t)((int)ng) << 8) + ((uint32_t)((int)nb) << 16) + ((uint32_t)a << 24);
}

/* the blend of svo.cu:318-332 (vec4 * 256).  Q21: 1.0 -> 256 carries into the
 * next channel through the integer adds. */
static uint32_t blend_vec4(uint32_t current_value, const float rgba[4]) {
  float nr = rgba[0] * 256.0f, ng = rgba[1] * 256.0f, nb = rgba[2] * 256.0f;
  int current_alpha = (int)(current_value >> 24);
  int cr = current_value & 0xFF, cg = (current_value >> 8) & 0xFF, cb = (current_value >> 16) & 0xFF;
  float f1 = 1 - ((float)current_alpha / 256.0f);
  float f2 = (float)current_alpha / 256.0f;
  nr = nr * f1 + (float)cr * f2;
  ng = ng * f1 + (float)cg * f2;
  nb = nb * f1 + (float)cb * f2;
  int a = current_alpha + 2 < 255 ? current_alpha + 2 : 255;
  return (uint32_t)((int)nr) + ((uint32_t)((int)ng) << 8) + ((uint32_t)((int)nb) << 16) + ((uint32_t)a << 24);
}

/* svo.cu:291-382.  Deterministic resolution of the duplicate-key race (Q8):
 * every thread reads the pre-kernel leaf value; among the points that map to
 * one leaf the LOWEST index is the surviving writer.  color_of[i] gives the
 * index into the colour array used by element i (identity for the cloud path;
 * for the voxel-grid path element i is the i-th SORTED key and uses colour i,
 * Q20). */
static void fill_nodes(const octkey *keys, int n, const void *values, int is_vec4, uint32_t *octree, int num_nodes) {
  int *owner = (int *)malloc(sizeof(int) * (size_t)num_nodes);
  for (int i = 0; i < num_nodes; i++) owner[i] = -1;
  int *leaf = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    leaf[i] = -1;
    if (keys[i] == 1) continue;
    int node = walk_key(octree, keys[i], NULL);
    leaf[i] = node;
    if (owner[node] < 0) owner[node] = i;
  }
  for (int i = 0; i < n; i++) {
    if (leaf[i] < 0 || owner[leaf[i]] != i) continue;
    uint32_t cur = octree[2 * (size_t)leaf[i] + 1];
    octree[2 * (size_t)leaf[i] + 1] = is_vec4 ? blend_vec4(cur, (const float *)values + 4 * (size_t)i)
                                             : blend_color256(cur, (const uint8_t *)values + 3 * (size_t)i);
  }
  free(owner); free(leaf);
}

/* svo.cu:384-441.  Q5: the occupancy test `(child_val >> 24) & 0xFF == 0`
 * parses as `& (0xFF == 0)` and never skips, so the mean always divides by 8.
 * Sums of <= 8 bytes and /8 are exact in binary32. */
static uint32_t average_children_value(const uint32_t *octree, int child_idx) {
  float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f;
  int num_occ = 0;
  for (int i = 0; i < 8; i++) {
    int child_val = (int)octree[2 * (size_t)(child_idx + i) + 1];
    r += (float)(child_val & 0xFF);
    g += (float)((child_val >> 8) & 0xFF);
    b += (float)((child_val >> 16) & 0xFF);
    float ca = (float)((child_val >> 24) & 0xFF);
    a = a > ca ? a : ca;
    num_occ++;
  }
  if (num_occ > 0) { r = r / (float)num_occ; g = g / (float)num_occ; b = b / (float)num_occ; }
  return (uint32_t)((int)r) + ((uint32_t)((int)g) << 8) + ((uint32_t)((int)b) << 16) + ((uint32_t)((int)a) << 24);
}

/* svo.cu:443-465.  One pass = one kernel launch; all threads of a launch read
 * the pre-launch pool ("snapshot"): this only matters for the final pass,
 * whose threads all read root children 0..7 and write node 0 (Q6). */
static void mipmap_nodes(octkey *keys, int num_keys, uint32_t *octree) {
  int *nodes = (int *)malloc(sizeof(int) * (size_t)(num_keys > 0 ? num_keys : 1));
  uint32_t *vals = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(num_keys > 0 ? num_keys : 1));
  for (;;) {
    int m = 0;
    for (int i = 0; i < num_keys; i++) /* remove_if(depth_is_zero) */
      if (ora_depth_from_key(keys[i]) != 0) keys[m++] = keys[i];
    num_keys = m;
    if (num_keys <= 0) break;
    if (num_keys > 100000) { /* thrust::unique: adjacent only */
      int u = 1;
      for (int i = 1; i < num_keys; i++)
        if (keys[i] != keys[u - 1]) keys[u++] = keys[i];
      num_keys = u;
    }
    for (int i = 0; i < num_keys; i++) {
      octkey key = keys[i] >> 3;
      keys[i] = key;
      int child_idx;
      nodes[i] = walk_key(octree, key, &child_idx);
      vals[i] = average_children_value(octree, child_idx);
    }
    for (int i = 0; i < num_keys; i++) octree[2 * (size_t)nodes[i] + 1] = vals[i];
  }
  free(nodes); free(vals);
}

static void svo_insert(octkey *keys, int n, int max_depth, ora_pool *pool, const void *values, int is_vec4) {
  octkey **codes = (octkey **)malloc(sizeof(octkey *) * (size_t)max_depth);
  int *code_sizes = (int *)malloc(sizeof(int) * (size_t)max_depth);
  int new_nodes = prepass(keys, n, max_depth, pool->data, codes, code_sizes);
  pool->data = (uint32_t *)realloc(pool->data, sizeof(uint32_t) * 2 * ((size_t)pool->size + 8 * (size_t)new_nodes));
  expand_tree_at_keys(codes, code_sizes, max_depth, pool->data, &pool->size);
  for (int i = 0; i < max_depth; i++) free(codes[i]);
  free(codes); free(code_sizes);
  fill_nodes(keys, n, values, is_vec4, pool->data, pool->size);
  mipmap_nodes(keys, n, pool->data);
}

/* svo.cu:642-696 */
void ora_svo_from_point_cloud(const float *points, const uint8_t *colors, int n, int max_depth, ora_pool *pool,
                              const float center[3], float edge_length) {
  if (pool->size == 0) init_octree(pool);
  octkey *keys = (octkey *)malloc(sizeof(octkey) * (size_t)(n > 0 ? n : 1));
  ora_compute_keys(points, 3, n, max_depth, center, edge_length, keys);
  svo_insert(keys, n, max_depth, pool, colors, 0);
  free(keys);
}

/* svo.cu:584-640.  Q20: thrust::sort permutes the keys only; colour i stays
 * paired with the i-th sorted key. */
void ora_svo_from_voxel_grid(const float *centers, const float *colors, int n, int max_depth, ora_pool *pool,
                             const float center[3], float edge_length) {
  if (pool->size == 0) init_octree(pool);
  octkey *keys = (octkey *)malloc(sizeof(octkey) * (size_t)(n > 0 ? n : 1));
  ora_compute_keys(centers, 4, n, max_depth, center, edge_length, keys);
  qsort(keys, (size_t)n, sizeof(octkey), cmp_key);
  svo_insert(keys, n, max_depth, pool, colors, 1);
  free(keys);
}

/* svo.cu:498-582, 699-745.  BFS, order-preserving compaction per level. */
int ora_extract_voxel_grid(const ora_pool *pool, int max_depth, const float center_in[3], float edge_length,
                           float **centers, float **colors) {
  const uint32_t *octree = pool->data;
  int num = 1;
  octkey *list = (octkey *)malloc(sizeof(octkey));
  list[0] = 1;
  for (int lvl = 0; lvl < max_depth; lvl++) {
    octkey *next = (octkey *)malloc(sizeof(octkey) * 8 * (size_t)(num > 0 ? num : 1));
    int m = 0;
    for (int index = 0; index < num; index++) {
      octkey key = list[index], temp_key = key;
      int has_children = 1;
      int pointer = 0;
      while (temp_key != 1) {
        pointer += ora_get_first_value_and_shift_down(&temp_key);
        has_children = (octree[2 * (size_t)pointer] & ORA_FLAG_CHILDREN) != 0;
        pointer = (int)(octree[2 * (size_t)pointer] & ORA_CHILD_MASK);
      }
      for (int i = 0; i < 8; i++) {
        if (has_children) {
          uint32_t val2 = octree[2 * (size_t)(pointer + i) + 1];
          if (((val2 >> 24) & 0xFF) > 127) next[m++] = (key << 3) + i;
        }
      }
    }
    free(list);
    list = next;
    num = m;
  }
  *centers = (float *)malloc(sizeof(float) * 4 * (size_t)(num > 0 ? num : 1));
  *colors = (float *)malloc(sizeof(float) * 4 * (size_t)(num > 0 ? num : 1));
  for (int idx = 0; idx < num; idx++) { /* svo.cu:538-582 */
    octkey key = list[idx];
    float c[3] = {center_in[0], center_in[1], center_in[2]};
    float el = edge_length;
    int node_idx = 0, child_idx = 0;
    while (key != 1) {
      int pos = ora_get_first_value_and_shift_down(&key);
      node_idx = child_idx + pos;
      child_idx = (int)(octree[2 * (size_t)node_idx] & ORA_CHILD_MASK);
      int x = pos & 0x1, y = pos & 0x2, z = pos & 0x4;
      el /= 2.0f;
      c[0] += el * (x ? 1 : -1);
      c[1] += el * (y ? 1 : -1);
      c[2] += el * (z ? 1 : -1);
    }
    uint32_t val = octree[2 * (size_t)node_idx + 1];
    float *ce = *centers + 4 * (size_t)idx, *co = *colors + 4 * (size_t)idx;
    ce[0] = c[0]; ce[1] = c[1]; ce[2] = c[2]; ce[3] = 1.0f;
    co[0] = (float)(val & 0xFF) / 255.0f;
    co[1] = (float)((val >> 8) & 0xFF) / 255.0f;
    co[2] = (float)((val >> 16) & 0xFF) / 255.0f;
    co[3] = (float)((val >> 24) & 0xFF) / 255.0f;
  }
  free(list);
  return num;
}

/* ======================================================================== */
/* mat4, glm 0.9.5.4 semantics (external/include/glm/detail/type_mat4x4.inl, */
/* gtc/matrix_transform.inl).  Column-major: m[4*col + row].                 */
/* ======================================================================== */

#define M(m, c, r) ((m)[4 * (c) + (r)])

void ora_mat4_identity(float m[16]) {
  for (int i = 0; i < 16; i++) m[i] = 0.0f;
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}

/* type_mat4x4.inl:753-775 : Result[c] = ((A0*B[c][0] + A1*B[c][1]) + A2*B[c][2]) + A3*B[c][3] */
void ora_mat4_mul(const float a[16], const float b[16], float out[16]) {
  float r[16];
  for (int c = 0; c < 4; c++)
    for (int row = 0; row < 4; row++)
      r[4 * c + row] = ((M(a, 0, row) * M(b, c, 0) + M(a, 1, row) * M(b, c, 1)) + M(a, 2, row) * M(b, c, 2)) +
                       M(a, 3, row) * M(b, c, 3);
  memcpy(out, r, sizeof(r));
}

/* type_mat4x4.inl:651-687 : (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
static void mat4_mul_vec4(const float m[16], const float v[4], float out[4]) {
  float r[4];
  for (int row = 0; row < 4; row++)
    r[row] = (M(m, 0, row) * v[0] + M(m, 1, row) * v[1]) + (M(m, 2, row) * v[2] + M(m, 3, row) * v[3]);
  memcpy(out, r, sizeof(r));
}

/* type_mat4x4.inl:699-710 : row vector times matrix */
static void vec4_mul_mat4(const float v[4], const float m[16], float out[4]) {
  float r[4];
  for (int i = 0; i < 4; i++)
    r[i] = ((M(m, i, 0) * v[0] + M(m, i, 1) * v[1]) + M(m, i, 2) * v[2]) + M(m, i, 3) * v[3];
  memcpy(out, r, sizeof(r));
}

/* type_mat4x4.inl:477-534 */
void ora_mat4_inverse(const float m[16], float out[16]) {
  float Coef00 = M(m,2,2) * M(m,3,3) - M(m,3,2) * M(m,2,3);
  float Coef02 = M(m,1,2) * M(m,3,3) - M(m,3,2) * M(m,1,3);
  float Coef03 = M(m,1,2) * M(m,2,3) - M(m,2,2) * M(m,1,3);
  float Coef04 = M(m,2,1) * M(m,3,3) - M(m,3,1) * M(m,2,3);
  float Coef06 = M(m,1,1) * M(m,3,3) - M(m,3,1) * M(m,1,3);
  float Coef07 = M(m,1,1) * M(m,2,3) - M(m,2,1) * M(m,1,3);
  float Coef08 = M(m,2,1) * M(m,3,2) - M(m,3,1) * M(m,2,2);
  float Coef10 = M(m,1,1) * M(m,3,2) - M(m,3,1) * M(m,1,2);
  float Coef11 = M(m,1,1) * M(m,2,2) - M(m,2,1) * M(m,1,2);
  float Coef12 = M(m,2,0) * M(m,3,3) - M(m,3,0) * M(m,2,3);
  float Coef14 = M(m,1,0) * M(m,3,3) - M(m,3,0) * M(m,1,3);
  float Coef15 = M(m,1,0) * M(m,2,3) - M(m,2,0) * M(m,1,3);
  float Coef16 = M(m,2,0) * M(m,3,2) - M(m,3,0) * M(m,2,2);
  float Coef18 = M(m,1,0) * M(m,3,2) - M(m,3,0) * M(m,1,2);
  float Coef19 = M(m,1,0) * M(m,2,2) - M(m,2,0) * M(m,1,2);
  float Coef20 = M(m,2,0) * M(m,3,1) - M(m,3,0) * M(m,2,1);
  float Coef22 = M(m,1,0) * M(m,3,1) - M(m,3,0) * M(m,1,1);
  float Coef23 = M(m,1,0) * M(m,2,1) - M(m,2,0) * M(m,1,1);
  float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
  float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
  float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
  float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
  float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
  float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
  float Vec0[4] = {M(m,1,0), M(m,0,0), M(m,0,0), M(m,0,0)};
  float Vec1[4] = {M(m,1,1), M(m,0,1), M(m,0,1), M(m,0,1)};
  float Vec2[4] = {M(m,1,2), M(m,0,2), M(m,0,2), M(m,0,2)};
  float Vec3[4] = {M(m,1,3), M(m,0,3), M(m,0,3), M(m,0,3)};
  static const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
  float inv[16];
  for (int i = 0; i < 4; i++) {
    float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
    float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
    float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
    float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
    inv[4 * 0 + i] = Inv0 * SignA[i];
    inv[4 * 1 + i] = Inv1 * SignB[i];
    inv[4 * 2 + i] = Inv2 * SignA[i];
    inv[4 * 3 + i] = Inv3 * SignB[i];
  }
  float Dot0[4] = {M(m,0,0) * inv[0], M(m,0,1) * inv[4], M(m,0,2) * inv[8], M(m,0,3) * inv[12]};
  float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
  float ood = 1.0f / Dot1;
  for (int i = 0; i < 16; i++) out[i] = inv[i] * ood;
}

/* gtc/matrix_transform.inl:35-45 : Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
void ora_mat4_translate(const float m[16], const float v[3], float out[16]) {
  float r[16];
  memcpy(r, m, sizeof(r));
  for (int row = 0; row < 4; row++)
    r[12 + row] = ((M(m, 0, row) * v[0] + M(m, 1, row) * v[1]) + M(m, 2, row) * v[2]) + M(m, 3, row);
  memcpy(out, r, sizeof(r));
}

/* Deterministic sin/cos for glm::rotate (gtc/matrix_transform.inl:60-61 calls
 * cos(a), sin(a) of the host libm, whose last-ulp behaviour the reference does
 * not fix).  Evaluated in binary64 with explicit fma(): Cody-Waite reduction by
 * pi/2 and the fdlibm kernel polynomials, then rounded once to binary32.  The
 * product evaluates the same operation sequence on the device. */
void ora_sincos(float af, float *s_out, float *c_out) {
  double x = (double)af;
  double kd = rint(x * 0.63661977236758134308);
  double r = fma(kd, -1.57079632673412561417e+00, x); /* pio2_1 (33 bits) */
  r = fma(kd, -6.07710050650619224932e-11, r);        /* pio2_1t */
  double z = r * r;
  /* fdlibm k_sin.c */
  double sp = 1.58969099521155010221e-10;
  sp = fma(sp, z, -2.50507602534068634195e-08);
  sp = fma(sp, z, 2.75573137070700676789e-06);
  sp = fma(sp, z, -1.98412698298579493134e-04);
  sp = fma(sp, z, 8.33333333332248946124e-03);
  sp = fma(sp, z, -1.66666666666666324348e-01);
  double sn = fma(r * z, sp, r);
  /* fdlibm k_cos.c */
  double cp = -1.13596475577881948265e-11;
  cp = fma(cp, z, 2.08757232129817482790e-09);
  cp = fma(cp, z, -2.75573143513906633035e-07);
  cp = fma(cp, z, 2.48015872894767294178e-05);
  cp = fma(cp, z, -1.38888888888741095749e-03);
  cp = fma(cp, z, 4.16666666666666019037e-02);
  double cs = fma(z * z, cp, fma(z, -0.5, 1.0));
  long long k = (long long)kd;
  double s, c;
  switch ((int)(k & 3)) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
  }
  *s_out = (float)s;
  *c_out = (float)c;
}

/* gtc/matrix_transform.inl:47-86 (degrees API: no GLM_FORCE_RADIANS) */
void ora_mat4_rotate_deg(const float m[16], float angle, const float v[3], float out[16]) {
  float a = angle * 0.01745329251994329576923690768489f; /* glm::radians */
  float c, s;
  ora_sincos(a, &s, &c);
  float sqr = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float inv = 1.0f / sqrtf(sqr);
  float axis[3] = {v[0] * inv, v[1] * inv, v[2] * inv};
  float temp[3] = {(1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2]};
  float R[3][3];
  R[0][0] = c + temp[0] * axis[0];
  R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
  R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
  R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
  R[1][1] = c + temp[1] * axis[1];
  R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
  R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
  R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
  R[2][2] = c + temp[2] * axis[2];
  float r[16];
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 4; row++)
      r[4 * col + row] = (M(m, 0, row) * R[col][0] + M(m, 1, row) * R[col][1]) + M(m, 2, row) * R[col][2];
  for (int row = 0; row < 4; row++) r[12 + row] = M(m, 3, row);
  memcpy(out, r, sizeof(r));
}

static void normalize3(const float v[3], float out[3]) { /* func_geometric.inl:256-265 */
  float sqr = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float inv = 1.0f / sqrtf(sqr);
  out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
}
static void cross3(const float x[3], const float y[3], float out[3]) { /* func_geometric.inl cross */
  float r0 = x[1] * y[2] - y[1] * x[2];
  float r1 = x[2] * y[0] - y[2] * x[0];
  float r2 = x[0] * y[1] - y[0] * x[1];
  out[0] = r0; out[1] = r1; out[2] = r2;
}
static float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static float length3(const float a[3]) { return sqrtf(dot3(a, a)); }

/* gtc/matrix_transform.inl:416-441 */
void ora_mat4_look_at(const float eye[3], const float center[3], const float up[3], float out[16]) {
  float d[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
  float f[3], s[3], u[3], t[3];
  normalize3(d, f);
  cross3(f, up, t);
  normalize3(t, s);
  cross3(s, f, u);
  ora_mat4_identity(out);
  M(out,0,0) = s[0]; M(out,1,0) = s[1]; M(out,2,0) = s[2];
  M(out,0,1) = u[0]; M(out,1,1) = u[1]; M(out,2,1) = u[2];
  M(out,0,2) = -f[0]; M(out,1,2) = -f[1]; M(out,2,2) = -f[2];
  M(out,3,0) = -dot3(s, eye);
  M(out,3,1) = -dot3(u, eye);
  M(out,3,2) = dot3(f, eye);
}

/* ======================================================================== */
/* cone trace (src/rendering/cone_tracing_kernels.cu:24-198)                 */
/* ======================================================================== */

/* ceil(log(q)/log(2)) of cone_tracing_kernels.cu:69 evaluated exactly from the
 * binary32 exponent (the reference's float log() may differ by one level when
 * q is within an ulp of a power of two). */
static int ceil_log2_pos(float q) {
  uint32_t u = f2bits(q);
  if ((int32_t)u <= 0) return 0; /* zero, negative */
  int ex = (int)(u >> 23);
  uint32_t man = u & 0x7FFFFFu;
  if (ex == 255) return 128;
  if (ex == 0) { /* subnormal */
    int hb = 31 - __builtin_clz(man);
    return (hb - 149) + ((man & (man - 1)) != 0);
  }
  return (ex - 127) + (man != 0);
}

/* float -> uint8_t conversion of cone_tracing_kernels.cu:110-112,133-135:
 * truncate toward zero to a 32-bit unsigned (negative and NaN give 0, as the
 * PTX cvt.rzi.u32.f32 the CUDA compiler emits does), keep the low 8 bits. */
static uint8_t f2u8(float f) {
  if (!(f > 0.0f)) return 0;
#ifdef ORA_VARIANT_SAT_U8
  /* sensitivity variant (oracle/Makefile `variants`, tests/test_oracle_compiler_choices.py): the other reading of the float ->
   * uint8_t conversion, a saturating cvt.rzi.u8.f32 (resolution R9 takes the u32 conversion + low byte) */
  if (f >= 255.0f) return 0xFF;
  return (uint8_t)f;
#else
  if (f >= 4294967296.0f) return 0xFF;
  return (uint8_t)((uint32_t)f & 0xFFu);
#endif
}

#define ORA_MAX_RANGE 10.0f
#define ORA_START_DIST 0.002f
#define ORA_MAX_STEPS (1 << 20)

int64_t ora_cone_trace_svo(uint8_t *pos, int w, int h, float fov, const float view[16], const uint32_t *octree,
                           const float center[3], float size, int mode, int64_t *levels_descended) {
  float inv[16];
  ora_mat4_inverse(view, inv);
  float o4[4], v0[4] = {0, 0, 0, 1}, vx[4] = {-1, 0, 0, 0}, vy[4] = {0, -1, 0, 0}, xd4[4], yd4[4];
  mat4_mul_vec4(inv, v0, o4);
  mat4_mul_vec4(inv, vx, xd4);
  mat4_mul_vec4(inv, vy, yd4);
  const float origin[3] = {o4[0], o4[1], o4[2]};
  const float x_dir[3] = {xd4[0], xd4[1], xd4[2]}, y_dir[3] = {yd4[0], yd4[1], yd4[2]};
  const float res_x = (float)w, res_y = (float)h;
  const float pix_scale = tanf(fov * 3.14159f / 180.0f) / res_y; /* :171 */
  int64_t total_steps = 0, total_levels = 0;
  const float neg_y[3] = {-y_dir[0], -y_dir[1], -y_dir[2]};
  float fwd[3];
  cross3(x_dir, neg_y, fwd);
  for (int idx = 0; idx < w * h; idx++) {
    /* createRays :29-51 */
    int px = idx % w, py = idx / w;
    float magx = ((float)px - res_x / 2.0f) / 532.57f;
    float magy = ((float)py - res_y / 2.0f) / 531.54f;
    float dir[3], ray[3];
    for (int k = 0; k < 3; k++) dir[k] = ((magx * x_dir[k]) + (magy * y_dir[k])) + fwd[k];
    normalize3(dir, ray);
    for (int k = 0; k < 3; k++) ray[k] = ORA_START_DIST * ray[k];
    uint8_t value[4] = {0, 0, 0, 0}; /* cudaMemset :180 */
    uint8_t outp[4] = {0, 0, 0, 0};
    for (int step = 0; step < ORA_MAX_STEPS; step++) {
      total_steps++;
      if (mode == ORA_RENDER_REFERENCE) value[0] = value[1] = value[2] = value[3] = 0; /* pos[index] is still 0 (Q9) */
      float target[3] = {origin[0] + ray[0], origin[1] + ray[1], origin[2] + ray[2]};
      float ray_len = length3(ray);
      float pix_size = ray_len * pix_scale;
      int depth = ceil_log2_pos((float)(size / pix_size));
      int node_idx = 0, child_idx = 0;
      float temp_size = size;
      float c[3] = {center[0], center[1], center[2]};
      for (int i = 0; i < depth; i++) {
        int x = target[0] > c[0], y = target[1] > c[1], z = target[2] > c[2];
        int child = x + 2 * y + 4 * z;
        node_idx = child_idx + child;
        total_levels++;
        uint32_t w0 = octree[2 * (size_t)node_idx];
        if (!(w0 & ORA_FLAG_CHILDREN)) { depth = i + 1; break; }
        child_idx = (int)(w0 & ORA_CHILD_MASK);
        temp_size /= 2.0f;
        c[0] += temp_size * (x ? 1 : -1);
        c[1] += temp_size * (y ? 1 : -1);
        c[2] += temp_size * (z ? 1 : -1);
      }
      uint32_t oct_val = octree[2 * (size_t)node_idx + 1];
      /* :108 `max(0, (oct_val >> 24) - 127)` is an (int, unsigned) overload that
       * returns the unsigned operand: alpha = A - 127 as a signed int, unclamped. */
#ifdef ORA_VARIANT_SAT_U8
      /* the same variant reads `max(0, unsigned - 127)` as a signed clamp at zero (resolution R8: the unsigned operand, unclamped) */
      int alpha = (int)(oct_val >> 24) - 127;
      if (alpha < 0) alpha = 0;
#else
      int alpha = (int)((oct_val >> 24) - 127u);
#endif
      float af = (float)alpha / 127.0f;
      value[0] = (uint8_t)(value[0] + f2u8(af * (float)(oct_val & 0xFF)));
      value[1] = (uint8_t)(value[1] + f2u8(af * (float)((oct_val >> 8) & 0xFF)));
      value[2] = (uint8_t)(value[2] + f2u8(af * (float)((oct_val >> 16) & 0xFF)));
      int retired = 0;
      if ((int)value[3] + alpha < 127) {
        value[3] = (uint8_t)(value[3] + alpha);
      } else {
        value[3] = 255;
        memcpy(outp, value, 4);
        retired = 1;
      }
      if (!retired) {
        float new_dist = size / ldexpf(1.0f, depth); /* pow(2.0f, depth) :126 */
        float s = (ray_len + new_dist) / ray_len;
        ray[0] *= s; ray[1] *= s; ray[2] *= s;
        if (length3(ray) > ORA_MAX_RANGE) {
          float sc = 127.0f / (float)value[3];
          value[0] = f2u8((float)value[0] * sc);
          value[1] = f2u8((float)value[1] * sc);
          value[2] = f2u8((float)value[2] * sc);
          value[3] = 255;
          memcpy(outp, value, 4);
          retired = 1;
        }
      }
      if (retired) break;
    }
    memcpy(pos + 4 * (size_t)idx, outp, 4);
  }
  if (levels_descended) *levels_descended = total_levels;
  return total_steps;
}

/* SURVEY 8f.3, second half: the map ray-cast into a DEPTH image, the model a frame-to-model ICP tracks against.
 * OWN SPECIFICATION -- the reference has no such function; rgbd_camera.cpp:185 only leaves the TODO ("ICP should not
 * swap, as last_frame should be updated by a different function").  Stated in the terms of the functions it sits
 * between: pixel (x, y) looks along d = ((x - w/2) / fx, (h/2 - y) / fy, 1) of the sensor frame (the direction
 * generateVertexMap gives that pixel, image_kernels.cu:24-58), carried into the map by cam_to_world (the matrix
 * main.cpp:40 applies to the vertex map, operator*(mat4, vec4)); the ray is marched exactly as coneTrace marches
 * (cone_tracing_kernels.cu:53-146: START_DIST, LOD = ceil(log2(size / (ray length x pixel scale))) with the pixel
 * scale 1 / fy, descent to the first childless node or the LOD, step = size / 2^level, MAX_RANGE) and stops at the
 * first sample whose node carries A >= 254 -- what retires a ray there.  Its distance along the optical axis,
 * ray length / |d|, becomes the pixel: rint(1000 z) as uint16 (the sensor's unit), 0 = nothing met / out of range.
 * Returns the number of march steps. */
int64_t ora_raycast_model_depth(uint16_t *depth_out, int w, int h, float fx, float fy, const float cam_to_world[16],
                                const uint32_t *octree, const float center[3], float size) {
  int64_t total_steps = 0;
  const float zero4[4] = {0.0f, 0.0f, 0.0f, 1.0f};
  float o4[4];
  mat4_mul_vec4(cam_to_world, zero4, o4);
  const float origin[3] = {o4[0], o4[1], o4[2]};
  const float pix_scale = 1.0f / fy;
  for (int idx = 0; idx < w * h; idx++) {
    const int px = idx % w, py = idx / w;
    const float dc[4] = {(float)(px - w / 2) / fx, (float)(h / 2 - py) / fy, 1.0f, 1.0f};
    float p4[4];
    mat4_mul_vec4(cam_to_world, dc, p4);
    const float dir[3] = {p4[0] - origin[0], p4[1] - origin[1], p4[2] - origin[2]};
    const float len_d = length3(dir);
    uint16_t out = 0;
    if (finitef_(len_d) && len_d > 0.0f) {
      float ray[3];
      normalize3(dir, ray);
      for (int k = 0; k < 3; k++) ray[k] = ORA_START_DIST * ray[k];
      for (int step = 0; step < ORA_MAX_STEPS; step++) {
        total_steps++;
        const float target[3] = {origin[0] + ray[0], origin[1] + ray[1], origin[2] + ray[2]};
        const float ray_len = length3(ray);
        const float pix_size = ray_len * pix_scale;
        int depth = ceil_log2_pos((float)(size / pix_size));
        int node_idx = 0, child_idx = 0;
        float temp_size = size;
        float c[3] = {center[0], center[1], center[2]};
        for (int i = 0; i < depth; i++) {
          const int x = target[0] > c[0], y = target[1] > c[1], z = target[2] > c[2];
          node_idx = child_idx + (x + 2 * y + 4 * z);
          const uint32_t w0 = octree[2 * (size_t)node_idx];
          if (!(w0 & ORA_FLAG_CHILDREN)) { depth = i + 1; break; }
          child_idx = (int)(w0 & ORA_CHILD_MASK);
          temp_size /= 2.0f;
          c[0] += temp_size * (x ? 1 : -1);
          c[1] += temp_size * (y ? 1 : -1);
          c[2] += temp_size * (z ? 1 : -1);
        }
        const uint32_t oct_val = octree[2 * (size_t)node_idx + 1];
        if ((oct_val >> 24) >= 254u) {
          const float mm = (ray_len / len_d) * 1000.0f;
          if (mm < 65535.0f) out = (uint16_t)rintf(mm);
          break;
        }
        const float new_dist = size / ldexpf(1.0f, depth);
        const float s = (ray_len + new_dist) / ray_len;
        ray[0] *= s; ray[1] *= s; ray[2] *= s;
        if (length3(ray) > ORA_MAX_RANGE) break;
      }
    }
    depth_out[idx] = out;
  }
  return total_steps;
}

/* ======================================================================== */
/* sensor (src/sensor/image_kernels.cu)                                      */
/* ======================================================================== */

/* image_kernels.cu:24-58 */
void ora_generate_vertex_map(const uint16_t *depth_pixels, float *vmap, int width, int height, float fx, float fy,
                             int img_w, int img_h) {
  for (int idx = 0; idx < width * height; idx++) {
    int x = idx % width, y = idx / width;
    int depth = depth_pixels[idx];
    float *v = vmap + 3 * (size_t)idx;
    if (depth == 0 || depth > 15000) { v[0] = v[1] = v[2] = INFINITY; continue; }
    const float milli = 0.001f;
    v[0] = ((img_w / width) * x - img_w / 2) * (float)depth / fx * milli;
    v[1] = (img_h / 2 - (img_h / height) * y) * (float)depth / fy * milli;
    v[2] = depth * milli;
  }
}

/* image_kernels.cu:104-139 */
void ora_generate_normal_map(const float *vmap, float *nmap, int width, int height) {
  for (int idx = 0; idx < width * height; idx++) {
    int x = idx % width, y = idx / width;
    float *n = nmap + 3 * (size_t)idx;
    if (x == width - 1 || y == height - 1) { n[0] = n[1] = n[2] = INFINITY; continue; }
    const float *c = vmap + 3 * (size_t)idx, *a = vmap + 3 * (size_t)(idx + 1), *b = vmap + 3 * (size_t)(idx + width);
    float v1[3] = {a[0] - c[0], a[1] - c[1], a[2] - c[2]};
    float v2[3] = {b[0] - c[0], b[1] - c[1], b[2] - c[2]};
    float cr[3];
    cross3(v1, v2, cr);
    float neg[3] = {-cr[0], -cr[1], -cr[2]};
    normalize3(neg, n);
  }
}

/* exp() for the bilateral weight.  The reference uses the __expf fast-math
 * intrinsic (image_kernels.cu:170), accurate to ~2 ulp and not reproducible off
 * NVIDIA hardware.  The oracle and the product both evaluate this fixed
 * Cody-Waite + degree-6 polynomial with explicit fmaf (<= 1 ulp), so they
 * agree bit for bit; results below the binary32 normal range flush to 0 as the
 * .ftz intrinsic does. */
static float det_expf(float x) {
  if (!(x >= -87.0f)) return 0.0f;
  if (x > 88.0f) return INFINITY;
  float kf = rintf(x * 1.44269504088896341f);
  float r = fmaf(kf, -0.693359375f, x);
  r = fmaf(kf, 2.12194440e-4f, r);
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  float e = fmaf(p, r * r, r) + 1.0f;
  return ldexpf(e, (int)kf);
}

/* image_kernels.cu:142-186.  Q12: window [x-3, min(x+4, W-1)) excludes the last
 * column/row.  sum1 accumulates with one fused multiply-add per tap in row-major
 * window order. */
void ora_bilateral_filter(const uint16_t *in, uint16_t *out, int width, int height) {
  const int kernel_size = 7;
  const float sig_spat = 0.5f / (4.5f * 4.5f);
  const float sig_dep = (float)(0.5 / (40.0f * 40.0f));
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      int value = in[y * width + x];
      int tx = x - kernel_size / 2 + kernel_size; if (tx > width - 1) tx = width - 1;
      int ty = y - kernel_size / 2 + kernel_size; if (ty > height - 1) ty = height - 1;
      float sum1 = 0, sum2 = 0;
      int y0 = y - kernel_size / 2; if (y0 < 0) y0 = 0;
      int x0 = x - kernel_size / 2; if (x0 < 0) x0 = 0;
      for (int cy = y0; cy < ty; ++cy)
        for (int cx = x0; cx < tx; ++cx) {
          int depth = in[cy * width + cx];
          float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
          /* a 32-bit int product that wraps (mul.lo.s32): written unsigned so that the wrap is defined behaviour in C */
          unsigned diff = (unsigned)(value - depth);
          float color2 = (float)(int)(diff * diff);
          float weight = det_expf(-(space2 * sig_spat + color2 * sig_dep));
          sum1 = fmaf((float)depth, weight, sum1);
          sum2 += weight;
        }
      float q = sum1 / sum2;
      /* __float2int_rn then store to uint16_t */
      int r = (q != q) ? 0 : (int)rintf(q);
      out[y * width + x] = (uint16_t)r;
    }
}

/* image_kernels.cu:236-289 ; width,height are the INPUT dims; in place via a temporary */
#define SUBSAMPLE_DEPTH_BODY(T)                                                        \
  int ow = width / 2, oh = height / 2;                                                 \
  T *tmp = (T *)malloc(sizeof(T) * (size_t)(ow * oh > 0 ? ow * oh : 1));               \
  const float sigma_depth = 40.0f * 3.0f;                                              \
  for (int y = 0; y < oh; y++)                                                         \
    for (int x = 0; x < ow; x++) {                                                     \
      const int D = 5;                                                                 \
      float center = (float)data[4 * y * ow + 2 * x];                                  \
      int tx = 2 * x - D / 2 + D; if (tx > 2 * ow - 1) tx = 2 * ow - 1;                \
      int ty = 2 * y - D / 2 + D; if (ty > 2 * oh - 1) ty = 2 * oh - 1;                \
      float sum = 0, count = 0;                                                        \
      int y0 = 2 * y - D / 2; if (y0 < 0) y0 = 0;                                      \
      int x0 = 2 * x - D / 2; if (x0 < 0) x0 = 0;                                      \
      for (int cy = y0; cy < ty; ++cy)                                                 \
        for (int cx = x0; cx < tx; ++cx) {                                             \
          float val = (float)data[2 * cy * ow + cx];                                   \
          if (fabsf(val - center) < sigma_depth) { sum += val; ++count; }              \
        }                                                                              \
      tmp[y * ow + x] = (T)((count == 0) ? 0 : sum / count);                           \
    }                                                                                  \
  memcpy(data, tmp, sizeof(T) * (size_t)(ow * oh));                                    \
  free(tmp);

void ora_subsample_depth_u16(uint16_t *data, int width, int height) { SUBSAMPLE_DEPTH_BODY(uint16_t) }
void ora_subsample_depth_f32(float *data, int width, int height) { SUBSAMPLE_DEPTH_BODY(float) }

/* image_kernels.cu:291-326 */
void ora_subsample_f32(float *data, int width, int height) {
  int ow = width / 2, oh = height / 2;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)(ow * oh > 0 ? ow * oh : 1));
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) tmp[y * ow + x] = data[4 * y * ow + 2 * x];
  memcpy(data, tmp, sizeof(float) * (size_t)(ow * oh));
  free(tmp);
}
void ora_subsample_rgb8(uint8_t *data, int width, int height) {
  int ow = width / 2, oh = height / 2;
  uint8_t *tmp = (uint8_t *)malloc((size_t)(3 * ow * oh > 0 ? 3 * ow * oh : 1));
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) memcpy(tmp + 3 * (y * ow + x), data + 3 * (4 * y * ow + 2 * x), 3);
  memcpy(data, tmp, (size_t)(3 * ow * oh));
  free(tmp);
}

/* image_kernels.cu:188-203.  Q13: the green weight multiplies .b */
void ora_color_to_intensity(const uint8_t *rgb, float *out, int n) {
  for (int i = 0; i < n; i++) {
    float r = rgb[3 * i] / 255.0f, b = rgb[3 * i + 2] / 255.0f;
    out[i] = (r * 0.299f + b * 0.587f) + b * 0.114f;
  }
}

/* image_kernels.cu:206-219 ; Q19: INF*0 -> NaN for invalid pixels */
void ora_transform_vertex_map(float *v, const float m[16], int n) {
  for (int i = 0; i < n; i++) {
    float in[4] = {v[3 * i], v[3 * i + 1], v[3 * i + 2], 1.0f}, o[4];
    mat4_mul_vec4(m, in, o);
    v[3 * i] = o[0]; v[3 * i + 1] = o[1]; v[3 * i + 2] = o[2];
  }
}
/* image_kernels.cu:221-234 */
void ora_transform_normal_map(float *v, const float m[16], int n) {
  for (int i = 0; i < n; i++) {
    float in[4] = {v[3 * i], v[3 * i + 1], v[3 * i + 2], 0.0f}, o[4];
    mat4_mul_vec4(m, in, o);
    v[3 * i] = o[0]; v[3 * i + 1] = o[1]; v[3 * i + 2] = o[2];
  }
}

/* image_kernels.cu:60-102.  thrust::reduce with min_vec3/max_vec3: points whose
 * x or z is not finite are skipped (Q1); a zero vector on the left means
 * "unset".  The reduction order of thrust::reduce is unspecified; min/max are
 * exact, so the oracle takes the plain min/max over the valid points and then
 * combines it with the caller's bbox under the "zero = unset" rule. */
void ora_point_cloud_bbox(const float *pts, int n, float bbox0[3], float bbox1[3]) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  int any = 0;
  for (int i = 0; i < n; i++) {
    const float *p = pts + 3 * (size_t)i;
    if (!finitef_(p[0]) || !finitef_(p[2])) continue;
    for (int k = 0; k < 3; k++) { lo[k] = fminf(p[k], lo[k]); hi[k] = fmaxf(p[k], hi[k]); }
    any = 1;
  }
  if (!any) return;
  int unset0 = bbox0[0] == 0.0f && bbox0[1] == 0.0f && bbox0[2] == 0.0f;
  int unset1 = bbox1[0] == 0.0f && bbox1[1] == 0.0f && bbox1[2] == 0.0f;
  for (int k = 0; k < 3; k++) {
    bbox0[k] = unset0 ? lo[k] : fminf(lo[k], bbox0[k]);
    bbox1[k] = unset1 ? hi[k] : fmaxf(hi[k], bbox1[k]);
  }
}

/* ======================================================================== */
/* ICP (src/sensor/localization_kernels.cu:154-229, 303-326)                 */
/* ======================================================================== */

#define ICP_DIST_THRESH 0.1f
#define ICP_NORM_THRESH 0.87f
#define ICP_SCALE_A 1048576.0    /* 2^20 */
#define ICP_SCALE_B 1073741824.0 /* 2^30 */

/* Accumulation spec.  The reference sums binary32 products in an unspecified
 * order (per-thread serial partials, then thrust::reduce).  The oracle and the
 * product instead accumulate every per-pixel binary32 product EXACTLY in fixed
 * point: q = rint(product * 2^20) for the 21 upper-triangle A terms and
 * rint(product * 2^30) for the 6 b terms, summed as integers.  The sum is
 * associative, hence independent of thread, block and GPU partitioning, and
 * more accurate than the reference's binary32 sums.
 * Q15: with load_size = 20*w/640 the reference reduces floor(n/load_size)
 * partials, so pixels >= floor(n/load)*load are excluded. */
static void icp_cost2_raw_mode(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n,
                               int first_pixel, int num_pixels, int w, int h, int corrected, int64_t acc[27]);
void ora_icp_cost2_raw(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n,
                       int first_pixel, int num_pixels, int w, int h, int64_t acc[27]) {
  icp_cost2_raw_mode(last_v, last_n, cur_v, cur_n, first_pixel, num_pixels, w, h, 0, acc);
}
/* corrected != 0: the CORRECTED tracker of this build (ora_camera_set_strict_reference): the rotational rows of the Jacobian
 * are [v2]x = ((0,-z,y),(z,0,-x),(-y,x,0)), i.e. A_T[0..2] = v2 x n1 -- the linearisation of n1 . (v1 - (R v2 + t)) in a small
 * rotation vector -- instead of the reference's rows (Q14).  Everything else (gates, products in this order, exact sums) as above. */
void ora_icp_cost2_raw_corrected(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n,
                                 int first_pixel, int num_pixels, int w, int h, int64_t acc[27]) {
  icp_cost2_raw_mode(last_v, last_n, cur_v, cur_n, first_pixel, num_pixels, w, h, 1, acc);
}
static void icp_cost2_raw_mode(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n,
                               int first_pixel, int num_pixels, int w, int h, int corrected, int64_t acc[27]) {
  for (int i = 0; i < 27; i++) acc[i] = 0;
  int n = w * h;
  int load_size = 20 * w / 640;
  int limit = n;
  if (load_size > 0) limit = (n / load_size) * load_size;
  int end = first_pixel + num_pixels;
  if (end > limit) end = limit;
  for (int p = first_pixel; p < end; p++) {
    const float *v2 = cur_v + 3 * (size_t)p, *n2 = cur_n + 3 * (size_t)p;
    const float *v1 = last_v + 3 * (size_t)p, *n1 = last_n + 3 * (size_t)p;
    if (!finitef_(v2[0]) || !finitef_(v2[1]) || !finitef_(v2[2]) || !finitef_(v1[0]) || !finitef_(v1[1]) ||
        !finitef_(v1[2]) || (v1[2] < 0.1f) || (v2[2] < 0.1f) || (v1[2] > 10.0f) || (v2[2] > 10.0f))
      continue;
    if (!finitef_(n2[0]) || !finitef_(n2[1]) || !finitef_(n2[2]) || !finitef_(n1[0]) || !finitef_(n1[1]) ||
        !finitef_(n1[2]))
      continue;
    float d[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    if (length3(d) > ICP_DIST_THRESH) continue;
    if (dot3(n2, n1) < ICP_NORM_THRESH) continue;
    /* G_T rows (Q14); corrected: the rows of [v2]x */
    const float G_ref[18] = {0.0f, -v2[0], -v2[1], -v2[2], 0.0f, v2[0], v2[1], v2[2], 0.0f,
                             1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    const float G_cor[18] = {0.0f, -v2[2], v2[1], v2[2], 0.0f, -v2[0], -v2[1], v2[0], 0.0f,
                             1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    const float *G_T = corrected ? G_cor : G_ref;
    float A_T[6];
    for (int i = 0; i < 6; i++) A_T[i] = (G_T[3 * i] * n1[0] + G_T[3 * i + 1] * n1[1]) + G_T[3 * i + 2] * n1[2];
    float dv[3] = {v1[0] - v2[0], v1[1] - v2[1], v1[2] - v2[2]};
    float bb = dot3(n1, dv);
    int k = 0;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) {
        float prod = A_T[i] * A_T[j];
        acc[k++] += (int64_t)rint((double)prod * ICP_SCALE_A);
      }
    for (int i = 0; i < 6; i++) {
      float prod = bb * A_T[i];
      acc[21 + i] += (int64_t)rint((double)prod * ICP_SCALE_B);
    }
  }
}

void ora_icp_finish(const int64_t acc[27], float A[36], float b[6]) {
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      float v = (float)((double)acc[k++] * (1.0 / ICP_SCALE_A));
      A[6 * i + j] = v;
      A[6 * j + i] = v;
    }
  for (int i = 0; i < 6; i++) b[i] = (float)((double)acc[21 + i] * (1.0 / ICP_SCALE_B));
}

void ora_icp_cost2(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n, int w, int h,
                   float A[36], float b[6]) {
  int64_t acc[27];
  ora_icp_cost2_raw(last_v, last_n, cur_v, cur_n, 0, w * h, w, h, acc);
  ora_icp_finish(acc, A, b);
}

/* computeICPCost (localization_kernels.cu:59-152,231-301), the variant with an explicit correspondence
 * stencil: gates = all components finite, |v2 - v1| <= DIST_THRESH, n2.n1 >= NORM_THRESH (no depth-range
 * gate, unlike computeICPCost2); order-preserving compaction of the matches; load_size = 10 and a reduce
 * over floor(M/10) partials, so only the first floor(M/10)*10 matches contribute (the tail partial is
 * written past the end of d_A and never summed, :289-293); J uses v2 and the LAST frame's normal.
 * Returns M; with M <= 0 the reference returns before touching A and b (:251-253).
 * Sums in exact fixed point as ora_icp_cost2_raw (R3). */
int ora_icp_cost_raw(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n, int w, int h,
                     int64_t acc[27]) {
  for (int i = 0; i < 27; i++) acc[i] = 0;
  const int n = w * h;
  int m = 0;
  for (int p = 0; p < n; p++) {
    const float *v2 = cur_v + 3 * (size_t)p, *n2 = cur_n + 3 * (size_t)p;
    const float *v1 = last_v + 3 * (size_t)p, *n1 = last_n + 3 * (size_t)p;
    if (!finitef_(v2[0]) || !finitef_(v2[1]) || !finitef_(v2[2]) || !finitef_(v1[0]) || !finitef_(v1[1]) || !finitef_(v1[2])) continue;
    if (!finitef_(n2[0]) || !finitef_(n2[1]) || !finitef_(n2[2]) || !finitef_(n1[0]) || !finitef_(n1[1]) || !finitef_(n1[2])) continue;
    float d[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    if (length3(d) > ICP_DIST_THRESH) continue;
    if (dot3(n2, n1) < ICP_NORM_THRESH) continue;
    m++;
  }
  const int limit = (m / 10) * 10;
  int rank = 0;
  for (int p = 0; p < n && rank < limit; p++) {
    const float *v2 = cur_v + 3 * (size_t)p, *n2 = cur_n + 3 * (size_t)p;
    const float *v1 = last_v + 3 * (size_t)p, *n1 = last_n + 3 * (size_t)p;
    if (!finitef_(v2[0]) || !finitef_(v2[1]) || !finitef_(v2[2]) || !finitef_(v1[0]) || !finitef_(v1[1]) || !finitef_(v1[2])) continue;
    if (!finitef_(n2[0]) || !finitef_(n2[1]) || !finitef_(n2[2]) || !finitef_(n1[0]) || !finitef_(n1[1]) || !finitef_(n1[2])) continue;
    float d[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    if (length3(d) > ICP_DIST_THRESH) continue;
    if (dot3(n2, n1) < ICP_NORM_THRESH) continue;
    rank++;
    const float G_T[18] = {0.0f, -v2[0], -v2[1], -v2[2], 0.0f, v2[0], v2[1], v2[2], 0.0f,
                           1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    float A_T[6];
    for (int i = 0; i < 6; i++) A_T[i] = (G_T[3 * i] * n1[0] + G_T[3 * i + 1] * n1[1]) + G_T[3 * i + 2] * n1[2];
    float dv[3] = {v1[0] - v2[0], v1[1] - v2[1], v1[2] - v2[2]};
    float bb = dot3(n1, dv);
    int k = 0;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) {
        float prod = A_T[i] * A_T[j];
        acc[k++] += (int64_t)rint((double)prod * ICP_SCALE_A);
      }
    for (int i = 0; i < 6; i++) {
      float prod = bb * A_T[i];
      acc[21 + i] += (int64_t)rint((double)prod * ICP_SCALE_B);
    }
  }
  return m;
}

int ora_icp_cost(const float *last_v, const float *last_n, const float *cur_v, const float *cur_n, int w, int h,
                 float A[36], float b[6]) {
  int64_t acc[27];
  const int m = ora_icp_cost_raw(last_v, last_n, cur_v, cur_n, w, h, acc);
  if (m > 0) ora_icp_finish(acc, A, b);
  return m;
}

/* rgbd_camera.cpp:194-222 : float storage, double inner sums */
void ora_solve_cholesky(int dimension, const float *A, const float *b, float *x) {
  float LU[36] = {0}, y[6] = {0};
  for (int k = 0; k < dimension; ++k) {
    double sum = 0.;
    for (int p = 0; p < k; ++p) sum += LU[k * dimension + p] * LU[k * dimension + p];
    LU[k * dimension + k] = (float)sqrt(A[k * dimension + k] - sum);
    for (int i = k + 1; i < dimension; ++i) {
      double sum2 = 0.;
      for (int p = 0; p < k; ++p) sum2 += LU[i * dimension + p] * LU[k * dimension + p];
      LU[i * dimension + k] = (float)((A[i * dimension + k] - sum2) / LU[k * dimension + k]);
    }
  }
  for (int i = 0; i < dimension; ++i) {
    double sum = 0.;
    for (int k = 0; k < i; ++k) sum += LU[i * dimension + k] * y[k];
    y[i] = (float)((b[i] - sum) / LU[i * dimension + i]);
  }
  for (int i = dimension - 1; i >= 0; --i) {
    double sum = 0.;
    for (int k = i + 1; k < dimension; ++k) sum += LU[k * dimension + i] * x[k];
    x[i] = (float)((y[i] - sum) / LU[i * dimension + i]);
  }
}

/* rgbd_camera.cpp:154-158 */
void ora_icp_update_transform(const float x[6], float out[16]) {
  float I[16], rz[16], ry[16], rx[16], tr[16], t1[16], t2[16];
  static const float ax[3] = {1, 0, 0}, ay[3] = {0, 1, 0}, az[3] = {0, 0, 1};
  ora_mat4_identity(I);
  ora_mat4_rotate_deg(I, -x[2] * 180.0f / 3.14159f, az, rz);
  ora_mat4_rotate_deg(I, -x[1] * 180.0f / 3.14159f, ay, ry);
  ora_mat4_rotate_deg(I, -x[0] * 180.0f / 3.14159f, ax, rx);
  float tv[3] = {x[3], x[4], x[5]};
  ora_mat4_translate(I, tv, tr);
  ora_mat4_mul(rz, ry, t1);
  ora_mat4_mul(t1, rx, t2);
  ora_mat4_mul(t2, tr, out);
}

/* the CORRECTED tracker's update (own specification): x = (rotation vector, translation) of the linearisation above, so a
 * point of the current frame moves to R v + t with R = Rz(x2) Ry(x1) Rx(x0) (positive angles; the reference negates them and
 * translates BEFORE rotating, :154-158): translate(x3, x4, x5) * Rz * Ry * Rx with the same glm calls */
void ora_icp_update_transform_corrected(const float x[6], float out[16]) {
  float I[16], rz[16], ry[16], rx[16], tr[16], t1[16], t2[16];
  static const float ax[3] = {1, 0, 0}, ay[3] = {0, 1, 0}, az[3] = {0, 0, 1};
  ora_mat4_identity(I);
  ora_mat4_rotate_deg(I, x[2] * 180.0f / 3.14159f, az, rz);
  ora_mat4_rotate_deg(I, x[1] * 180.0f / 3.14159f, ay, ry);
  ora_mat4_rotate_deg(I, x[0] * 180.0f / 3.14159f, ax, rx);
  float tv[3] = {x[3], x[4], x[5]};
  ora_mat4_translate(I, tv, tr);
  ora_mat4_mul(tr, rz, t1);
  ora_mat4_mul(t1, ry, t2);
  ora_mat4_mul(t2, rx, out);
}

/* ======================================================================== */
/* tracker (src/sensor/rgbd_camera.cpp:22-191)                               */
/* ======================================================================== */

#define PYR 3
static const int PYRAMID_ITERS[PYR] = {10, 5, 4};

/* ---- photometric RGB-D term -------------------------------------------------------------------------
 * The reference DECLARES gradient / difference (image_kernels.h:45-49) and computeRGBDCost (localization_kernels.h:42)
 * but ships no definition of the first two and an empty body for the third (localization_kernels.cu:328-331); the call
 * site is commented out (rgbd_camera.cpp:126-141), W_RGBD = 0.1 (:20).  There is NO reference behaviour to match: what
 * follows is this build's own specification, restated here as the checker of the HIP kernels.
 *   gradient    Sobel 3x3 / 8 on interior pixels, (0,0) on the border
 *   difference  out = in1 - in2
 *   rgbd cost   same-index association like computeICPCost2 (no reprojection); residual r = I_last - I_cur; Jacobian
 *               J = G_T * (gx * du/dv + gy * dv/dv) with the last frame's gradient, the pinhole derivative at the
 *               current (already transformed) vertex and the SAME G_T rows as the geometric term (:208-213, Q14), so
 *               that both systems share one parametrisation; exact fixed-point sums (2^8 for A, 2^20 for b). */
#define ORA_W_RGBD 0.1f
#define RGBD_SCALE_A 256.0
#define RGBD_SCALE_B 1048576.0

void ora_gradient(const float *in, float *grad, int w, int h) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      float *g = grad + 2 * ((size_t)y * w + x);
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) { g[0] = 0.0f; g[1] = 0.0f; continue; }
      const float *r0 = in + (size_t)(y - 1) * w + x, *r1 = in + (size_t)y * w + x, *r2 = in + (size_t)(y + 1) * w + x;
      float gx = ((r0[1] - r0[-1]) + 2.0f * (r1[1] - r1[-1])) + (r2[1] - r2[-1]);
      float gy = ((r2[-1] - r0[-1]) + 2.0f * (r2[0] - r0[0])) + (r2[1] - r0[1]);
      g[0] = gx * 0.125f; g[1] = gy * 0.125f;
    }
}

void ora_difference(const float *in1, const float *in2, float *out, int n) {
  for (int i = 0; i < n; i++) out[i] = in1[i] - in2[i];
}

void ora_rgbd_cost_raw(const float *last_i, const float *last_g, const float *last_v, const float *cur_i, const float *cur_v,
                       int w, int h, float fx, float fy, int img_w, int img_h, int64_t acc[27]) {
  for (int i = 0; i < 27; i++) acc[i] = 0;
  const float sx = (float)(img_w / w), sy = (float)(img_h / h); /* level pixels per full-resolution pixel, as generateVertexMap */
  for (int p = 0; p < w * h; p++) {
    const float *v2 = cur_v + 3 * (size_t)p, *v1 = last_v + 3 * (size_t)p;
    if (!finitef_(v2[0]) || !finitef_(v2[1]) || !finitef_(v2[2]) || !finitef_(v1[0]) || !finitef_(v1[1]) ||
        !finitef_(v1[2]) || (v1[2] < 0.1f) || (v2[2] < 0.1f) || (v1[2] > 10.0f) || (v2[2] > 10.0f))
      continue;
    float d[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    if (length3(d) > ICP_DIST_THRESH) continue;
    const float gx = last_g[2 * (size_t)p], gy = last_g[2 * (size_t)p + 1];
    const float r = last_i[p] - cur_i[p];
    const float iz = 1.0f / v2[2];
    const float ax = (fx * iz) / sx, ay = (fy * iz) / sy; /* d u / d X,  -(d v / d Y) */
    float wv[3];
    wv[0] = gx * ax;
    wv[1] = -(gy * ay);
    wv[2] = (gy * ay) * (v2[1] * iz) - (gx * ax) * (v2[0] * iz);
    const float G_T[18] = {0.0f, -v2[0], -v2[1], -v2[2], 0.0f, v2[0], v2[1], v2[2], 0.0f,
                           1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    float J[6];
    for (int i = 0; i < 6; i++) J[i] = (G_T[3 * i] * wv[0] + G_T[3 * i + 1] * wv[1]) + G_T[3 * i + 2] * wv[2];
    int k = 0;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) {
        float prod = J[i] * J[j];
        acc[k++] += (int64_t)rint((double)prod * RGBD_SCALE_A);
      }
    for (int i = 0; i < 6; i++) {
      float prod = r * J[i];
      acc[21 + i] += (int64_t)rint((double)prod * RGBD_SCALE_B);
    }
  }
}

void ora_rgbd_finish(const int64_t acc[27], float A[36], float b[6]) {
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      float v = (float)((double)acc[k++] * (1.0 / RGBD_SCALE_A));
      A[6 * i + j] = v;
      A[6 * j + i] = v;
    }
  for (int i = 0; i < 6; i++) b[i] = (float)((double)acc[21 + i] * (1.0 / RGBD_SCALE_B));
}

void ora_rgbd_cost(const float *last_i, const float *last_g, const float *last_v, const float *cur_i, const float *cur_v, int w, int h,
                   float fx, float fy, int img_w, int img_h, float A[36], float b[6]) {
  int64_t acc[27];
  ora_rgbd_cost_raw(last_i, last_g, last_v, cur_i, cur_v, w, h, fx, fy, img_w, img_h, acc);
  ora_rgbd_finish(acc, A, b);
}

struct ora_camera {
  int width, height;
  float fx, fy;
  int pass;
  long long latest_stamp;
  float position[3];
  float orientation[9]; /* column-major mat3 */
  float *last_v[PYR], *last_n[PYR], *cur_v[PYR], *cur_n[PYR];
  float lastA[36], lastb[6], lastx[6];
  int lost_count; /* levels abandoned with "Camera tracking is lost." (rgbd_camera.cpp:148-151) */
  float last_update[16]; /* update_trans of the frame most recently tracked (identity for a first frame) */
  /* photometric RGB-D term (off by default = the reference, which ships it commented out) */
  int rgbd;
  float *last_i[PYR], *cur_i[PYR], *last_g[PYR], *cur_g[PYR]; /* intensity and its Sobel gradient per level */
  /* frame-to-model tracking (off by default): maps the ICP tracks against instead of the previous frame's */
  int to_model, have_model;
  float *model_v[PYR], *model_n[PYR];
  /* ora_camera_set_strict_reference(c, 0): the corrected tracker (own specification; default 1 = the reference's, Q14 / Q17) */
  int corrected;
};

ora_camera *ora_camera_create(int w, int h, float fx, float fy) {
  ora_camera *c = (ora_camera *)calloc(1, sizeof(ora_camera));
  c->width = w; c->height = h; c->fx = fx; c->fy = fy;
  c->latest_stamp = -1; /* the reference leaves latest_stamp_ uninitialised; frames are stamped from 0 */
  c->orientation[0] = c->orientation[4] = c->orientation[8] = 1.0f;
  for (int i = 0; i < PYR; i++) {
    size_t n = (size_t)(w >> i) * (size_t)(h >> i) * 3;
    c->last_v[i] = (float *)malloc(sizeof(float) * n);
    c->last_n[i] = (float *)malloc(sizeof(float) * n);
    c->cur_v[i] = (float *)malloc(sizeof(float) * n);
    c->cur_n[i] = (float *)malloc(sizeof(float) * n);
  }
  return c;
}

void ora_camera_destroy(ora_camera *c) {
  if (!c) return;
  for (int i = 0; i < PYR; i++) { free(c->last_v[i]); free(c->last_n[i]); free(c->cur_v[i]); free(c->cur_n[i]); }
  for (int i = 0; i < PYR; i++) { free(c->last_i[i]); free(c->cur_i[i]); free(c->last_g[i]); free(c->cur_g[i]); }
  for (int i = 0; i < PYR; i++) { free(c->model_v[i]); free(c->model_n[i]); }
  free(c);
}

static void mat3_to_mat4(const float m3[9], float m4[16]) {
  ora_mat4_identity(m4);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) m4[4 * c + r] = m3[3 * c + r];
}

/* rgbd_camera.cpp:172-173 (Q17: row-vector products).  The row-vector product gives R^T p and drops the update's
 * translation; main.cpp:40 maps a camera point x to orientation * (x + position), so the pose that composes camera -> map with
 * the frame-to-frame transform v_last = R v_cur + t is orientation * R and R^T (position + t): the corrected tracker adds t
 * to the position first (same product, same order otherwise). */
static void pose_step(ora_camera *c, const float update_trans[16]) {
  float p4[4] = {c->position[0], c->position[1], c->position[2], 1.0f}, np[4];
  if (c->corrected) { p4[0] = p4[0] + update_trans[12]; p4[1] = p4[1] + update_trans[13]; p4[2] = p4[2] + update_trans[14]; }
  vec4_mul_mat4(p4, update_trans, np);
  if (c->corrected) { /* (the translation column contributes to w only; the rotation part of the product is R^T (p + t)) */ }
  c->position[0] = np[0]; c->position[1] = np[1]; c->position[2] = np[2];
  float o4[16], no[16];
  mat3_to_mat4(c->orientation, o4);
  ora_mat4_mul(o4, update_trans, no);
  for (int cc = 0; cc < 3; cc++)
    for (int r = 0; r < 3; r++) c->orientation[3 * cc + r] = no[4 * cc + r];
}

/* Frame-parallel tracking.  update_trans starts at the identity for every frame (:100) and the ICP
 * loop reads the maps of the last and the current frame only (:103-168): a frame's update_trans is a function of two
 * depth images, and only :172-173 chain the frames.  last_update() exposes a tracked frame's update_trans;
 * apply_delta() is :55-59 + :172-173 for an update_trans from anywhere (NULL, or a camera's first frame -- `pass >= 1`,
 * :99 -- leaves the pose alone) plus the levels it abandoned. */
void ora_camera_last_update(const ora_camera *c, float out[16]) { memcpy(out, c->last_update, sizeof(c->last_update)); }

int ora_camera_apply_delta(ora_camera *c, const float *update_trans, int levels_lost, long long timestamp) {
  if (timestamp <= c->latest_stamp) return 0;
  c->latest_stamp = timestamp;
  if (c->pass >= 1 && update_trans) {
    pose_step(c, update_trans);
    c->lost_count += levels_lost;
  }
  if (c->pass < 2) c->pass++;
  return 1;
}

int ora_camera_update(ora_camera *c, const uint16_t *depth, const uint8_t *rgb, long long timestamp) {
  if (timestamp <= c->latest_stamp) return 0;
  c->latest_stamp = timestamp;
  const int W = c->width, H = c->height;
  if (c->rgbd) { /* rgbd_camera.cpp:66-69,85,90: intensity pyramid by plain 2x2 subsampling; + Sobel gradient per level */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
    ora_color_to_intensity(rgb, tmp, W * H);
    for (int i = 0; i < PYR; i++) {
      int w = W >> i, h = H >> i;
      memcpy(c->cur_i[i], tmp, sizeof(float) * (size_t)w * h);
      ora_gradient(c->cur_i[i], c->cur_g[i], w, h);
      if (i != PYR - 1) ora_subsample_f32(tmp, w, h);
    }
    free(tmp);
  }
  uint16_t *filtered = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)W * H);
  ora_bilateral_filter(depth, filtered, W, H);
  for (int i = 0; i < PYR; i++) { /* rgbd_camera.cpp:72-93 */
    int w = W >> i, h = H >> i;
    ora_generate_vertex_map(filtered, c->cur_v[i], w, h, c->fx, c->fy, W, H);
    ora_generate_normal_map(c->cur_v[i], c->cur_n[i], w, h);
    if (i != PYR - 1) ora_subsample_depth_u16(filtered, w, h);
  }
  free(filtered);
  if (c->pass >= 1) { /* :99-174 */
    float update_trans[16];
    ora_mat4_identity(update_trans);
    for (int i = PYR - 1; i >= 0; i--) {
      int w = W >> i, h = H >> i, n = w * h;
      float *fv = (float *)malloc(sizeof(float) * 3 * (size_t)n), *fn = (float *)malloc(sizeof(float) * 3 * (size_t)n);
      memcpy(fv, c->cur_v[i], sizeof(float) * 3 * (size_t)n);
      memcpy(fn, c->cur_n[i], sizeof(float) * 3 * (size_t)n);
      if (i < PYR - 1) {
        ora_transform_vertex_map(fv, update_trans, n);
        ora_transform_normal_map(fn, update_trans, n);
      }
      for (int j = 0; j < PYRAMID_ITERS[i]; j++) {
        float A1[36], b1[6], x[6];
        const int model = c->to_model && c->have_model; /* the TODO of rgbd_camera.cpp:185: the maps tracked against come from elsewhere */
        if (c->corrected) {
          int64_t acc[27];
          ora_icp_cost2_raw_corrected(model ? c->model_v[i] : c->last_v[i], model ? c->model_n[i] : c->last_n[i], fv, fn, 0, w * h, w, h, acc);
          ora_icp_finish(acc, A1, b1);
        } else
        ora_icp_cost2(model ? c->model_v[i] : c->last_v[i], model ? c->model_n[i] : c->last_n[i], fv, fn, w, h, A1, b1);
        if (c->rgbd) { /* rgbd_camera.cpp:126-141 with W_RGBD (:20) applied to the photometric system */
          float A2[36], b2[6];
          ora_rgbd_cost(c->last_i[i], c->last_g[i], c->last_v[i], c->cur_i[i], fv, w, h, c->fx, c->fy, W, H, A2, b2);
          for (int k = 0; k < 36; k++) A1[k] = A1[k] + ORA_W_RGBD * A2[k];
          for (int k = 0; k < 6; k++) b1[k] = b1[k] + ORA_W_RGBD * b2[k];
        }
        ora_solve_cholesky(6, A1, b1, x);
        memcpy(c->lastA, A1, sizeof(A1)); memcpy(c->lastb, b1, sizeof(b1)); memcpy(c->lastx, x, sizeof(x));
        if (x[0] != x[0] || x[1] != x[1] || x[2] != x[2] || x[3] != x[3] || x[4] != x[4] || x[5] != x[5]) { c->lost_count++; break; }
        float this_trans[16];
        if (c->corrected) ora_icp_update_transform_corrected(x, this_trans);
        else ora_icp_update_transform(x, this_trans);
        ora_mat4_mul(this_trans, update_trans, update_trans);
        if (j < PYRAMID_ITERS[i] - 1) {
          ora_transform_vertex_map(fv, this_trans, n);
          ora_transform_normal_map(fn, this_trans, n);
        }
      }
      free(fv); free(fn);
    }
    memcpy(c->last_update, update_trans, sizeof(update_trans));
    pose_step(c, update_trans);
  } else {
    ora_mat4_identity(c->last_update);
  }
  if (c->pass < 2) c->pass++;
  for (int i = 0; i < PYR; i++) { /* :181-189 */
    float *t;
    t = c->cur_v[i]; c->cur_v[i] = c->last_v[i]; c->last_v[i] = t;
    t = c->cur_n[i]; c->cur_n[i] = c->last_n[i]; c->last_n[i] = t;
    t = c->cur_i[i]; c->cur_i[i] = c->last_i[i]; c->last_i[i] = t;
    t = c->cur_g[i]; c->cur_g[i] = c->last_g[i]; c->last_g[i] = t;
  }
  return 1;
}

void ora_camera_set_rgbd(ora_camera *c, int enable) {
  if (enable && c->to_model) return; /* (not combined with frame-to-model tracking) */
  c->rgbd = enable != 0;
  for (int i = 0; i < PYR && enable; i++) {
    size_t n = (size_t)(c->width >> i) * (size_t)(c->height >> i);
    if (!c->last_i[i]) {
      c->last_i[i] = (float *)calloc(n, sizeof(float)); c->cur_i[i] = (float *)calloc(n, sizeof(float));
      c->last_g[i] = (float *)calloc(2 * n, sizeof(float)); c->cur_g[i] = (float *)calloc(2 * n, sizeof(float));
    }
  }
}

/* Frame-to-model tracking (own specification, see ora_raycast_model_depth).  set_model_depth: `depth` -- a depth image
 * in the sensor's pixel grid and unit, e.g. the map ray-cast from the pose of the frame just tracked -- goes through the
 * front end of a sensor frame (bilateral filter, three pyramid levels, vertex and normal maps: rgbd_camera.cpp:62-93)
 * into a map set of its own.  With set_frame_to_model(1) every ICP iteration associates the incoming frame with THAT set
 * instead of the previous frame's maps, until the next set_model_depth replaces it; frames tracked before any model was
 * given, and the pose composition, are the reference's.  Not combined with the photometric term (the model has no
 * intensity image): returns -1. */
int ora_camera_set_model_depth(ora_camera *c, const uint16_t *depth) {
  if (!depth) { c->have_model = 0; return 0; } /* no model: the following frames are tracked against the previous frame again */
  const int W = c->width, H = c->height;
  for (int i = 0; i < PYR; i++)
    if (!c->model_v[i]) {
      size_t n = (size_t)(W >> i) * (size_t)(H >> i) * 3;
      c->model_v[i] = (float *)malloc(sizeof(float) * n);
      c->model_n[i] = (float *)malloc(sizeof(float) * n);
    }
  uint16_t *filtered = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)W * H);
  ora_bilateral_filter(depth, filtered, W, H);
  for (int i = 0; i < PYR; i++) {
    int w = W >> i, h = H >> i;
    ora_generate_vertex_map(filtered, c->model_v[i], w, h, c->fx, c->fy, W, H);
    ora_generate_normal_map(c->model_v[i], c->model_n[i], w, h);
    if (i != PYR - 1) ora_subsample_depth_u16(filtered, w, h);
  }
  free(filtered);
  c->have_model = 1;
  return 0;
}

int ora_camera_set_frame_to_model(ora_camera *c, int enable) {
  if (enable && c->rgbd) return -1;
  c->to_model = enable != 0;
  return 0;
}

int ora_camera_tracking_lost_count(const ora_camera *c) { return c->lost_count; }

/* strict = 1 (default): RGBDCamera::update as the reference has it, every quirk included.  strict = 0: this build's CORRECTED
 * tracker -- the three places where the reference's update is not a rigid-motion estimate are replaced (Jacobian rows = [v2]x
 * instead of Q14's; this_trans = T(t) Rz Ry Rx with positive angles; the position takes the update's translation, Q17) --
 * own specification, no reference behaviour; before the first frame. */
void ora_camera_set_strict_reference(ora_camera *c, int strict) { c->corrected = !strict; }

void ora_camera_pose(const ora_camera *c, float position[3], float orientation[9]) {
  memcpy(position, c->position, sizeof(float) * 3);
  memcpy(orientation, c->orientation, sizeof(float) * 9);
}

/* overwrite position_ / orientation_ (a session continued from a pose tracked elsewhere; not a reference function) */
void ora_camera_set_pose(ora_camera *c, const float position[3], const float orientation[9]) {
  memcpy(c->position, position, sizeof(float) * 3);
  memcpy(c->orientation, orientation, sizeof(float) * 9);
}

/* main.cpp:40 : glm::mat4(orientation) * glm::translate(glm::mat4(1.0f), position) */
void ora_camera_fusion_transform(const ora_camera *c, float out[16]) {
  float o4[16], I[16], t[16];
  mat3_to_mat4(c->orientation, o4);
  ora_mat4_identity(I);
  ora_mat4_translate(I, c->position, t);
  ora_mat4_mul(o4, t, out);
}

void ora_camera_last_system(const ora_camera *c, float A[36], float b[6], float x[6]) {
  memcpy(A, c->lastA, sizeof(c->lastA));
  memcpy(b, c->lastb, sizeof(c->lastb));
  memcpy(x, c->lastx, sizeof(c->lastx));
}
