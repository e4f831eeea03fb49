/*
 * svoslam_oracle_mesh.c -- CPU restatement of the mesh -> voxel grid path of the reference:
 * OBJ loading (external/src/objUtil/objloader.cpp:14-122, obj.cpp:33-135,227-238), BMP loading
 * (src/world/scene.cpp:35-62) and meshToVoxelGrid (src/world/voxelization/voxelization.cu:50-139,
 * 219-236,381-405) with the VoxelPipe THIN_RASTER / NO_BLENDING / FP32S rule it instantiates
 * (external/include/voxelpipe/coarse.h:59-102, utils.h:185-254, fine.h:130-152,239-365,936-959,
 * tile.h:45-51).  TEST INFRASTRUCTURE ONLY.  The OBJ loader is pinned against the reference's own objUtil
 * build (oracle/_ref, tests/test_ref_obj_loader.py); the voxelizer's parity is unpinned by the reference
 * (see svoslam_oracle.h).
 *
 * Deterministic resolution: NO_BLENDING is a plain store (last writer wins, a race between
 * triangles that share a voxel); the oracle lets the HIGHEST triangle id win.
 * Texture fetches outside the image (u or v == 1.0) read out of bounds in the reference; the
 * oracle clamps the linear texel index into the image.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svoslam_oracle.h"

/* ---------------------------------------------------------------- OBJ */
typedef struct { float *v; int n, cap; } fvec;
typedef struct { int *v; int n, cap; } ivec;
static void fpush(fvec *a, float x) { if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 256; a->v = (float *)realloc(a->v, sizeof(float) * (size_t)a->cap); } a->v[a->n++] = x; }
static void ipush(ivec *a, int x) { if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 256; a->v = (int *)realloc(a->v, sizeof(int) * (size_t)a->cap); } a->v[a->n++] = x; }

/* getline(ss, tok, ' ') semantics: next token up to the next single space (empty tokens possible) */
static int next_tok(const char **p, char *out, size_t cap) {
  if (**p == 0) return 0;
  size_t k = 0;
  while (**p && **p != ' ') { if (k + 1 < cap) out[k++] = **p; (*p)++; }
  out[k] = 0;
  if (**p == ' ') (*p)++;
  return 1;
}

static float sub3n[3];
static void face_normal(const float *p, const int *f, int i0, int i1, int i2, int i3, float out[3]) {
  /* normalize(cross(p[i0]-p[i1], p[i2]-p[i3])) as in obj::isConvex */
  float a[3], b[3];
  for (int k = 0; k < 3; k++) { a[k] = p[4 * f[i0] + k] - p[4 * f[i1] + k]; b[k] = p[4 * f[i2] + k] - p[4 * f[i3] + k]; }
  float c[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
  float inv = 1.0f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  out[0] = c[0] * inv; out[1] = c[1] * inv; out[2] = c[2] * inv;
  (void)sub3n;
}

/* obj::isConvex, obj.cpp:137-164 (EPSILON = DBL_EPSILON; abs() on the float difference) */
static int is_convex(const float *pts, const int *f, int n) {
  if (n <= 3) return 1;
  const double EPS = 2.220446049250313e-16;
  int k = n - 1;
  float nn[3], m[3];
  face_normal(pts, f, 0, k, 0, 1, nn);
  for (int i = 2; i < n; i++) {
    face_normal(pts, f, i - 1, i - 2, i - 1, i, m);
    if (fabs(m[0] - nn[0]) > EPS || fabs(m[1] - nn[1]) > EPS || fabs(m[2] - nn[2]) > EPS) return 0;
  }
  face_normal(pts, f, k, k - 1, k, 0, m);
  if (fabs(m[0] - nn[0]) > EPS || fabs(m[1] - nn[1]) > EPS || fabs(m[2] - nn[2]) > EPS) return 0;
  return 1;
}

/* Returns the number of triangles; *vbo = 9 floats per triangle (non-indexed), *tbo = 6 floats per
 * triangle or NULL when the OBJ has no texture coordinates; bbox0/bbox1 as Scene::objToMesh builds
 * them (scene.cpp:129-130).  -1 if the file cannot be opened. */
int ora_mesh_load_obj(const char *path, float **vbo, float **tbo, int *tbosize, float bbox0[3], float bbox1[3]) {
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  fvec pts = {0}, tcs = {0};      /* points: 4 floats each (vec4), texcoords: 4 floats each */
  ivec fidx = {0}, fstart = {0}, tidx = {0}, tstart = {0};
  int have_tex_faces = 0;
  char *line = NULL; size_t cap = 0; ssize_t len;
  int maxmin_set = 0;
  float xmax = 0, xmin = 0, ymax = 0, ymin = 0, zmax = 0, zmin = 0;
  while ((len = getline(&line, &cap, fp)) >= 0) {
    while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;  /* getline() of the reference strips '\n' only; '\r' would end up in the last token and atof ignores it */
    if (len == 0) continue;
    const char *p = line;
    char tok[256];
    if (line[0] == 'v' && line[1] == 't') {
      float c[3] = {0, 0, 0};
      next_tok(&p, tok, sizeof tok);
      for (int k = 0; k < 3; k++) { tok[0] = 0; next_tok(&p, tok, sizeof tok); c[k] = (float)atof(tok); }
      fpush(&tcs, c[0]); fpush(&tcs, c[1]); fpush(&tcs, c[2]); fpush(&tcs, 1.0f);
    } else if (line[0] == 'v' && line[1] == 'n') {
      /* normals are not used by the voxelizer */
    } else if (line[0] == 'v') {
      float c[3] = {0, 0, 0};
      next_tok(&p, tok, sizeof tok);
      for (int k = 0; k < 3; k++) { tok[0] = 0; next_tok(&p, tok, sizeof tok); c[k] = (float)atof(tok); }
      fpush(&pts, c[0]); fpush(&pts, c[1]); fpush(&pts, c[2]); fpush(&pts, 1.0f);
      /* obj::addPoint -> compareMaxMin (obj.cpp:112-135) */
      if (maxmin_set) {
        if (c[0] > xmax) xmax = c[0];
        if (c[0] < xmin) xmin = c[0];
        if (c[1] > ymax) ymax = c[1];
        if (c[1] < ymin) ymin = c[1];
        if (c[2] > zmax) zmax = c[2];
        if (c[2] < zmin) zmin = c[2];
      } else { xmax = xmin = c[0]; ymax = ymin = c[1]; zmax = zmin = c[2]; maxmin_set = 1; }
    } else if (line[0] == 'f') {
      next_tok(&p, tok, sizeof tok);
      ipush(&fstart, fidx.n);
      int has_slash = strchr(line, '/') != NULL, has_dslash = strstr(line, "//") != NULL;
      if (has_slash && !has_dslash) { ipush(&tstart, tidx.n); have_tex_faces = 1; }
      while (next_tok(&p, tok, sizeof tok)) {
        /* pointList.push_back(atof(f) - 1) with f = text before the first '/' */
        char *s1 = strchr(tok, '/');
        if (s1) *s1 = 0;
        ipush(&fidx, (int)(atof(tok) - 1));
        if (has_slash && !has_dslash) {
          const char *t = s1 ? s1 + 1 : "";
          char tb[64]; size_t k = 0;
          while (*t && *t != '/' && k + 1 < sizeof tb) tb[k++] = *t++;
          tb[k] = 0;
          if (s1) ipush(&tidx, (int)(atof(tb) - 1));
        }
      }
    }
  }
  free(line);
  fclose(fp);
  ipush(&fstart, fidx.n);
  if (have_tex_faces) ipush(&tstart, tidx.n);
  const int npts = pts.n / 4, nfaces = fstart.n - 1;
  /* obj::recenter, obj.cpp:227-238 */
  if (npts > 0) {
    float center[3] = {(xmax + xmin) / 2, ymin, (zmax + zmin) / 2};
    xmax = xmin = pts.v[0] - center[0];
    ymax = ymin = pts.v[1] - center[1];
    zmax = zmin = pts.v[2] - center[2];
    for (int i = 0; i < npts; i++) {
      float *q = pts.v + 4 * i;
      q[0] = q[0] - center[0]; q[1] = q[1] - center[1]; q[2] = q[2] - center[2];
      if (q[0] > xmax) xmax = q[0];
      if (q[0] < xmin) xmin = q[0];
      if (q[1] > ymax) ymax = q[1];
      if (q[1] < ymin) ymin = q[1];
      if (q[2] > zmax) zmax = q[2];
      if (q[2] < zmin) zmin = q[2];
    }
  }
  /* obj::buildVBOs, obj.cpp:33-110: fan triangulation of convex faces, non-indexed */
  fvec V = {0}, T = {0};
  const int has_texture = have_tex_faces && tstart.n - 1 > 0;
  for (int k = 0; k < nfaces; k++) {
    const int *f = fidx.v + fstart.v[k];
    const int n = fstart.v[k + 1] - fstart.v[k];
    if (!is_convex(pts.v, f, n)) continue;
    for (int i = 2; i < n; i++) {
      const int tri[3] = {f[0], f[i - 1], f[i]};
      for (int c = 0; c < 3; c++)
        for (int d = 0; d < 3; d++) fpush(&V, pts.v[4 * tri[c] + d]);
      if (has_texture) {
        /* the reference pushes facetexture[0], [1], [2] for every fan triangle (obj.cpp:75-80) */
        const int *ft = tidx.v + tstart.v[k];
        for (int c = 0; c < 3; c++) { fpush(&T, tcs.v[4 * ft[c]]); fpush(&T, tcs.v[4 * ft[c] + 1]); }
      }
    }
  }
  *vbo = V.v;
  *tbo = T.v;
  *tbosize = T.n;
  bbox0[0] = xmin; bbox0[1] = ymin; bbox0[2] = zmin;
  bbox1[0] = xmax; bbox1[1] = ymax; bbox1[2] = zmax;
  free(pts.v); free(tcs.v); free(fidx.v); free(fstart.v); free(tidx.v); free(tstart.v);
  return V.n / 9;
}

/* Scene::loadBMP, scene.cpp:35-62: 54-byte header, 24-bit BGR, no row padding handling.
 * *data = width*height*3 floats (r,g,b in 0..1).  Returns 0 on success. */
int ora_load_bmp(const char *path, float **data, int *width, int *height) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  unsigned char info[54];
  if (fread(info, 1, 54, f) != 54) { fclose(f); return -1; }
  int w, h;
  memcpy(&w, info + 18, 4);
  memcpy(&h, info + 22, 4);
  const int size = 3 * w * h;
  unsigned char *raw = (unsigned char *)calloc((size_t)size, 1);
  size_t got = fread(raw, 1, (size_t)size, f);
  (void)got;
  fclose(f);
  float *out = (float *)malloc(sizeof(float) * (size_t)size);
  for (int i = 0; i < size; i += 3) {
    out[i] = (int)raw[i + 2] / 255.0f;
    out[i + 1] = (int)raw[i + 1] / 255.0f;
    out[i + 2] = (int)raw[i] / 255.0f;
  }
  free(raw);
  *data = out; *width = w; *height = h;
  return 0;
}

/* ---------------------------------------------------------------- voxelizer */
static float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ColorShader::shade, voxelization.cu:90-138 */
static uint32_t shade(int tri_id, const float *tex, int tw, int th, const float *texcoord, int texcoord_size) {
  if (tw == 0) return (255u << 8) + (127u << 24);
  if (texcoord_size == 0) {
    int r = (int)(tex[0] * 255.0), g = (int)(tex[1] * 255.0), b = (int)(tex[2] * 255.0);
    return (uint32_t)(r + (g << 8) + (b << 16)) + (127u << 24);
  }
  int t1_x = (int)(texcoord[6 * tri_id] * tw);
  int t1_y = (int)(texcoord[6 * tri_id + 1] * th);
  long idx = (long)t1_y * tw + t1_x;
  if (idx < 0) idx = 0;
  if (idx > (long)tw * th - 1) idx = (long)tw * th - 1;
  const float *c = tex + 3 * idx;
  int r = (int)(clampf(c[0], 0.0f, 1.0f) * 255.0f);
  int g = (int)(clampf(c[1], 0.0f, 1.0f) * 255.0f);
  int b = (int)(clampf(c[2], 0.0f, 1.0f) * 255.0f);
  return (uint32_t)(r + (g << 8) + (b << 16)) + (127u << 24);
}

typedef struct { int64_t idx; uint32_t val; int tri; } frag_t;
static int cmp_frag(const void *a, const void *b) {
  const frag_t *x = (const frag_t *)a, *y = (const frag_t *)b;
  if (x->idx != y->idx) return (x->idx > y->idx) - (x->idx < y->idx);
  return (x->tri > y->tri) - (x->tri < y->tri);
}

static inline float sel3(const float *a, int k) { return a[k]; }

/* compute_scanline_bounds, fine.h:130-152 */
static void scanline_bounds(const float b[3], const float ndu[3], const float inv[3], int *min_u, int *max_u) {
  for (int k = 0; k < 3; k++) {
    if (ndu[k] > 0.0f) { int c = (int)ceilf(-b[k] * inv[k]); if (c > *min_u) *min_u = c; }
    else if (ndu[k] < 0.0f) { int c = (int)(-b[k] * inv[k]); if (c < *max_u) *max_u = c; }
    else if (b[k] < 0.0f) *min_u = *max_u + 1;
  }
}

/* meshToVoxelGrid with N = 2^log_N cells per axis over the mesh's own AABB and tiles of 2^log_T.
 * Output: *centers / *colors = count x 4 floats in ascending tiled-index order
 * (tile*T^3 + pix, voxelization.cu:141-164,312); colors[].a is left 0 (the reference leaves it
 * unwritten, Q21); *indices (optional) = the tiled indices.  Returns count. */
int ora_mesh_to_voxel_grid(const float *vbo, int n_tris, const float *tbo, int tbosize, const float *tex, int tex_w,
                           int tex_h, const float bbox0[3], const float bbox1[3], int log_N, int log_T,
                           float **centers, float **colors, int64_t **indices) {
  const int N = 1 << log_N, T = 1 << log_T, M = 1 << (log_N - log_T);
  float delta[3], inv_delta[3];
  for (int k = 0; k < 3; k++) { delta[k] = (bbox1[k] - bbox0[k]) / (float)N; inv_delta[k] = (float)N / (bbox1[k] - bbox0[k]); }
  frag_t *frags = NULL; size_t nf = 0, capf = 0;
  static const int UVW[3][3] = {{1, 2, 0}, {0, 2, 1}, {0, 1, 2}};  /* utils.h:114-176 */
  for (int t = 0; t < n_tris; t++) {
    const float *v0 = vbo + 9 * (size_t)t, *v1 = v0 + 3, *v2 = v0 + 6;
    /* coarse.h:59-102: bbox, integer bbox, dominant axis */
    float lo[3], hi[3]; int loi[3], hii[3];
    for (int k = 0; k < 3; k++) {
      lo[k] = (v0[k] - bbox0[k]) * inv_delta[k];
      lo[k] = fminf((v1[k] - bbox0[k]) * inv_delta[k], lo[k]);
      lo[k] = fminf((v2[k] - bbox0[k]) * inv_delta[k], lo[k]);
      hi[k] = (v0[k] - bbox0[k]) * inv_delta[k];
      hi[k] = fmaxf((v1[k] - bbox0[k]) * inv_delta[k], hi[k]);
      hi[k] = fmaxf((v2[k] - bbox0[k]) * inv_delta[k], hi[k]);
      int a = (int)lo[k]; if (a < 0) a = 0; if (a > N - 1) a = N - 1; loi[k] = a;
      int b = (int)ceilf(hi[k]); if (b < 0) b = 0; if (b > N - 1) b = N - 1; hii[k] = b;
    }
    const float edge0[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float edge1[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
    const float edge2[3] = {v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2]};
    /* anti_cross(edge0, edge2), utils.h:97-103 */
    const float n[3] = {edge0[2] * edge2[1] - edge0[1] * edge2[2], edge0[0] * edge2[2] - edge0[2] * edge2[0],
                        edge0[1] * edge2[0] - edge0[0] * edge2[1]};
    const int byx = fabsf(n[1]) > fabsf(n[0]), byz = fabsf(n[1]) > fabsf(n[2]), bzx = fabsf(n[2]) > fabsf(n[0]);
    const int axis = byx ? (byz ? 1 : 2) : (bzx ? 2 : 0);
    const int U = UVW[axis][0], V = UVW[axis][1], W = UVW[axis][2];
    const float sgn = axis == 0 ? (n[0] > 0.0f ? 1.0f : -1.0f) : axis == 1 ? (n[1] < 0.0f ? 1.0f : -1.0f) : (n[2] > 0.0f ? 1.0f : -1.0f);
    /* triangle_setup, utils.h:185-232 */
    const float *vv[3] = {v0, v1, v2};
    const float *ee[3] = {edge0, edge1, edge2};
    float a[3], ndu[3], ndv[3];
    for (int k = 0; k < 3; k++) {
      const float nx = -sel3(ee[k], V) * sgn, ny = sel3(ee[k], U) * sgn;
      const float d = -(nx * sel3(vv[k], U) + ny * sel3(vv[k], V)) + fmaxf(0.0f, delta[U] * nx) + fmaxf(0.0f, delta[V] * ny);
      a[k] = (nx * bbox0[U] + ny * bbox0[V]) + d;
      ndu[k] = nx * delta[U];
      ndv[k] = ny * delta[V];
    }
    /* plane_setup, utils.h:236-254 (__frcp_rn = correctly rounded reciprocal) */
    const float inv_n = 1.0f / n[W];
    const float px = n[U] * inv_n, py = n[V] * inv_n;
    const float pz = px * v0[U] + py * v0[V] + v0[W] - bbox0[W] - px * bbox0[U] - py * bbox0[V];
    const float inv_du[3] = {1.0f / ndu[0], 1.0f / ndu[1], 1.0f / ndu[2]};
    const uint32_t value = shade(t, tex, tex_w, tex_h, tbo, tbosize);
    /* tile loop: tiles overlapping the integer bbox that pass the plane test (fine.h:936-959) */
    for (int tz = loi[2] >> log_T; tz <= hii[2] >> log_T; tz++)
      for (int ty = loi[1] >> log_T; ty <= hii[1] >> log_T; ty++)
        for (int tx = loi[0] >> log_T; tx <= hii[0] >> log_T; tx++) {
          const int tile[3] = {tx << log_T, ty << log_T, tz << log_T};
          const float c[3] = {n[0] > 0 ? delta[0] * T : 0.0f, n[1] > 0 ? delta[1] * T : 0.0f, n[2] > 0 ? delta[2] * T : 0.0f};
          const float r1 = n[0] * (c[0] - v0[0]) + n[1] * (c[1] - v0[1]) + n[2] * (c[2] - v0[2]);
          const float r2 = n[0] * (delta[0] * T - c[0] - v0[0]) + n[1] * (delta[1] * T - c[1] - v0[1]) + n[2] * (delta[2] * T - c[2] - v0[2]);
          const float np = n[0] * (bbox0[0] + tile[0] * delta[0]) + n[1] * (bbox0[1] + tile[1] * delta[1]) + n[2] * (bbox0[2] + tile[2] * delta[2]);
          if (!((np + r1) * (np + r2) <= 0.0f)) continue;
          int b0[3], b1[3];
          for (int k = 0; k < 3; k++) { b0[k] = loi[k] > tile[k] ? loi[k] : tile[k]; b1[k] = hii[k] < tile[k] + T - 1 ? hii[k] : tile[k] + T - 1; }
          for (int v = b0[V]; v <= b1[V]; ++v) {  /* generate_mask, fine.h:229-262 */
            const float b[3] = {a[0] + (float)v * ndv[0], a[1] + (float)v * ndv[1], a[2] + (float)v * ndv[2]};
            int min_u = b0[U], max_u = b1[U];
            scanline_bounds(b, ndu, inv_du, &min_u, &max_u);
            if (min_u > max_u) continue;
            /* packed into LOG_TILE_SIZE-bit fields relative to the tile (fine.h:254-262) */
            const uint32_t lm = (uint32_t)(min_u - tile[U]) & (uint32_t)(T - 1), rm = (uint32_t)(max_u - tile[U]) & (uint32_t)(T - 1);
            const float vf = (v + 0.5f) * delta[V];
            for (int u = (int)lm + tile[U]; u <= (int)rm + tile[U]; ++u) {  /* rasterize_scanline, fine.h:318-365 */
              const float uf = (u + 0.5f) * delta[U];
              const float wf = pz - (px * uf + py * vf);
              const int w = (int)(wf * inv_delta[W]);
              if (w >= tile[W] && w < tile[W] + T) {
                int xyz[3];
                xyz[U] = u; xyz[V] = v; xyz[W] = w;
                const int64_t tl = (int64_t)(xyz[0] >> log_T) + (int64_t)M * (xyz[1] >> log_T) + (int64_t)M * M * (xyz[2] >> log_T);
                const int64_t pix = (xyz[0] & (T - 1)) + T * (xyz[1] & (T - 1)) + T * T * (xyz[2] & (T - 1));
                if (nf == capf) { capf = capf ? 2 * capf : 4096; frags = (frag_t *)realloc(frags, sizeof(frag_t) * capf); }
                frags[nf].idx = tl * T * T * T + pix; frags[nf].val = value; frags[nf].tri = t; nf++;
              }
            }
          }
        }
  }
  qsort(frags, nf, sizeof(frag_t), cmp_frag);
  size_t count = 0;
  for (size_t i = 0; i < nf; i++)
    if (i + 1 == nf || frags[i + 1].idx != frags[i].idx) frags[count++] = frags[i];  /* highest tri id of each voxel */
  *centers = (float *)malloc(sizeof(float) * 4 * (count ? count : 1));
  *colors = (float *)calloc(4 * (count ? count : 1), sizeof(float));
  if (indices) *indices = (int64_t *)malloc(sizeof(int64_t) * (count ? count : 1));
  /* createVoxelGrid / getCenterFromIndex, voxelization.cu:50-76,219-236 */
  float t_d[3], p_d[3];
  for (int k = 0; k < 3; k++) { t_d[k] = (bbox1[k] - bbox0[k]) / (float)M; p_d[k] = t_d[k] / (float)T; }
  const int64_t T3 = (int64_t)T * T * T;
  for (size_t i = 0; i < count; i++) {
    const int64_t idx = frags[i].idx;
    const int64_t tile_num = idx / T3, pix_num = idx % T3;
    const int tz = (int)(tile_num / ((int64_t)M * M) % M), pz = (int)(pix_num / (T * T) % T);
    const int ty = (int)(tile_num / M % M), py = (int)(pix_num / T % T);
    const int tx = (int)(tile_num % M), px = (int)(pix_num % T);
    float *ce = *centers + 4 * i, *co = *colors + 4 * i;
    ce[0] = bbox0[0] + tx * t_d[0] + px * p_d[0] + p_d[0] / 2.0f;
    ce[1] = bbox0[1] + ty * t_d[1] + py * p_d[1] + p_d[1] / 2.0f;
    ce[2] = bbox0[2] + tz * t_d[2] + pz * p_d[2] + p_d[2] / 2.0f;
    ce[3] = 1.0f;
    const int color = (int)frags[i].val;
    co[0] = (float)((color & 0xFF) / 255.0);
    co[1] = (float)(((color >> 8) & 0xFF) / 255.0);
    co[2] = (float)(((color >> 16) & 0xFF) / 255.0);
    if (indices) (*indices)[i] = idx;
  }
  free(frags);
  return (int)count;
}

/* createCubeMesh, voxelization.cu:184-217 */
void ora_voxel_grid_to_mesh(const float *centers, const float *colors, int n, float scale_factor, const float *cube_vbo,
                            int cube_vbosize, const int *cube_ibo, int cube_ibosize, const float *cube_nbo, float *out_vbo,
                            int *out_ibo, float *out_nbo, float *out_cbo) {
  for (int idx = 0; idx < n; idx++) {
    const int vbo_offset = idx * cube_vbosize, ibo_offset = idx * cube_ibosize;
    const float *c = centers + 4 * (size_t)idx, *k = colors + 4 * (size_t)idx;
    for (int i = 0; i < cube_vbosize; i++) {
      const int a = i % 3;
      out_vbo[vbo_offset + i] = cube_vbo[i] * scale_factor + c[a];
      out_cbo[vbo_offset + i] = k[a];
      out_nbo[vbo_offset + i] = cube_nbo[i];
    }
    for (int i = 0; i < cube_ibosize; i++) out_ibo[ibo_offset + i] = cube_ibo[i] + ibo_offset;
  }
}
