// radix_sort.hip -- stable LSD radix sort of (64-bit Morton key, 32-bit point
// index) pairs, 8-bit digits, hand-written for wave64.
//
// Replaces the D per-level thrust::sort calls of the reference
// (src/world/svo/svo.cu:216, :602) with ONE sort of the full-depth keys per
// fused frame; every per-level unique list is then a prefix property of the
// sorted array (see svo_build.hip).
//
// Per pass: upsweep (tile digit histogram) -> row_scan (exclusive scan of each
// digit row over tiles) -> downsweep (stable in-tile ranking by ballot match,
// scatter).  A tile is 256 threads x IPT items laid out wave-striped so that
// (wave, round, lane) order equals index order, which keeps the sort stable.
#include "radix_sort.hpp"
#include "wave_rank.hpp"

namespace svoslam {

constexpr int kSortIPT = 4;
constexpr int kSortTile = 256 * kSortIPT;

__global__ __launch_bounds__(256) void radix_upsweep_kernel(const unsigned long long *__restrict__ keys, int n, int shift,
                                                            unsigned mask, unsigned *__restrict__ tile_hist,
                                                            int num_tiles) {
  __shared__ unsigned hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int tile = blockIdx.x;
  const long long base = (long long)tile * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortIPT; r++) {
    const long long idx = base + r * 256 + threadIdx.x;
    if (idx < n) atomicAdd(&hist[(unsigned)(keys[idx] >> shift) & mask], 1u);
  }
  __syncthreads();
  tile_hist[(size_t)threadIdx.x * num_tiles + tile] = hist[threadIdx.x];
}

// One workgroup per row: in-place exclusive scan of row[0..num_tiles), row total to totals[row].
__global__ __launch_bounds__(256) void row_scan_kernel(unsigned *__restrict__ rows, int num_tiles,
                                                       unsigned *__restrict__ totals) {
  __shared__ unsigned tmp[4];
  unsigned *row = rows + (size_t)blockIdx.x * num_tiles;
  unsigned carry = 0;
  for (int base = 0; base < num_tiles; base += 256) {
    const int i = base + threadIdx.x;
    const unsigned v = i < num_tiles ? row[i] : 0u;
    unsigned total;
    const unsigned ex = block256_exclusive_scan(v, tmp, total);
    if (i < num_tiles) row[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void radix_downsweep_kernel(const unsigned long long *__restrict__ keys_in,
                                                              const unsigned *__restrict__ vals_in,
                                                              unsigned long long *__restrict__ keys_out,
                                                              unsigned *__restrict__ vals_out, int n, int shift,
                                                              unsigned mask, const unsigned *__restrict__ row_prefix,
                                                              const unsigned *__restrict__ totals, int num_tiles,
                                                              int iota_vals) {
  __shared__ unsigned cnt[4][256];  // per-wave digit counters, then per-wave exclusive offsets
  __shared__ unsigned dbase[256];   // global output base of each digit for this tile
  __shared__ unsigned tmp[4];
  const int tile = blockIdx.x;
  const unsigned wave = threadIdx.x >> 6, lane = lane_id();
#pragma unroll
  for (int w = 0; w < 4; w++) cnt[w][threadIdx.x] = 0;
  {
    unsigned total;
    const unsigned ex = block256_exclusive_scan(totals[threadIdx.x], tmp, total);
    dbase[threadIdx.x] = ex + row_prefix[(size_t)threadIdx.x * num_tiles + tile];
  }
  __syncthreads();

  const long long base = (long long)tile * kSortTile + (long long)wave * (kWave * kSortIPT);
  unsigned long long k[kSortIPT];
  unsigned v[kSortIPT], dig[kSortIPT], pre[kSortIPT];
  bool ok[kSortIPT];
  volatile unsigned *wc = cnt[wave];
  const unsigned long long lt = lanemask_lt();
#pragma unroll
  for (int r = 0; r < kSortIPT; r++) {
    const long long idx = base + r * kWave + lane;
    ok[r] = idx < n;
    k[r] = ok[r] ? keys_in[idx] : 0ull;
    v[r] = ok[r] ? (iota_vals ? (unsigned)idx : vals_in[idx]) : 0u;
    dig[r] = (unsigned)(k[r] >> shift) & mask;
    const unsigned long long peers = match_digit8(ok[r], dig[r]);
    pre[r] = 0;
    if (ok[r]) {
      const unsigned old = wc[dig[r]];
      const unsigned rank = __popcll(peers & lt);
      pre[r] = old + rank;
      if (rank == 0) wc[dig[r]] = old + __popcll(peers);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {  // exclusive offsets of each wave's share of digit threadIdx.x
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const unsigned t = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kSortIPT; r++) {
    if (ok[r]) {
      const unsigned pos = dbase[dig[r]] + cnt[wave][dig[r]] + pre[r];
      keys_out[pos] = k[r];
      vals_out[pos] = v[r];
    }
  }
}

// Long arrays: exclusive scan in chunks of 256 threads x 8 items.  Pass 1 sums each chunk, a single
// workgroup scans the chunk sums (row_scan_kernel), pass 2 scans inside the chunks on top of them.
constexpr int kScanItems = 8;
constexpr int kScanChunk = 256 * kScanItems;

__global__ __launch_bounds__(256) void scan_chunk_sum_kernel(const unsigned *__restrict__ data, unsigned n,
                                                             unsigned *__restrict__ sums) {
  __shared__ unsigned tmp[4];
  const unsigned base = blockIdx.x * (unsigned)kScanChunk + threadIdx.x * kScanItems;
  unsigned v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) v += base + k < n ? data[base + k] : 0u;
  unsigned total;
  (void)block256_exclusive_scan(v, tmp, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_chunk_apply_kernel(unsigned *__restrict__ data, unsigned n,
                                                               const unsigned *__restrict__ sums) {
  __shared__ unsigned tmp[4];
  const unsigned base = blockIdx.x * (unsigned)kScanChunk + threadIdx.x * kScanItems;
  unsigned item[kScanItems], v = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) { item[k] = base + k < n ? data[base + k] : 0u; v += item[k]; }
  unsigned total;
  unsigned run = sums[blockIdx.x] + block256_exclusive_scan(v, tmp, total);
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if (base + k < n) data[base + k] = run;
    run += item[k];
  }
}

int exclusive_scan_u32(svoslam_workspace *ws, unsigned *data, unsigned n, unsigned *total, hipStream_t stream) {
  if (n <= 4u * kScanChunk) {
    row_scan_kernel<<<1, 256, 0, stream>>>(data, (int)n, total);
    return SVOSLAM_OK;
  }
  const unsigned chunks = (unsigned)cdiv(n, kScanChunk);
  SVO_TRY(ws->scan_tmp.reserve((size_t)chunks * 4));
  unsigned *sums = ws->scan_tmp.as<unsigned>();
  scan_chunk_sum_kernel<<<chunks, 256, 0, stream>>>(data, n, sums);
  row_scan_kernel<<<1, 256, 0, stream>>>(sums, (int)chunks, total);
  scan_chunk_apply_kernel<<<chunks, 256, 0, stream>>>(data, n, sums);
  SVO_LAUNCH_CHECK();
  return SVOSLAM_OK;
}

// ---- packed sort (round 2): the asynchronous fusion's keys ---------------------------------------------
// A depth-D Morton key is 3D + 1 bits and a point index ceil(log2 n) bits: at 640x480 / depth 12 that is 37 + 19, at
// 1920x1080 / depth 14 43 + 21 = 64.  Where both fit one 64-bit word the sort carries (key << idx_bits | index)
// and no value array -- 8 instead of 12 bytes per element and pass -- with digits of up to 11 bits: 4 passes instead of
// 5 (depth 12) or 6 (depth 14).  Tiles of 2048 keys keep the [tile][digit] histograms (written as whole rows, read as
// whole rows) small next to the keys; the column scan over the tiles is its own launch (one coalesced pass on bins / 64
// workgroups: folding it into the downsweep was measured slower).  The first pass's histogram comes
// from whoever writes the keys (svo_build.hip: keys_packed_kernel); the last pass unpacks into the (key, index) arrays
// the planner reads.  Stable by construction: equal keys differ in their index bits' original order.
constexpr int kPkThreads = 512, kPkWaves = kPkThreads / 64;
constexpr int kPkIPT = 4;
constexpr int kPkTile = kPkThreads * kPkIPT;  // 2048: at 640x480 150 workgroups of 8 wavefronts (tiles of 4096 keys on 256 threads
// were measured 1.8x slower per pass: too few wavefronts in flight for a latency-bound kernel)

int radix_packed_tile() { return kPkTile; }
int radix_packed_threads() { return kPkThreads; }
int radix_packed_tiles(int n) { return (int)cdiv(n, kPkTile); }
int radix_packed_passes(int key_bits) { int p = (key_bits + kPackedMaxBits - 1) / kPackedMaxBits; return p < 1 ? 1 : p; }
int radix_packed_first_bits(int key_bits) {
  const int p = radix_packed_passes(key_bits);
  const int w = (key_bits + p - 1) / p;
  return w < 1 ? 1 : w;
}

__global__ __launch_bounds__(kPkThreads) void packed_upsweep_kernel(const unsigned long long *__restrict__ keys, int n, int shift, int bits,
                                                                    unsigned *__restrict__ tile_hist) {
  __shared__ unsigned hist[kPackedMaxBins];
  const int bins = 1 << bits;
  for (int d = threadIdx.x; d < bins; d += kPkThreads) hist[d] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * kPkTile;
  const unsigned mask = (unsigned)bins - 1u;
#pragma unroll
  for (int r = 0; r < kPkIPT; r++) {
    const long long idx = base + r * kPkThreads + threadIdx.x;
    if (idx < n) atomicAdd(&hist[(unsigned)(keys[idx] >> shift) & mask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += kPkThreads) tile_hist[(size_t)blockIdx.x * bins + d] = hist[d];
}

// hist[t][d] (t < tiles, d < bins) -> exclusive prefix over t, in place; totals[d] = column sum.  One workgroup of 16
// wavefronts per 64 digits: wavefront w scans the tiles [w C, (w + 1) C) of its 64 columns (256-byte row segments).
// MODE 0: the whole matrix by bins / 64 workgroups (a frame's keys: <= 1013 tiles).  Large inputs (round 5: config 5 sorts 375 M
// fragments = 183 000 tiles through 8-bit digits -- FOUR workgroups walked a 187 MB matrix, 2.3 ms per pass and 30 ms of the
// configuration's 130) cut the tiles into chunks (blockIdx.y): MODE 1 writes each chunk's column sums to chunk_mat[chunk][d],
// MODE 0 scans that small matrix in place, MODE 2 scans every chunk again on top of its entry.
template <int MODE>
__global__ __launch_bounds__(1024) void packed_column_scan_kernel(unsigned *__restrict__ hist, int tiles, int bins,
                                                                  unsigned *__restrict__ totals, unsigned *__restrict__ chunk_mat,
                                                                  int chunk_tiles) {
  __shared__ unsigned csum[16][64];
  const int lane = (int)(threadIdx.x & 63u), w = (int)(threadIdx.x >> 6);
  const int d = blockIdx.x * 64 + lane;
  const int c_begin = MODE == 0 ? 0 : (int)blockIdx.y * chunk_tiles;
  int c_end = MODE == 0 ? tiles : c_begin + chunk_tiles;
  if (c_end > tiles) c_end = tiles;
  const int span = c_end > c_begin ? c_end - c_begin : 0;
  const int C = (span + 15) / 16;
  const int t0 = c_begin + w * C, t1 = (t0 + C < c_end) ? t0 + C : c_end;
  const bool col = d < bins;
  unsigned sum = 0;
  for (int t = t0; t < t1; t += 8) {
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (col && t + k < t1) ? hist[(size_t)(t + k) * bins + d] : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) sum += v[k];
  }
  csum[w][lane] = sum;
  __syncthreads();
  unsigned run = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const unsigned c = csum[k][lane];
    if (k < w) run += c;
    total += c;
  }
  if (MODE == 1) {
    if (w == 0 && col) chunk_mat[(size_t)blockIdx.y * bins + d] = total;
    return;
  }
  if (MODE == 2 && col) run += chunk_mat[(size_t)blockIdx.y * bins + d];
  for (int t = t0; t < t1; t += 8) {
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (col && t + k < t1) ? hist[(size_t)(t + k) * bins + d] : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (col && t + k < t1) hist[(size_t)(t + k) * bins + d] = run;
      run += v[k];
    }
  }
  if (MODE == 0 && w == 0 && col) totals[d] = total;
}

// the column scan of one pass: one launch, or three for large matrices (see above); chunk_mat: >= 256 x bins words
static void packed_column_scan(unsigned *tile_hist, int tiles, int bins, unsigned *totals, unsigned *chunk_mat, hipStream_t stream) {
  const unsigned gx = (unsigned)(bins + 63) / 64;
  if (tiles <= 8192 || !chunk_mat) {
    packed_column_scan_kernel<0><<<gx, 1024, 0, stream>>>(tile_hist, tiles, bins, totals, nullptr, 0);
    return;
  }
  int chunks = tiles / 1024;
  if (chunks > 256) chunks = 256;
  const int chunk_tiles = (tiles + chunks - 1) / chunks;
  chunks = (tiles + chunk_tiles - 1) / chunk_tiles;
  packed_column_scan_kernel<1><<<dim3(gx, (unsigned)chunks), 1024, 0, stream>>>(tile_hist, tiles, bins, nullptr, chunk_mat, chunk_tiles);
  packed_column_scan_kernel<0><<<gx, 1024, 0, stream>>>(chunk_mat, chunks, bins, totals, nullptr, 0);
  packed_column_scan_kernel<2><<<dim3(gx, (unsigned)chunks), 1024, 0, stream>>>(tile_hist, tiles, bins, nullptr, chunk_mat, chunk_tiles);
}

template <bool LAST>
__global__ __launch_bounds__(kPkThreads) void packed_downsweep_kernel(const unsigned long long *__restrict__ in,
                                                                      unsigned long long *__restrict__ out_keys,
                                                                      unsigned *__restrict__ out_vals, int n, int shift, int bits,
                                                                      int idx_bits, const unsigned *__restrict__ tile_prefix,
                                                                      const unsigned *__restrict__ totals) {
  __shared__ unsigned short cnt[kPkWaves][kPackedMaxBins];  // per-wave digit counters (<= 256), then per-wave exclusive offsets
  __shared__ unsigned dbase[kPackedMaxBins];                // global output base of each digit for this tile
  __shared__ unsigned wsum[kPkWaves];
  const int bins = 1 << bits;
  const unsigned mask = (unsigned)bins - 1u;
  const int tile = blockIdx.x;
  const unsigned wave = threadIdx.x >> 6, lane = lane_id();
  // keys first: their latency covers the digit-base scan
  const long long base = (long long)tile * kPkTile + (long long)wave * (kWave * kPkIPT);
  unsigned long long k[kPkIPT];
#pragma unroll
  for (int r = 0; r < kPkIPT; r++) {
    const long long idx = base + r * kWave + lane;
    k[r] = idx < n ? in[idx] : ~0ull;
  }
  for (int d = threadIdx.x; d < bins; d += kPkThreads) {
#pragma unroll
    for (int w = 0; w < kPkWaves; w++) cnt[w][d] = 0;
  }
  {  // exclusive scan of the digit totals: thread t owns the `per` consecutive digits from t * per
    const int per = (bins + kPkThreads - 1) / kPkThreads;
    const int d0 = (int)threadIdx.x * per;
    unsigned own = 0;
    for (int q = 0; q < per; q++) own += d0 + q < bins ? totals[d0 + q] : 0u;
    const unsigned inc = wave_inclusive_scan(own);
    if (lane == kWave - 1) wsum[wave] = inc;
    __syncthreads();
    unsigned run = inc - own;
#pragma unroll
    for (int w = 0; w < kPkWaves; w++) run += (unsigned)w < wave ? wsum[w] : 0u;
    for (int q = 0; q < per; q++) {
      if (d0 + q < bins) {
        dbase[d0 + q] = run + tile_prefix[(size_t)tile * bins + d0 + q];
        run += totals[d0 + q];
      }
    }
  }
  __syncthreads();
  unsigned short pre[kPkIPT];
  volatile unsigned short *wc = cnt[wave];
  const unsigned long long lt = lanemask_lt();
#pragma unroll
  for (int r = 0; r < kPkIPT; r++) {
    const bool ok = base + r * kWave + lane < n;
    const unsigned dig = (unsigned)(k[r] >> shift) & mask;
    unsigned long long peers = __ballot(ok);
    for (int b = 0; b < bits; b++) {
      const bool bit = (dig >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    pre[r] = 0;
    if (ok) {
      const unsigned old = wc[dig];
      const unsigned rank = __popcll(peers & lt);
      pre[r] = (unsigned short)(old + rank);
      if (rank == 0) wc[dig] = (unsigned short)(old + __popcll(peers));
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (int d = threadIdx.x; d < bins; d += kPkThreads) {  // exclusive offsets of each wave's share of digit d
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < kPkWaves; w++) {
      const unsigned t = cnt[w][d];
      cnt[w][d] = (unsigned short)run;
      run += t;
    }
  }
  __syncthreads();
  const unsigned long long imask = (1ull << idx_bits) - 1ull;
#pragma unroll
  for (int r = 0; r < kPkIPT; r++) {
    if (base + r * kWave + lane < n) {
      const unsigned dig = (unsigned)(k[r] >> shift) & mask;
      const unsigned pos = dbase[dig] + cnt[wave][dig] + pre[r];
      if (LAST) {
        out_keys[pos] = k[r] >> idx_bits;
        if (out_vals) out_vals[pos] = (unsigned)(k[r] & imask);
      } else {
        out_keys[pos] = k[r];
      }
    }
  }
}

// Sorts the n packed words of ws->keys_a on their key bits [idx_bits, idx_bits + key_bits).  ws->tile_hist already holds
// the first pass's histograms ([tile][1 << radix_packed_first_bits(key_bits)], tiles of radix_packed_tile() elements).
// Output: keys (unpacked) in keys_a or keys_b, indices in vals_a.
int radix_sort_packed(svoslam_workspace *ws, int n, int key_bits, int idx_bits, hipStream_t stream, unsigned long long **sorted_keys,
                      unsigned **sorted_vals) {
  return radix_sort_packed_ex(ws, n, key_bits, idx_bits, kPackedMaxBits, true, true, stream, sorted_keys, sorted_vals);
}

// digit width by size: the [tile][digit] matrix of a pass is tiles x 2^bits words -- as large as half the keys at 11 bits, which
// is fine while everything sits in the L2 / Infinity Cache (a frame's 0.3-2 M keys) and a third of the traffic beyond that
int radix_packed_digit_bits_for(long long n) { return n <= (4ll << 20) ? kPackedMaxBits : (n <= (64ll << 20) ? 9 : 8); }

int radix_sort_packed_ex(svoslam_workspace *ws, int n, int key_bits, int idx_bits, int max_bits, bool have_first_hist, bool want_vals,
                         hipStream_t stream, unsigned long long **sorted_keys, unsigned **sorted_vals) {
  if (max_bits < 1 || max_bits > kPackedMaxBits) return SVOSLAM_ERR_INVALID_ARG;
  const int tiles = radix_packed_tiles(n);
  const size_t row_words = (size_t)1 << max_bits;  // (rows are written with the pass's own width as their stride: <= this)
  SVO_TRY(ws->tile_hist.reserve(((size_t)tiles + 1) * row_words * 4));
  unsigned long long *ka = ws->keys_a.as<unsigned long long>(), *kb = ws->keys_b.as<unsigned long long>();
  unsigned *tile_hist = ws->tile_hist.as<unsigned>();
  int passes = (key_bits + max_bits - 1) / max_bits;
  if (passes < 1) passes = 1;
  unsigned *totals = tile_hist + (size_t)tiles * row_words;  // behind the largest histogram matrix
  unsigned *chunk_mat = nullptr;  // large inputs: the chunks' column sums (packed_column_scan)
  if (tiles > 8192) {
    SVO_TRY(ws->scan_tmp.reserve((size_t)256 * row_words * 4));
    chunk_mat = ws->scan_tmp.as<unsigned>();
  }
  int bit = 0;
  for (int p = 0; p < passes; p++) {
    const int remaining_bits = key_bits - bit, remaining_passes = passes - p;
    int width = (remaining_bits + remaining_passes - 1) / remaining_passes;
    if (width < 1) width = 1;
    const int bins = 1 << width, shift = idx_bits + bit;
    if (p > 0 || !have_first_hist) packed_upsweep_kernel<<<tiles, kPkThreads, 0, stream>>>(ka, n, shift, width, tile_hist);
    packed_column_scan(tile_hist, tiles, bins, totals, chunk_mat, stream);
    if (p == passes - 1)
      packed_downsweep_kernel<true><<<tiles, kPkThreads, 0, stream>>>(ka, kb, want_vals ? ws->vals_a.as<unsigned>() : nullptr, n, shift, width, idx_bits, tile_hist, totals);
    else
      packed_downsweep_kernel<false><<<tiles, kPkThreads, 0, stream>>>(ka, kb, nullptr, n, shift, width, idx_bits, tile_hist, totals);
    SVO_LAUNCH_CHECK();
    unsigned long long *tk = ka; ka = kb; kb = tk;
    bit += width;
  }
  *sorted_keys = ka;
  *sorted_vals = want_vals ? ws->vals_a.as<unsigned>() : nullptr;
  return SVOSLAM_OK;
}

int radix_sort_packed_output(svoslam_workspace *ws, int key_bits, unsigned long long **sorted_keys, unsigned **sorted_vals) {
  const bool in_b = (radix_packed_passes(key_bits) & 1) != 0;
  *sorted_keys = in_b ? ws->keys_b.as<unsigned long long>() : ws->keys_a.as<unsigned long long>();
  *sorted_vals = ws->vals_a.as<unsigned>();
  return SVOSLAM_OK;
}

int radix_sort_num_tiles(int n) { return (int)cdiv(n, kSortTile); }

void row_scan_rows(unsigned *rows, int num_tiles, unsigned *totals, hipStream_t stream) {
  row_scan_kernel<<<256, 256, 0, stream>>>(rows, num_tiles, totals);
}
void row_scan_rows1(unsigned *row, int num_tiles, unsigned *total, hipStream_t stream) {
  row_scan_kernel<<<1, 256, 0, stream>>>(row, num_tiles, total);
}

int radix_sort_pairs(svoslam_workspace *ws, int n, int num_bits, hipStream_t stream, unsigned long long **sorted_keys,
                     unsigned **sorted_vals, bool iota_vals) {
  // keys are in ws->keys_a; values are the identity permutation (generated in the first pass) or ws->vals_a
  unsigned long long *ka = ws->keys_a.as<unsigned long long>(), *kb = ws->keys_b.as<unsigned long long>();
  unsigned *va = ws->vals_a.as<unsigned>(), *vb = ws->vals_b.as<unsigned>();
  unsigned *tile_hist = ws->tile_hist.as<unsigned>();
  unsigned *totals = ws->small.as<unsigned>();  // first 256 words
  const int tiles = radix_sort_num_tiles(n);
  int passes = (num_bits + 7) / 8;
  if (passes < 1) passes = 1;
  int bit = 0;
  for (int p = 0; p < passes; p++) {
    // spread the bits evenly over the passes (e.g. 37 bits -> 8,8,7,7,7)
    const int remaining_bits = num_bits - bit, remaining_passes = passes - p;
    int width = (remaining_bits + remaining_passes - 1) / remaining_passes;
    if (width < 1) width = 1;
    const unsigned mask = (1u << width) - 1u;
    radix_upsweep_kernel<<<tiles, 256, 0, stream>>>(ka, n, bit, mask, tile_hist, tiles);
    row_scan_kernel<<<256, 256, 0, stream>>>(tile_hist, tiles, totals);
    radix_downsweep_kernel<<<tiles, 256, 0, stream>>>(ka, va, kb, vb, n, bit, mask, tile_hist, totals, tiles, (p == 0 && iota_vals) ? 1 : 0);
    SVO_LAUNCH_CHECK();
    unsigned long long *tk = ka; ka = kb; kb = tk;
    unsigned *tv = va; va = vb; vb = tv;
    bit += width;
  }
  *sorted_keys = ka;
  *sorted_vals = va;
  return SVOSLAM_OK;
}

// which ping-pong buffers a radix_sort_pairs of num_bits leaves its result in (for replayed graphs)
int radix_sort_output(svoslam_workspace *ws, int n, int num_bits, unsigned long long **sorted_keys, unsigned **sorted_vals) {
  (void)n;
  int passes = (num_bits + 7) / 8;
  if (passes < 1) passes = 1;
  const bool in_b = (passes & 1) != 0;
  *sorted_keys = in_b ? ws->keys_b.as<unsigned long long>() : ws->keys_a.as<unsigned long long>();
  *sorted_vals = in_b ? ws->vals_b.as<unsigned>() : ws->vals_a.as<unsigned>();
  return SVOSLAM_OK;
}

}  // namespace svoslam
