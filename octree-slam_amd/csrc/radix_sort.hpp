// radix_sort.hpp -- see radix_sort.hip
#pragma once
#include "common.hpp"
#include "workspace.hpp"

namespace svoslam {
int radix_sort_num_tiles(int n);
// Sorts the n keys stored in ws->keys_a on bits [0, num_bits), carrying the
// original index of each key as the value.  Stable.  Uses keys_b/vals_a/vals_b,
// tile_hist (>= 256*tiles words) and the first 256 words of ws->small.
// iota_vals = false: the values to carry are in ws->vals_a.
int radix_sort_pairs(svoslam_workspace *ws, int n, int num_bits, hipStream_t stream,
                     unsigned long long **sorted_keys, unsigned **sorted_vals, bool iota_vals = true);
// Where that sort leaves its result (depends only on the pass count).
int radix_sort_output(svoslam_workspace *ws, int n, int num_bits, unsigned long long **sorted_keys, unsigned **sorted_vals);
// Packed form (see radix_sort.hip): n words (key << idx_bits | index) in ws->keys_a, first-pass histograms in
// ws->tile_hist ([tile][1 << radix_packed_first_bits(key_bits)] for tiles of radix_packed_tile() elements; reserve
// radix_packed_hist_words(n) words).  Output: unpacked keys in keys_a / keys_b, indices in vals_a.
constexpr int kPackedMaxBits = 11;
constexpr int kPackedMaxBins = 1 << kPackedMaxBits;
int radix_packed_tile();
int radix_packed_threads();
int radix_packed_tiles(int n);
int radix_packed_passes(int key_bits);
int radix_packed_first_bits(int key_bits);
inline size_t radix_packed_hist_words(int n) { return ((size_t)radix_packed_tiles(n) + 1) * kPackedMaxBins; }
int radix_sort_packed(svoslam_workspace *ws, int n, int key_bits, int idx_bits, hipStream_t stream,
                      unsigned long long **sorted_keys, unsigned **sorted_vals);
// The general form (round 5: the blocking insert and the mesh voxelizer's fragment sort use it too).  max_bits: digit width
// (<= kPackedMaxBits; large inputs take narrower digits -- the [tile][digit] matrix of an 11-bit pass is n x 4 bytes, half the keys);
// have_first_hist = false: the first pass counts its own histogram (an upsweep like the later passes'); want_vals = false: the
// low idx_bits are not unpacked (keys-only sorts: svoFromVoxelGrid pairs sorted key i with colour i, Q20).  Reserves what it needs.
int radix_sort_packed_ex(svoslam_workspace *ws, int n, int key_bits, int idx_bits, int max_bits, bool have_first_hist, bool want_vals,
                         hipStream_t stream, unsigned long long **sorted_keys, unsigned **sorted_vals);
int radix_packed_digit_bits_for(long long n);  // the digit width radix_sort_packed_ex callers use for n elements
int radix_sort_packed_output(svoslam_workspace *ws, int key_bits, unsigned long long **sorted_keys, unsigned **sorted_vals);
// In-place exclusive scan of each of 256 rows of num_tiles counters; row totals to totals[256].
void row_scan_rows(unsigned *rows, int num_tiles, unsigned *totals, hipStream_t stream);
// Same for a single row.
void row_scan_rows1(unsigned *row, int num_tiles, unsigned *total, hipStream_t stream);
// in-place exclusive scan of data[0..n) on any number of workgroups; *total (device) = the sum
int exclusive_scan_u32(svoslam_workspace *ws, unsigned *data, unsigned n, unsigned *total, hipStream_t stream);
}  // namespace svoslam
