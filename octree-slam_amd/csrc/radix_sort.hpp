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
// In-place exclusive scan of each of 256 rows of num_tiles counters; row totals to totals[256].
void row_scan_rows(unsigned *rows, int num_tiles, unsigned *totals, hipStream_t stream);
// Same for a single row.
void row_scan_rows1(unsigned *row, int num_tiles, unsigned *total, hipStream_t stream);
// in-place exclusive scan of data[0..n) on any number of workgroups; *total (device) = the sum
int exclusive_scan_u32(svoslam_workspace *ws, unsigned *data, unsigned n, unsigned *total, hipStream_t stream);
}  // namespace svoslam
