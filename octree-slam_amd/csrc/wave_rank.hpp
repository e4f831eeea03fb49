// wave_rank.hpp -- wave64 ballot-based multi-split ranking and block scans.
//
// These are the primitives behind (a) the LSD radix sort of the Morton keys and
// (b) the split-record numbering of the fusion planner: both need, for every
// item, its stable rank among the items of the same 8-bit "digit" -- first in
// its wavefront (ballot match + popcount prefix), then across the 4 wavefronts
// of the workgroup (LDS counters), then across workgroups (a row scan kernel).
#pragma once

#include "common.hpp"

namespace svoslam {

__device__ inline unsigned lane_id() { return threadIdx.x & (kWave - 1); }
__device__ inline unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Lanes whose (valid, digit) equal mine.  8 ballots for an 8-bit digit.
// Result is meaningful only for valid lanes.
__device__ inline unsigned long long match_digit8(bool valid, unsigned digit) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; b++) {
    const bool bit = (digit >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// wave64 inclusive scan (Hillis-Steele over __shfl_up)
__device__ inline unsigned wave_inclusive_scan(unsigned v) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    unsigned t = __shfl_up(v, o);
    if ((int)lane_id() >= o) v += t;
  }
  return v;
}

// 256-thread exclusive scan; `tmp` is a 4-entry LDS array.  Returns the
// exclusive prefix of v for this thread and the block total in `total`.
__device__ inline unsigned block256_exclusive_scan(unsigned v, unsigned *tmp, unsigned &total) {
  const unsigned wave = threadIdx.x >> 6;
  const unsigned inc = wave_inclusive_scan(v);
  if (lane_id() == kWave - 1) tmp[wave] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const unsigned t = tmp[w];
    if ((unsigned)w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + inc - v;
}

}  // namespace svoslam
