// graph_cache.hpp -- replay of fixed launch sequences as HIP graphs.
//
// A tracked frame is ~50 launches of 4-10 us kernels and a fusion ~35: enqueued one by
// one the HOST becomes the bottleneck (~5 us per launch, measured 0.6 ms per frame against
// 0.45 ms of device work).  Each such sequence is a pure function of a few pointers and
// sizes, so the first call with a given key records it (stream capture) and later calls
// replay the instantiated graph with one hipGraphLaunch.
//
// Rules for the enqueue functor: launches / async memsets on the given stream only -- no
// allocation, no synchronisation, no host-state change (it may run twice: once to record,
// or again directly when recording is not possible).
#pragma once

#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "config.hpp"

namespace svoslam {

struct GraphKey {
  unsigned long long v[10];
  int used = 0;
  GraphKey() { memset(v, 0, sizeof(v)); }
  GraphKey &add(unsigned long long x) { if (used < 10) v[used++] = x; else v[9] = v[9] * 1099511628211ull ^ x; return *this; }
  GraphKey &add(const void *p) { return add((unsigned long long)(uintptr_t)p); }
  GraphKey &addf(float f) { uint32_t u; memcpy(&u, &f, 4); return add((unsigned long long)u); }
  bool operator==(const GraphKey &o) const { return used == o.used && memcmp(v, o.v, sizeof(v)) == 0; }
};

// Off by default since round 2: with the tracker in one launch a frame is ~37 launches, which the host enqueues in
// ~0.15 ms against a 0.4 ms frame, and direct launches turned out 4-5 % faster than replaying the same sequences as
// graphs (2494 against 2380 frames/s at cfg3; no difference at cfg4).  svoslam_config.graphs = 1 turns the replay on -- worth
// it where the host is the bottleneck (several ranks sharing few cores, the 38-launch chain tracker on a slow host).
inline bool graphs_enabled() {
  const bool on = config().graphs != 0;
  return on;
}

class GraphCache {
 public:
  ~GraphCache() { clear(); }
  void clear() {
    for (auto &e : entries_) (void)hipGraphExecDestroy(e.exec);
    entries_.clear();
  }
  template <class F>
  int run(const GraphKey &key, hipStream_t s, F &&enqueue) {
    if (!graphs_enabled() || s == nullptr) return enqueue();  // the legacy default stream cannot be captured
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return enqueue();  // caller is recording
    for (size_t i = 0; i < entries_.size(); i++) {
      if (entries_[i].key == key) {
        entries_[i].stamp = ++clock_;
        SVO_HIP(hipGraphLaunch(entries_[i].exec, s));
        return SVOSLAM_OK;
      }
    }
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return enqueue(); }
    const int rc = enqueue();
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != SVOSLAM_OK || e != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();
      return rc != SVOSLAM_OK ? rc : enqueue();
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess || !exec) { (void)hipGetLastError(); return enqueue(); }
    if (entries_.size() >= kMaxEntries) {  // evict the least recently used
      size_t lru = 0;
      for (size_t i = 1; i < entries_.size(); i++) if (entries_[i].stamp < entries_[lru].stamp) lru = i;
      (void)hipGraphExecDestroy(entries_[lru].exec);
      entries_.erase(entries_.begin() + (long)lru);
    }
    entries_.push_back(Entry{key, exec, ++clock_});
    SVO_HIP(hipGraphLaunch(exec, s));
    return SVOSLAM_OK;
  }

 private:
  struct Entry { GraphKey key; hipGraphExec_t exec; unsigned long long stamp; };
  static constexpr size_t kMaxEntries = 16;
  std::vector<Entry> entries_;
  unsigned long long clock_ = 0;
};

}  // namespace svoslam
