// stage_timing.hip -- see stage_timing.hpp
#include <mutex>
#include <vector>

#include "stage_timing.hpp"

namespace svoslam {

namespace {
std::mutex g_mu;  // stages are enqueued from several host threads / on several streams
unsigned g_mask = 0;
struct Entry { hipEvent_t start = nullptr, stop = nullptr; bool done = false; };
// gen: advanced whenever THIS stage's log is cleared, so that a bracket that began before does not end into the new log.  Per stage
// (ADVICE r04): reading stage A while stage B's brackets are open on other streams used to drop B's open brackets as well
struct Log { std::vector<Entry> e; size_t used = 0; unsigned gen = 1; };  // entries since the last read
Log g_log[kStageCount];
}  // namespace

unsigned stage_timing_mask() {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_mask;
}

int stage_timing(unsigned mask) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_mask = mask;
  for (Log &l : g_log) { l.used = 0; l.gen++; }
  return SVOSLAM_OK;
}

int stage_timing_read(int stage, float *ms_sum, int *pairs) {
  if (stage < 0 || stage >= kStageCount || !ms_sum || !pairs) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  Log &l = g_log[stage];
  float total = 0.0f;
  int n = 0;
  for (size_t i = 0; i < l.used; i++) {
    if (!l.e[i].done) continue;  // (a bracket still open, or one whose second record failed)
    SVO_HIP(hipEventSynchronize(l.e[i].stop));
    float ms = 0.0f;
    SVO_HIP(hipEventElapsedTime(&ms, l.e[i].start, l.e[i].stop));
    total += ms;
    n++;
  }
  *ms_sum = total;
  *pairs = n;
  l.used = 0;
  l.gen++;
  return SVOSLAM_OK;
}

int stage_begin(int stage, hipStream_t stream, long long *token) {
  if (token) *token = -1;
  if (stage < 0 || stage >= kStageCount || !token) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  if (!(g_mask & (1u << stage))) return SVOSLAM_OK;
  Log &l = g_log[stage];
  if (l.used == l.e.size()) {
    Entry en;
    SVO_HIP(hipEventCreate(&en.start));
    if (hipEventCreate(&en.stop) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(en.start); return SVOSLAM_ERR_HIP; }
    l.e.push_back(en);
  }
  Entry &en = l.e[l.used];
  en.done = false;
  SVO_HIP(hipEventRecord(en.start, stream));  // (on failure the entry is not taken)
  // generation in bits 32..62 (the token stays non-negative: -1 means "stage off"), entry index below
  *token = (long long)((((unsigned long long)l.gen & 0x7FFFFFFFull) << 32) | (unsigned long long)l.used);
  l.used++;
  return SVOSLAM_OK;
}

int stage_end(int stage, long long token, hipStream_t stream) {
  if (token < 0) return SVOSLAM_OK;
  if (stage < 0 || stage >= kStageCount) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  Log &l = g_log[stage];
  if ((unsigned)((unsigned long long)token >> 32) != (l.gen & 0x7FFFFFFFu)) return SVOSLAM_OK;  // the log was cleared in between: the bracket is dropped
  const size_t i = (size_t)(token & 0xFFFFFFFFll);
  if (i >= l.used) return SVOSLAM_OK;
  SVO_HIP(hipEventRecord(l.e[i].stop, stream));
  l.e[i].done = true;
  return SVOSLAM_OK;
}

}  // namespace svoslam
