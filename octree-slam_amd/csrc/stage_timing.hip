// stage_timing.hip -- see stage_timing.hpp
#include <mutex>
#include <vector>

#include "stage_timing.hpp"

namespace svoslam {

namespace {
std::mutex g_mu;  // stages are enqueued from several host threads / on several streams
unsigned g_mask = 0;
struct Log { std::vector<hipEvent_t> ev; size_t used = 0; };  // pairs (start, stop) since the last read
Log g_log[kStageCount];
}  // namespace

unsigned stage_timing_mask() {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_mask;
}

int stage_timing(unsigned mask) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_mask = mask;
  for (Log &l : g_log) l.used = 0;
  return SVOSLAM_OK;
}

int stage_timing_read(int stage, float *ms_sum, int *pairs) {
  if (stage < 0 || stage >= kStageCount || !ms_sum || !pairs) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  Log &l = g_log[stage];
  float total = 0.0f;
  for (size_t i = 0; i + 1 < l.used; i += 2) {
    SVO_HIP(hipEventSynchronize(l.ev[i + 1]));
    float ms = 0.0f;
    SVO_HIP(hipEventElapsedTime(&ms, l.ev[i], l.ev[i + 1]));
    total += ms;
  }
  *ms_sum = total;
  *pairs = (int)(l.used / 2);
  l.used = 0;
  return SVOSLAM_OK;
}

int stage_event(int stage, hipStream_t stream) {
  if (stage < 0 || stage >= kStageCount) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  if (!(g_mask & (1u << stage))) return SVOSLAM_OK;
  Log &l = g_log[stage];
  if (l.used == l.ev.size()) {
    hipEvent_t e;
    SVO_HIP(hipEventCreate(&e));
    l.ev.push_back(e);
  }
  SVO_HIP(hipEventRecord(l.ev[l.used++], stream));
  return SVOSLAM_OK;
}

}  // namespace svoslam
