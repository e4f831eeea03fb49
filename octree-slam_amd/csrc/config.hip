// config.hip -- see config.hpp
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "config.hpp"

namespace svoslam {
namespace {
std::mutex g_mu;
bool g_ready = false;
svoslam_config g_cfg;

struct Field { const char *name; int32_t svoslam_config::*member; };
const Field kFields[] = {
    {"march_bricks", &svoslam_config::march_bricks}, {"track_mode", &svoslam_config::track_mode},
    {"track_workers", &svoslam_config::track_workers}, {"track_stream", &svoslam_config::track_stream},
    {"runner_deferred", &svoslam_config::runner_deferred}, {"runner_lead", &svoslam_config::runner_lead},
    {"runner_prio", &svoslam_config::runner_prio}, {"runner_replicas", &svoslam_config::runner_replicas},
    {"runner_timeline", &svoslam_config::runner_timeline}, {"sort_pairs", &svoslam_config::sort_pairs},
    {"graphs", &svoslam_config::graphs},
};

void init_locked() {
  if (g_ready) return;
  memset(&g_cfg, 0, sizeof(g_cfg));
  g_cfg.march_bricks = 1; g_cfg.track_stream = 1;
  g_cfg.runner_deferred = -1; g_cfg.runner_lead = -1; g_cfg.runner_prio = -1; g_cfg.runner_replicas = 1;
  // SVOSLAM_CONFIG = "name=value,name=value": the ONE environment variable the library reads (tools, child-process tests)
  if (const char *e = getenv("SVOSLAM_CONFIG")) {
    const char *p = e;
    while (*p) {
      const char *eq = strchr(p, '='), *end = strchr(p, ',');
      if (!end) end = p + strlen(p);
      if (eq && eq < end) {
        for (const Field &f : kFields)
          if ((size_t)(eq - p) == strlen(f.name) && strncmp(p, f.name, (size_t)(eq - p)) == 0) g_cfg.*(f.member) = (int32_t)atoi(eq + 1);
      }
      p = *end ? end + 1 : end;
    }
  }
  g_ready = true;
}
}  // namespace

svoslam_config config() {
  std::lock_guard<std::mutex> lock(g_mu);
  init_locked();
  return g_cfg;
}
}  // namespace svoslam

extern "C" {
int svoslam_config_get(svoslam_config *out) {
  if (!out) return SVOSLAM_ERR_INVALID_ARG;
  *out = svoslam::config();
  return SVOSLAM_OK;
}
int svoslam_config_set(const svoslam_config *in) {
  if (!in) return SVOSLAM_ERR_INVALID_ARG;
  if (in->runner_replicas != 1 && in->runner_replicas != 2) return SVOSLAM_ERR_INVALID_ARG;
  if (in->track_mode < 0 || in->track_mode > 2 || in->track_workers < 0) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(svoslam::g_mu);
  svoslam::init_locked();
  svoslam::g_cfg = *in;
  return SVOSLAM_OK;
}
}
