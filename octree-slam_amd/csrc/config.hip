// config.hip -- see config.hpp
#include <cstdio>
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
    {"graphs", &svoslam_config::graphs}, {"march_ahead", &svoslam_config::march_ahead},
};

bool config_valid(const svoslam_config &c) {
  if (c.runner_replicas != 1 && c.runner_replicas != 2) return false;
  if (c.track_mode < 0 || c.track_mode > 2 || c.track_workers < 0) return false;
  return true;
}

void init_locked() {
  if (g_ready) return;
  memset(&g_cfg, 0, sizeof(g_cfg));
  g_cfg.march_bricks = 1; g_cfg.track_stream = 1; g_cfg.march_ahead = 60;
  g_cfg.runner_deferred = -1; g_cfg.runner_lead = -1; g_cfg.runner_prio = -1; g_cfg.runner_replicas = 1;
  // SVOSLAM_CONFIG = "name=value,name=value": the ONE environment variable the library reads (tools, child-process tests).
  // Validated like svoslam_config_set: an unknown name or a malformed pair is reported once on stderr (a typo in an
  // A/B script would otherwise measure the default under the wrong label), an out-of-range value is reported and NOT taken
  // (a later get -> modify -> set of an unrelated field would fail on it).
  if (const char *e = getenv("SVOSLAM_CONFIG")) {
    const char *p = e;
    while (*p) {
      const char *eq = strchr(p, '='), *end = strchr(p, ',');
      if (!end) end = p + strlen(p);
      bool known = false;
      if (eq && eq < end && eq > p) {
        for (const Field &f : kFields)
          if ((size_t)(eq - p) == strlen(f.name) && strncmp(p, f.name, (size_t)(eq - p)) == 0) {
            known = true;
            svoslam_config trial = g_cfg;
            char *stop = nullptr;
            const long v = strtol(eq + 1, &stop, 10);
            trial.*(f.member) = (int32_t)v;
            if (stop == eq + 1 || stop != end || !config_valid(trial))
              fprintf(stderr, "svoslam: SVOSLAM_CONFIG: value of '%s' rejected (\"%.*s\"); the default stays\n", f.name, (int)(end - p), p);
            else
              g_cfg = trial;
          }
      }
      if (!known && end > p) fprintf(stderr, "svoslam: SVOSLAM_CONFIG: unknown or malformed entry \"%.*s\" ignored\n", (int)(end - p), p);
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
  if (!svoslam::config_valid(*in)) return SVOSLAM_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(svoslam::g_mu);
  svoslam::init_locked();
  svoslam::g_cfg = *in;
  return SVOSLAM_OK;
}
}
