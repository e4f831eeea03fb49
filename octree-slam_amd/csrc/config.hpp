// config.hpp -- the library's settings (include/svoslam.h svoslam_config): one struct, preset from SVOSLAM_CONFIG
#pragma once
#include "common.hpp"

namespace svoslam {
svoslam_config config();  // (a copy: read where a setting is used; set through svoslam_config_set)
}
