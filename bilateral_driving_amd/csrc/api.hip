// libbds.so: ABI version + error strings.
#include "bds_common.h"

extern "C" int bds_abi_version(void) { return BDS_ABI_VERSION; }

extern "C" const char *bds_strerror(int code) {
  switch (code) {
    case BDS_OK: return "ok";
    case BDS_EINVAL: return "invalid argument (null/misaligned pointer, bad shape or unsupported parameter)";
    case BDS_EWORKSPACE: return "workspace too small";
    case BDS_ELAUNCH: return "HIP launch or copy failed";
    case BDS_ECAPACITY: return "more intersections than the caller's buffers hold";
    default: return "unknown bds error";
  }
}

namespace bds {
static int g_options[kOptCount] = {0, 0, 0, /*debug*/ 0, /*short_sort*/ 1, 0, /*packed*/ 1, 0};
int option_get(int which) { return (which >= 0 && which < kOptCount) ? g_options[which] : 0; }
}  // namespace bds

extern "C" int bds_set_option(int which, int value) {
  if (which < 0 || which >= bds::kOptCount) return BDS_EINVAL;
  bds::g_options[which] = value;
  return BDS_OK;
}
extern "C" int bds_get_option(int which) { return bds::option_get(which); }
