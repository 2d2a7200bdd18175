// libbds.so: ABI version + error strings.
#include "bds_common.h"

extern "C" int bds_abi_version(void) { return BDS_ABI_VERSION; }

extern "C" const char *bds_strerror(int code) {
  switch (code) {
    case BDS_OK: return "ok";
    case BDS_EINVAL: return "invalid argument (null/misaligned pointer, bad shape or unsupported parameter)";
    case BDS_EWORKSPACE: return "workspace too small";
    case BDS_ELAUNCH: return "HIP launch or copy failed";
    default: return "unknown bds error";
  }
}
