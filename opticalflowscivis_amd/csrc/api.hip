// api.hip -- version / error-string entry points of the C-ABI.
#include "common.hpp"

extern "C" int fs_version(void) {
  FS_ENTER(); return FS_ABI_VERSION; }

extern "C" const char* fs_error_string(int code) {
  switch (code) {
    case FS_OK: return "ok";
    case FS_ERR_NULLPTR: return "required pointer is NULL";
    case FS_ERR_SHAPE: return "size out of the supported range";
    case FS_ERR_ARG: return "invalid option / mode";
    case FS_ERR_LAUNCH: return "kernel launch failed";
    case FS_ERR_UNSUPPORTED: return "no fused kernel for this shape / alignment (use the unfused entry points)";
    default: return "unknown error code";
  }
}
