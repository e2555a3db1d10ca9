#include "../../include/tbx_hip.h"

extern "C" int tbx_version(void) { return TBX_ABI_VERSION; }

extern "C" const char* tbx_error_string(int code) {
  switch (code) {
    case TBX_OK: return "ok";
    case TBX_ERR_ARG: return "invalid argument (null pointer or non-positive size)";
    case TBX_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case TBX_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
    case TBX_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown tbx error";
  }
}
