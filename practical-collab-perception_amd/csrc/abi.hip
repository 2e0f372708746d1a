// ABI bookkeeping of libpcp_hip.so.
#include "pcp_common.h"

extern "C" int pcp_abi_version(void) { return PCP_ABI_VERSION; }

extern "C" const char *pcp_status_string(int status) {
  switch (status) {
    case PCP_OK: return "ok";
    case PCP_ERR_ARG: return "bad argument";
    case PCP_ERR_WORKSPACE: return "workspace too small";
    case PCP_ERR_LAUNCH: return "kernel launch failed";
    case PCP_ERR_UNSUPPORTED: return "unsupported shape for this build";
    default: return "unknown status";
  }
}
