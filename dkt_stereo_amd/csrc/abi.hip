// abi.hip -- version / error-string entry points of libdktstereo.
#include "dkt_common.h"

extern "C" int dkt_version(void) { return DKT_ABI_VERSION; }

extern "C" const char *dkt_strerror(int rc) {
    if (rc == DKT_OK) return "ok";
    if (rc > 0) return hipGetErrorString((hipError_t)rc);
    switch (rc) {
        case DKT_E_NULL: return "null pointer argument";
        case DKT_E_SHAPE: return "non-positive or inconsistent dimension";
        case DKT_E_LEVELS: return "num_levels out of range or pyramid level of width 0";
        case DKT_E_RADIUS: return "radius out of range";
        case DKT_E_GROUPS: return "channels not divisible by groups";
        case DKT_E_ALIGN: return "alignment requirement not met";
        case DKT_E_UNSUPPORTED: return "configuration not supported by this build";
        default: return "unknown dktstereo error";
    }
}
