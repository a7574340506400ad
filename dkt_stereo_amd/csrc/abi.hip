// abi.hip -- version / error-string entry points of libdktstereo.
#include "dkt_common.h"

extern "C" int dkt_version(void) { return DKT_ABI_VERSION; }

// Timing-only ablation masks the convolution kernels were compiled with (conv2d.hip CONV_ABL per translation unit in the
// low byte, conv_c8.hip C8_ABL in the next one).  0 in the product: dkt_stereo_amd.build asserts it after linking and
// _ffi.lib() refuses a library that reports anything else (ablation builds compute wrong results by construction).
int conv2d_abl_p1();
int conv2d_abl_p2();
int conv2d_abl_p3();
int conv_c8_abl();
extern "C" int dkt_build_ablation(void) { return (conv2d_abl_p1() | conv2d_abl_p2() | conv2d_abl_p3()) | (conv_c8_abl() << 8); }

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
