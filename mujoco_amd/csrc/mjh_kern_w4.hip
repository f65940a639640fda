// libmjhip.so, translation unit of namespace w4: four environments per wavefront (one 16-lane DPP
// row each), LEAN feature set, 512 VGPRs (1 wave per SIMD).
#define MJH_BUILD_W4 1
#include "mjh_kernels.h"
MJH_DEFINE_WAVE_KERNELS(w4, 4, 1, w4::wv_sub())
