// libmjhip.so, translation unit of namespace w2: two environments per wavefront (32 lanes each),
// LEAN feature set, 256 VGPRs (2 waves per SIMD).
#define MJH_BUILD_W2 1
#include "mjh_kernels.h"
MJH_DEFINE_WAVE_KERNELS(w2, 2, 2, w2::wv_sub())
