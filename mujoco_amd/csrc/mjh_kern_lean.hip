// libmjhip.so, translation unit of namespace wl: one wavefront per environment, LEAN feature set.
#define MJH_BUILD_WL 1
#include "mjh_kernels.h"
MJH_DEFINE_WAVE_KERNELS(wl, 1, 4, 0)
