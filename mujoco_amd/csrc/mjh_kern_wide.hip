// libmjhip.so, translation unit of the generic kernels (namespace wv, every feature) compiled for TWO wavefronts per
// SIMD: 256 VGPRs instead of the 128 that keep 4096 one-environment wavefronts co-resident.  A launch of at most
// 2 x 4 x CUs environments (BASELINE config 4's 2048, config 5's 256) cannot put more than two wavefronts on a SIMD
// anyway, and the generic stage functions -- GJK / EPA, the primal solvers, the flex passes -- spill heavily at 128.
// Same sources, same operation order: results are bit-identical to the 4-per-SIMD build (the GPU suite runs on this
// one, its batches being small; bench.py's parity sample covers both).
#define MJH_BUILD_WV 1
#define MJH_WIDE_REGS 1
#include "mjh_kernels.h"
MJH_DEFINE_WAVE_KERNELS_AS(wv, wv2, 1, 2, 0)
