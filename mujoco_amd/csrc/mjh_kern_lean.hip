// libmjhip.so, translation unit of namespace wl: one wavefront per environment, LEAN feature set.
// The lean stage functions are INLINED into the kernel (the generic build keeps them out of line to
// bound register pressure across its many features): an out-of-line stage saves and restores the
// callee-saved VGPRs it uses (v40-47, v56-63, ... of the AMDGPU calling convention) on every call --
// ~32 dword stores + loads per call and lane, every step -- which was two thirds of the kernel's
// HBM traffic; measured 153 KB -> 47 KB per env-step at equal speed (profiles/r02c).
#define MJH_INLINE_STAGES 1
#define MJH_BUILD_WL 1
#include "mjh_kernels.h"
// (MJH_LEAN_WPE: measurement builds only -- tools/gpu_traffic.sh compiles a 2-waves-per-SIMD / 256-VGPR
// copy to separate spill traffic from the rest)
#ifndef MJH_LEAN_WPE
#define MJH_LEAN_WPE 4
#endif
MJH_DEFINE_WAVE_KERNELS(wl, 1, MJH_LEAN_WPE, 0)
