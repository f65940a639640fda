// libmjhip.so, translation unit of the multi-wavefront kernels: MJH_MW wavefronts per environment (mjh_modes.h,
// namespaces wn + wq), for launches that leave most of a CU idle under the one-wavefront mapping -- flex models at
// BASELINE config 5's 256 environments per GPU (one workgroup per CU).  Wave 0 runs the step; kinematics, comPos, the
// flex position / edge / passive passes, comVel and rne run on all wavefronts of the workgroup.  Two wavefronts per
// SIMD at most (the launch has one workgroup per CU): the 256-VGPR budget of mjh_kern_wide.hip.
#define MJH_BUILD_WN 1
#define MJH_WIDE_REGS 1
#include "mjh_kernels.h"
MJH_DEFINE_MULTIWAVE_KERNELS(2)
