// TEST INFRASTRUCTURE (oracle build only, never linked into the product).
//
// liboracle_dm.so = the compiled reference (same objects as liboracle.so) whose calls to sin / cos / sincos / atan2 / exp / acos / asin
// bind to mjh_sincos / mjh_atan2 / mjh_exp / mjh_acos / mjh_asin (mujoco_amd/csrc/mjh_math.h) -- the routine the HIP kernels evaluate, bit-reproducible between host and
// device (explicit fma, no tables) -- instead of glibc's.  Both are < 1 ulp; they differ in the last bit for a small
// fraction of arguments, and on a stiff contact-rich model (cube_3x3x3: aligned cubelet faces, EPA / face clipping
// are discontinuous in the poses) one such bit can move a contact point by millimetres.  Comparing the GPU with THIS
// build separates "the kernels follow the reference operation for operation" (then the results are identical) from
// "the platform's libm rounds sin(x) the other way" (which no implementation can match across platforms).
// The glibc build stays the primary oracle; tests and bench.py report against both.
#define MJH_HOSTSIM 1
#include "../mujoco_amd/csrc/mjh_math.h"

namespace mjhsim { thread_local WaveSim* g_wave = nullptr; }

extern "C" {
__attribute__((visibility("default"))) double sin(double x) { double s, c; mjh_sincos(x, &s, &c); return s; }
__attribute__((visibility("default"))) double cos(double x) { double s, c; mjh_sincos(x, &s, &c); return c; }
__attribute__((visibility("default"))) void sincos(double x, double* s, double* c) { mjh_sincos(x, s, c); }
__attribute__((visibility("default"))) double atan2(double y, double x) { return mjh_atan2(y, x); }
__attribute__((visibility("default"))) double exp(double x) { return mjh_exp(x); }
// (tendon wrapping, engine_util_misc.c: the only acos / asin of the stepping path)
__attribute__((visibility("default"))) double acos(double x) { return mjh_acos(x); }
__attribute__((visibility("default"))) double asin(double x) { return mjh_asin(x); }
}
