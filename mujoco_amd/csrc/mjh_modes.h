// The stage sources are written once against a small SPMD vocabulary (wv_lane, wv_sub, wv_sync,
// wv_bcast, wv_ballot, wv_sum_i, wv_exscan_i, wv_any, MJH_FOR_LANES, MJH_W) and a compile-time
// feature set (MJH_HAS, mjh_types.h), and compiled in several mappings, one namespace each:
//
//   wv : "one wavefront per environment", every feature.  MJH_W = 64 work items of a phase run on
//        the 64 lanes of the wavefront that owns the environment; phases are separated by
//        wv_sync(); cross-lane primitives are DPP/readlane/shuffles (mjh_spmd.h).
//   ws : the same mapping on SoA batches (strided views): constraint kernel of the 3-kernel pipeline.
//   ln : "one lane per environment".  MJH_W = 1: every GPU lane walks the serial algorithm of its
//        own environment, a wavefront steps 64 environments in lockstep; model constants are
//        wave-uniform (scalar loads), mjData fields are read SoA-across-environments.
//   wl : wv restricted to the LEAN feature set (PGS, pyramidal cones, Euler, primitive colliders,
//        no sensors / equalities / ...): the kernel a model like humanoid.xml actually needs,
//        without the stack frames and register pressure of everything else.
//
// A translation unit selects what it compiles with MJH_BUILD_<NS> (none given: everything, which
// is what the host emulation does); libmjhip.so compiles the namespaces in separate .hip files so
// they build in parallel.
#pragma once

#include "mjh_spmd.h"
#include "mjh_math.h"
#include "mjh_types.h"

#if !defined(MJH_BUILD_WV) && !defined(MJH_BUILD_WS) && !defined(MJH_BUILD_LN) && !defined(MJH_BUILD_WL)
#define MJH_BUILD_WV 1
#define MJH_BUILD_WS 1
#define MJH_BUILD_LN 1
#define MJH_BUILD_WL 1
#endif

// kernel variants (mjhipBatch_::variant, RolloutArgs consumers): which namespace steps a batch
// (round 2 also carried "lean2" / "lean4": two / four environments per wavefront.  Measured 2-3x slower
// at every batch size of interest (profiles/r02_variants) and deleted in round 3.)
enum { MJH_VAR_GENERIC = 0, MJH_VAR_LEAN = 1, MJH_NVARIANT = 2 };
static inline int mjh_variant_nsub(int) { return 1; }
static inline int mjh_variant_features(int v) { return v == MJH_VAR_GENERIC ? MJH_FT_ALL : MJH_FT_LEAN; }
static inline const char* mjh_variant_name(int v) {
  return v == MJH_VAR_GENERIC ? "generic" : "lean";
}

// ------------------------------------------------------------------------------------------------
// wave mode (one environment per wavefront)
// ------------------------------------------------------------------------------------------------
#define MJH_W 64
#define MJH_LANE_MODE 0
#define MJH_DEVN MJH_DEVN_WAVE
#define MJH_FOR_LANES(i, n) for (int i = wv_lane(); i < (n); i += MJH_W)
// entry of an out-of-line stage function: its arguments arrive in VGPRs; all three are wave-uniform
// namespace wv serves environment-major batches only (B.soa == 0): telling the compiler makes every
// strided view a unit-stride view (no index multiply per access)
#if defined(MJH_HOSTSIM) || !defined(MJH_STAGE_LAUNDER)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
#else
// (experiment -DMJH_STAGE_LAUNDER: every stage reaches the descriptors through pointers the compiler cannot relate to the
// previous stage's, so a descriptor word is re-read from the scalar cache per stage instead of being carried in SGPRs)
#define MJH_ENTER(M_, B_, e_) const MJH_CONST_AS DModel* mp_ = &wv_uniform_ref(M_); const MJH_CONST_AS DBatch* bp_ = &wv_uniform_ref(B_); \
                              asm volatile("" : "+s"(mp_), "+s"(bp_)); \
                              MREF M = *mp_; BREF B = *bp_; const int e = wv_uniform_i(e_); \
                              if (B.soa != 0) __builtin_unreachable()
#endif
#if MJH_BUILD_WV
#define MJH_FEATURES MJH_FT_ALL
namespace wv {
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#if MJH_BUILD_WL
#define MJH_FEATURES MJH_FT_LEAN
namespace wl {
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#undef MJH_ENTER
// namespace ws: the same wave mapping on SoA batches (constraint kernel of the per-step pipeline)
#define MJH_ENTER(M_, B_, e_) MREF M = wv_uniform_ref(M_); BREF B = wv_uniform_ref(B_); const int e = wv_uniform_i(e_)
#if MJH_BUILD_WS
#define MJH_FEATURES MJH_FT_ALL
namespace ws {
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#undef MJH_W
#undef MJH_FOR_LANES
#undef MJH_ENTER

#undef MJH_LANE_MODE
#undef MJH_DEVN

// ------------------------------------------------------------------------------------------------
// lane mode
// ------------------------------------------------------------------------------------------------
#define MJH_W 1
#define MJH_LANE_MODE 1
#define MJH_DEVN MJH_DEVN_LANE
#define MJH_FOR_LANES(i, n) for (int i = 0; i < (n); i++)
// (stage functions are inlined into the kernel here: descriptors are already uniform, e is per lane)
#define MJH_ENTER(M_, B_, e_) MREF M = M_; BREF B = B_; const int e = e_
#if MJH_BUILD_LN
#define MJH_FEATURES MJH_FT_ALL
namespace ln {
// the SPMD vocabulary for a width-1 "wave": these hide the wavefront primitives of mjh_spmd.h
MJH_DEV int wv_lane() { return 0; }
MJH_DEV int wv_sub() { return 0; }
MJH_DEV void wv_sync() {}
MJH_DEV double wv_bcast(double v, int) { return v; }
MJH_DEV int wv_bcast_i(int v, int) { return v; }
MJH_DEV uint64_t wv_ballot(int pred) { return pred ? 1ull : 0ull; }
MJH_DEV int wv_sum_i(int v) { return v; }
MJH_DEV int wv_exscan_i(int) { return 0; }
MJH_DEV int wv_any(int pred) { return pred != 0; }
#include "mjh_stages.inc"
}
#undef MJH_FEATURES
#endif
#undef MJH_W
#undef MJH_LANE_MODE
#undef MJH_DEVN
#undef MJH_FOR_LANES
#undef MJH_ENTER
